cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02/final3; O=gpurun_out/r02/final3
timeout 2400 python -m pytest tests/ -x -q -m gpu > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 900 python tests/tools/pmc_collect.py diff_1k_T10_K10 > $O/pmc_collect.log 2>&1; cp gpurun_out/r02/pmc_diff_1k_T10_K10.json profiles/r02_pmc.json; cp profiles/r02_pmc.json $O/r02_pmc.json
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_default -o p -- python $R/bench.py --no-cpu --no-latency > $R/$O/bench_under_rocprof.json 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_if1 -o p -- python $R/bench.py --no-cpu --no-latency --inflight 1 > $R/$O/bench_if1_under_rocprof.json 2>/dev/null
cd $R
timeout 900 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
timeout 300 python bench.py --inflight 1 --no-cpu > $O/r02_bench_inflight1.json 2>> $O/r02_bench.err
NPA_QP_COLD=1 timeout 300 python bench.py --no-cpu > $O/r02_bench_coldqp.json 2>> $O/r02_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu --no-latency > $O/r02_bench_torchrun1.json 2>> $O/r02_bench.err
timeout 300 python bench.py --workload dyna_4k_T10_K10 --batch 1024 --inflight 4 --no-cpu --no-latency --steps 32 --warmup 8 > $O/r02_bench_dyna_b1024.json 2>> $O/r02_bench.err
for w in acker_2k_T20_K15 dyna_4k_T10_K10 poly8_5k_T10_K10; do timeout 900 python bench.py --workload $w --cpu-scenes 64 > $O/r02_bench_$w.json 2>> $O/r02_bench.err; done
for n in 1 2 4 8 12 16 20 24 32; do timeout 200 python bench.py --no-cpu --no-latency --inflight $n 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('GPU_MAX_HW_QUEUES=32 --inflight %2d: %9.1f plans/s  %.4f ms/step  host issue %.4f ms/step  QP launch %.4f ms  select %.4f ms' % ($n,d['value'],d['ms_per_step'],d['host_issue_ms_per_step'],r['launch_ms'],r['select_launch_ms']))"; done > $O/r02_inflight_sweep.txt
for q in 2 4 8 16 32; do GPU_MAX_HW_QUEUES=$q timeout 200 python bench.py --no-cpu --no-latency 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=d['roofline']
print('GPU_MAX_HW_QUEUES=%2d (default --inflight): %9.1f plans/s  %.4f ms/step  QP launch %.4f ms  select %.4f ms' % ($q,d['value'],d['ms_per_step'],r['launch_ms'],r['select_launch_ms']))"; done > $O/r02_queue_sweep.txt
GPU_MAX_HW_QUEUES=32 timeout 300 python tests/tools/qp_scaling.py diff_1k_T10_K10 256 2>&1 | grep -v amdgpu.ids > $O/r02_qp_scaling.txt
GPU_MAX_HW_QUEUES=32 timeout 300 python tests/tools/select_scaling.py diff_1k_T10_K10 256 2>&1 | grep -v amdgpu.ids > $O/r02_select_scaling.txt
timeout 300 python tests/tools/qp_iter_stats.py 2>&1 | grep -v amdgpu.ids > $O/r02_qp_iter_stats.txt
timeout 300 python tests/tools/qp_phase_cycles.py 2>&1 | grep -v "amdgpu.ids\|hipcc" > $O/r02_qp_phase_cycles.txt
timeout 300 python tests/tools/geo_check.py 2>&1 | grep -v amdgpu.ids > $O/r02_geo_check.txt
python - <<'PY'
import json,csv,glob
O='gpurun_out/r02/final3'
for f in sorted(glob.glob(O+'/*.json')):
    try: d=json.loads(open(f).read().strip().split('\n')[-1])
    except Exception as e: print(f, 'unreadable', e); continue
    if 'roofline' not in d: continue
    r=d['roofline']
    print(f.split('/')[-1],d['value'],d['ms_per_step'],'qp',r['launch_ms'],'sel',r['select_launch_ms'],'frac',r.get('frac'),d.get('latency_B1_ms',{}) and d['latency_B1_ms'].get('K10_N1000'))
    p=d.get('parity')
    if p: print('   parity', {k:p[k] for k in p if k in ('scenes','scenes_well_posed','max_over_well_posed','A_well_posed_all_le_tol','B_others_inside_envelope','C_le_1e-5_until_ensemble_diverges')}, p.get('gpu_last_qp',{}).get('obj_gap_rel'))
    if d.get('cpu_baseline'): print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('seconds_per_plan_in_worker'))
for f in ('prof_default','prof_if1'):
    for r in list(csv.DictReader(open(O+'/'+f+'/p_kernel_stats.csv')))[:3]: print(f,r['Name'][:34],r['Calls'],float(r['AverageNs'])/1e3,r['Percentage'])
PY
