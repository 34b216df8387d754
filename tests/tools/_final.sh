cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02/final; O=gpurun_out/r02/final
timeout 900 python tests/tools/pmc_collect.py diff_1k_T10_K10 > $O/pmc_collect.log 2>&1; cp gpurun_out/r02/pmc_diff_1k_T10_K10.json profiles/r02_pmc.json; cp profiles/r02_pmc.json $O/r02_pmc.json
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_default -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-latency > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_if1 -o p -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-latency --inflight 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
timeout 300 python bench.py --inflight 1 --no-cpu > $O/r02_bench_inflight1.json 2>> $O/r02_bench.err
for w in acker_2k_T20_K15 dyna_4k_T10_K10 poly8_5k_T10_K10; do timeout 900 python bench.py --workload $w --cpu-scenes 64 > $O/r02_bench_$w.json 2>> $O/r02_bench.err; done
timeout 300 python bench.py --workload dyna_4k_T10_K10 --batch 1024 --inflight 4 --no-cpu --no-latency --steps 32 --warmup 8 > $O/r02_bench_dyna_b1024.json 2>> $O/r02_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu --no-latency > $O/r02_bench_torchrun1.json 2>> $O/r02_bench.err
for q in 2 4 8 16 32; do GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu --no-latency 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('queues',$q,'inflight 16:',d['value'],'plans/s')"; done > $O/queue_sweep.txt
for nf in 1 2 4 8 16 24 32; do timeout 300 python bench.py --no-cpu --no-latency --inflight $nf 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('inflight',$nf,d['value'],'plans/s  qp',d['roofline']['launch_ms'],'ms select',d['roofline']['select_launch_ms'],'ms')"; done > $O/inflight_sweep.txt
timeout 600 python tests/tools/geo_check.py > $O/geo_check.log 2>&1
NPA_QP_COLD=1 timeout 300 python bench.py --no-cpu > $O/r02_bench_coldqp.json 2>> $O/r02_bench.err
timeout 300 python tests/tools/qp_iter_stats.py > $O/qp_iter_stats.log 2>&1
timeout 300 python tests/tools/qp_scaling.py diff_1k_T10_K10 256 > $O/qp_scaling.log 2>&1
cat $O/queue_sweep.txt $O/inflight_sweep.txt; tail -3 $O/r02_bench.err
for f in $O/r02_bench*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().split('\n')[-1]); p=d.get('parity',{})
print('$f'.split('/')[-1], d['value'], d['ms_per_step'], d['roofline'].get('frac'), d.get('latency_B1_ms',{}).get('K10_N1000'), d.get('cpu_baseline',{}).get('value'), {k:p.get(k) for k in ('scenes','ctrl_l2_vs_oracle_median','scenes_well_posed','max_over_well_posed','A_well_posed_all_le_tol','B_others_inside_envelope','C_le_1e-5_until_ensemble_diverges','C_violations')} if p else '')"; done
