cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r02/final4; O=gpurun_out/r02/final4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "selection or far_clouds or dune_stage or maximum_slice or poly" > $O/pytest_sel.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest_sel.log
timeout 900 python tests/tools/pmc_collect.py diff_1k_T10_K10 > $O/pmc_collect.log 2>&1; cp gpurun_out/r02/pmc_diff_1k_T10_K10.json profiles/r02_pmc.json; cp profiles/r02_pmc.json $O/r02_pmc.json
timeout 900 python bench.py > $O/r02_bench.json 2> $O/r02_bench.err
timeout 300 python bench.py --inflight 1 --no-cpu > $O/r02_bench_inflight1.json 2>> $O/r02_bench.err
NPA_QP_COLD=1 timeout 300 python bench.py --no-cpu > $O/r02_bench_coldqp.json 2>> $O/r02_bench.err
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --no-cpu --no-latency > $O/r02_bench_torchrun1.json 2>> $O/r02_bench.err
timeout 900 python bench.py --workload poly8_5k_T10_K10 --cpu-scenes 64 > $O/r02_bench_poly8_5k_T10_K10.json 2>> $O/r02_bench.err
python - <<'PY'
import json,glob
O='gpurun_out/r02/final4'
for f in sorted(glob.glob(O+'/r02_bench*.json')):
    d=json.loads(open(f).read().strip().split('\n')[-1]); r=d['roofline']
    print(f.split('/')[-1],d['value'],d['ms_per_step'],'qp',r['launch_ms'],'sel',r['select_launch_ms'],'frac',r.get('frac'),(d.get('latency_B1_ms') or {}).get('K10_N1000'))
    p=d.get('parity')
    if p: print('   parity', {k:p[k] for k in p if k in ('scenes','scenes_well_posed','max_over_well_posed','A_well_posed_all_le_tol','B_others_inside_envelope','C_le_1e-5_until_ensemble_diverges')})
PY
