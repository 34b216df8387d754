cd /root/repo
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "eight_edge" 2>&1 | tail -5
python bench.py --workload poly8_5k_T10_K10 --no-cpu | tail -1 > gpurun_out/r01e_bench_poly8.json; python -c "
import json; d=json.loads(open('gpurun_out/r01e_bench_poly8.json').read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel'], r['frac'], r['launch_ms'], r['nrmp_qp_launch_ms'])"
