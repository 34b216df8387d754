cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RES', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['nrmp_qp_launch_ms'], d['parity'])"
python bench.py --inflight 1 --no-cpu | tail -1 > gpurun_out/bench_inf1.json; cat gpurun_out/bench_inf1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RES', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'], d['roofline']['nrmp_qp_launch_ms'])"
