cd /root/repo
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_default.json 2>gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.json
python bench.py --inflight 1 --no-cpu | tail -1 > gpurun_out/bench_inf1.json; cat gpurun_out/bench_inf1.json
python bench.py --workload acker_2k_T20_K15 --no-cpu | tail -1
python bench.py --workload dyna_4k_T10_K10 --no-cpu | tail -1
