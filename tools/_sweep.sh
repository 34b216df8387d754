cd /root/repo
R=/root/repo
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py > gpurun_out/r01f_bench.json 2> gpurun_out/r01f_bench.err; tail -c 300 gpurun_out/r01f_bench.json; echo
python bench.py --inflight 1 --no-cpu | tail -1 > gpurun_out/r01f_bench_inflight1.json
python bench.py --workload acker_2k_T20_K15 --no-cpu | tail -1 > gpurun_out/r01f_bench_acker.json
python bench.py --workload dyna_4k_T10_K10 --no-cpu | tail -1 > gpurun_out/r01f_bench_dyna.json
python bench.py --workload poly8_5k_T10_K10 --no-cpu | tail -1 > gpurun_out/r01f_bench_poly8.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f -o f -- python $R/bench.py --no-cpu > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_f1 -o f1 -- python $R/bench.py --no-cpu --inflight 1 > /dev/null 2>&1
cp $(find $R/gpurun_out/prof_f -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01f_kernel_stats.csv
cp $(find $R/gpurun_out/prof_f1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r01f_kernel_stats_inflight1.csv
rm -rf $R/gpurun_out/prof_f $R/gpurun_out/prof_f1
cd $R && python tools/hbm_traffic.py > /dev/null 2>&1; cp gpurun_out/traffic.json gpurun_out/r01f_traffic.json
head -4 gpurun_out/r01f_kernel_stats.csv | cut -c1-60,400-520; head -4 gpurun_out/r01f_kernel_stats_inflight1.csv | cut -c1-60,400-520
python tools/frontend_bench.py 2>/dev/null | tail -1 > gpurun_out/r01f_frontend.json
