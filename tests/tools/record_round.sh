#!/bin/bash
# The round's lines of record in ONE GPU call, most important first (every step under its own timeout, nothing fatal):
#     gpurun --timeout 1500 -- 'bash tests/tools/record_round.sh r04'
# Writes gpurun_out/<tag>/ ; copy what is to be judged into profiles/ afterwards (profiles/README.md names the commands).
tag=${1:-r04}
out=gpurun_out/${tag}_record
mkdir -p "$out" gpurun_out/r04
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a "$out/steps.log"; }
py=python

stamp "gpu tests"
timeout 420 $py -m pytest tests -m gpu -x -q > "$out/pytest_gpu.log" 2>&1; echo "rc $?" >> "$out/pytest_gpu.log"
tail -3 "$out/pytest_gpu.log" | tee -a "$out/steps.log"

stamp "smoke"
timeout 120 $py -c "import __graft_entry__ as g; g.smoke()" > "$out/smoke.log" 2>&1; echo "rc $?" >> "$out/smoke.log"

stamp "PMC passes (kernels alone, warm + cold operating points)"
timeout 480 $py tests/tools/pmc_collect.py > "$out/pmc_collect.log" 2>&1; echo "rc $?" >> "$out/pmc_collect.log"
if [ -s gpurun_out/r04/pmc_diff_1k_T10_K10.json ]; then
  cp gpurun_out/r04/pmc_diff_1k_T10_K10.json profiles/${tag}_pmc.json          # (on the box: the bench lines below read it)
  cp gpurun_out/r04/pmc_diff_1k_T10_K10.json "$out/${tag}_pmc.json"
fi

stamp "default line"
timeout 480 $py bench.py > "$out/${tag}_bench.json" 2> "$out/bench.err"; echo "rc $?" >> "$out/bench.err"
stamp "driver flags"
timeout 420 $py bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras > "$out/${tag}_bench_driver_flags.json" 2> "$out/bench_drv.err"
for i in 2 3; do timeout 120 $py bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --no-latency > "$out/drv_rep$i.json" 2>/dev/null; done

stamp "one rank under torch.distributed.run"
timeout 180 $py -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu --no-latency --no-extras > "$out/${tag}_bench_torchrun1.json" 2> "$out/torchrun1.err"
timeout 180 $py -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --no-extras > "$out/${tag}_bench_torchrun1_driver_flags.json" 2>> "$out/torchrun1.err"

stamp "kernel trace of the default command / sequential steps"
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $py $GRAFT_REPO_ROOT/bench.py --no-cpu --no-latency --no-extras > /dev/null 2>&1 )
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats.csv"
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks1 -o ks -- $py $GRAFT_REPO_ROOT/bench.py --no-cpu --no-latency --no-extras --inflight 1 --steps 32 --warmup 8 > /dev/null 2>&1 )
f=$(find /tmp/ks1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" "$out/${tag}_kernel_stats_inflight1.csv"

stamp "sequential steps, cold QP, iteration statistics"
timeout 120 $py bench.py --inflight 1 --no-cpu --no-latency --no-extras --steps 32 --warmup 8 > "$out/${tag}_bench_inflight1.json" 2>/dev/null
timeout 120 env NPA_QP_COLD=1 $py bench.py --no-cpu --no-latency --no-extras > "$out/${tag}_bench_coldqp.json" 2>/dev/null
timeout 120 $py tests/tools/qp_iter_stats.py > "$out/${tag}_qp_iter_stats.txt" 2>&1

stamp "launch shape (flags only): longer run, larger launches on fewer queues"
timeout 120 $py bench.py --no-cpu --no-latency --no-extras --steps 400 --warmup 40 > "$out/steps400.json" 2>/dev/null
for sh in "512 10" "1024 5" "128 22"; do
  set -- $sh
  timeout 120 $py bench.py --no-cpu --no-latency --no-extras --batch $1 --inflight $2 --steps $(( 32768 / $1 )) --warmup $(( 8192 / $1 )) > "$out/shape_b$1_f$2.json" 2>/dev/null
done
stamp "a burst of 5120 scenes (the driver's 20 steps) as fewer, larger launches; issue order"
for sh in "512 10 10" "1024 5 5" "1280 4 4"; do
  set -- $sh
  timeout 120 $py bench.py --no-cpu --no-latency --no-extras --batch $1 --inflight $2 --steps $3 --warmup 2 > "$out/burst5120_b$1_f$2.json" 2>/dev/null
done
timeout 200 $py tests/tools/burst_sweep.py 5 > "$out/${tag}_burst_sweep.txt" 2>/dev/null
stamp "done"
