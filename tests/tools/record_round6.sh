#!/bin/bash
# Round 6: the lines of record in ONE GPU call, most important first (every step under its own timeout, nothing fatal):
#     gpurun --timeout 3000 -- 'bash tests/tools/record_round6.sh --pmc'      (--pmc: re-collect the counter records first, after a kernel change;
#     python tests/tools/adopt_records.py [--pmc] afterwards copies the results to their tracked names under profiles/ and prints the figures)
# Writes gpurun_out/r06final/ (and gpurun_out/r06/pmc_*.json); profiles/README.md names the commands.
out=gpurun_out/r06final
rm -rf $out; mkdir -p $out
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
t0=$(date +%s)
stamp() { echo "[$(( $(date +%s) - t0 )) s] $*" | tee -a $out/steps.log; }
py=python
if [ "$1" = "--pmc" ]; then      # (counters first: the bench lines below price their roofline with profiles/r06_pmc*.json)
  stamp "PMC passes (kernels alone)"
  mkdir -p gpurun_out/r06
  for w in diff_1k_T10_K10 acker_2k_T20_K15 poly8_5k_T10_K10; do ( cd /tmp; timeout 450 $py $GRAFT_REPO_ROOT/tests/tools/pmc_collect.py $w > $GRAFT_REPO_ROOT/gpurun_out/r06/pmc_$w.log 2>&1 ); done
  cp gpurun_out/r06/pmc_diff_1k_T10_K10.json profiles/r06_pmc.json; cp gpurun_out/r06/pmc_acker_2k_T20_K15.json profiles/r06_pmc_acker.json; cp gpurun_out/r06/pmc_poly8_5k_T10_K10.json profiles/r06_pmc_poly8.json
fi
stamp "gpu tests"
timeout 600 $py -m pytest tests -m gpu -x -q > $out/gpu_tests_raw.txt 2>&1; echo "rc $?" >> $out/gpu_tests_raw.txt
grep -v "RCCL version\|HIP version\|ROCm version\|Hostname\|Librccl\|amdgpu.ids" $out/gpu_tests_raw.txt > $out/gpu_tests.txt
stamp "smoke"
timeout 200 $py -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "rc $?" >> $out/smoke.log
stamp "driver flags (full command: CPU baseline, parity, extras)"
NPA_BENCH_FULL=$PWD/$out/bench_driver_flags_full.json timeout 600 $py bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_flags.log 2> $out/bench_driver_flags.err
tail -1 $out/bench_driver_flags.log > $out/bench_driver_flags.json
stamp "default line"
NPA_BENCH_FULL=$PWD/$out/bench_full.json timeout 600 $py bench.py > $out/bench.log 2> $out/bench.err
tail -1 $out/bench.log > $out/bench.json
stamp "driver flags again x4 (GPU only), sequential steps"
for i in 2 3 4 5; do timeout 150 $py bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-extras --no-latency > $out/drv_rep$i.json 2>/dev/null; done
timeout 120 $py bench.py --inflight 1 --no-cpu --no-latency --no-extras --steps 32 --warmup 8 > $out/bench_inflight1.json 2>/dev/null
stamp "one rank under torch.distributed.run"
timeout 200 $py -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --no-cpu --no-latency --no-extras 2> $out/torchrun1.err | tail -1 > $out/bench_torchrun1.json
for i in 1 2 3; do timeout 200 $py -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29520 + i)) bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --no-extras 2>> $out/torchrun1.err | tail -1 > $out/bench_torchrun1_driver_flags_$i.json; done
stamp "kernel traces of both commands"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o ks -- $py $GRAFT_REPO_ROOT/bench.py --no-cpu --no-latency --no-extras > /dev/null 2>&1 )
f=$(find /tmp/ks -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats.csv
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks2 -o ks -- $py $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-latency --no-extras > /dev/null 2>&1 )
f=$(find /tmp/ks2 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" $out/kernel_stats_driver_flags.csv
stamp "phases of a QP solve / of a slice wave (s_memtime builds)"
timeout 300 $py tests/tools/qp_phase_cycles.py > $out/qp_phase_cycles_raw.txt 2>&1
grep -v "hipcc\|amdgpu.ids" $out/qp_phase_cycles_raw.txt > $out/qp_phase_cycles.txt
timeout 300 $py tests/tools/select_phase_cycles.py 2>&1 | grep -v "amdgpu.ids" > $out/select_phase_cycles.txt
stamp "other configurations, GPU only (the parity legs ride in the lines above)"
for w in acker_2k_T20_K15 dyna_4k_T10_K10 poly8_5k_T10_K10; do timeout 200 $py bench.py --workload $w --no-cpu --no-latency --no-extras > $out/bench_$w.json 2>/dev/null; done
stamp "64-scene x 12-member parity of the other configurations"
timeout 900 $py tests/tools/parity_wide.py 2> $out/parity_wide.log | tail -1 > $out/parity_wide.json
stamp "done"
python - <<P
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().split("\n")[-1]); print(f.split("/")[-1], d.get("value"), d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("vector_pipe"))
    except Exception as e: print(f, "ERR", str(e)[:80])
P
