#!/bin/bash
# Round 6: the status word of every QP over 32 768 further scenes of six workloads (scenes 10 240 .. 43 007; ~2 M QPs, two GPU-minutes):
#     gpurun --timeout 2400 -- 'bash tests/tools/qp_status_sweep.sh'     -> gpurun_out/r06scan3/scan_<workload>.txt, summarised in profiles/r06_qp_robustness.txt
out=gpurun_out/r06scan3; mkdir -p $out
for w in diff_1k_T10_K10 dyna_4k_T10_K10 poly8_5k_T10_K10 polygon_5k_T10_K10 acker_2k_T20_K15 uniform_1k_T10_K10; do
  for first in 10240 14336 18432 22528 26624 30720 34816 38912; do
    timeout 300 python tests/tools/qp_status_scan.py $w 4096 $first 2>&1 | grep -v amdgpu >> $out/scan_$w.txt
  done
  grep -c "scenes from" $out/scan_$w.txt; grep "bad: [1-9]\|   scene" $out/scan_$w.txt
done
