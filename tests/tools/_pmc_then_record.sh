#!/bin/bash
mkdir -p gpurun_out/r06
export TMPDIR=/tmp
for w in diff_1k_T10_K10 acker_2k_T20_K15 poly8_5k_T10_K10; do ( cd /tmp; timeout 450 python $GRAFT_REPO_ROOT/tests/tools/pmc_collect.py $w > $GRAFT_REPO_ROOT/gpurun_out/r06/pmc_$w.log 2>&1 ); echo "$w rc $?"; done
cp gpurun_out/r06/pmc_diff_1k_T10_K10.json profiles/r06_pmc.json; cp gpurun_out/r06/pmc_acker_2k_T20_K15.json profiles/r06_pmc_acker.json; cp gpurun_out/r06/pmc_poly8_5k_T10_K10.json profiles/r06_pmc_poly8.json
bash tests/tools/record_round6.sh
