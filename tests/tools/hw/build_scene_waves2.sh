#!/bin/bash
# Builds the experiments library with the scene kernel at TWO waves per SIMD (-DNPA_SCENE_WAVES=2: 256 registers, ~54 spilled) --
# the variant that faults on the GPU from the second solve of a scene on -- next to the product library:
#     neupan_amd/_variants/scene2/libneupan_amd.so   (git-ignored objects; inside the repository so that it travels to the GPU box
#                                                     with gpurun, and NOT under tests/: pytest must not find a second tree there)
# Then:  python tests/tools/scene_debug.py --lib=neupan_amd/_variants/scene2/libneupan_amd.so --k=2 --b=96
#        bash tests/tools/hw/scene_waves2_gdb.sh     (the same under rocgdb: faulting wave, pc, disassembly, registers)
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../../.." && pwd)
out="$root/neupan_amd/_variants/scene2"; mkdir -p "$out"
ver=$(hipcc --version | sed -n 's/^HIP version: *//p')
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wno-unused-result -Wno-unused-value -DNPA_EXPERIMENTS -DNPA_SCENE_WAVES=${NPA_SCENE_WAVES:-2} ${NPA_EXTRA_FLAGS:-}"
objs=""
for s in dune nrmp_qp frontend dune_labels c_api serve_group aset_reduce pan_scene; do
  hipcc $flags -DNPA_HIPCC_VERSION="\"$ver\"" -c "$root/neupan_amd/csrc/$s.hip" -o "$out/$s.o" &
  objs="$objs $out/$s.o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$out/libneupan_amd.so"
ls -l "$out/libneupan_amd.so"
