"""Where a wave of select_geo_kernel spends its cycles: the -DNPA_SEL_PROF build of the library (s_memtime stamps between
the phases of a wave, summed over all waves of a forward call).  Builds the variant next to the product library
(neupan_amd/libneupan_amd_selprof.so, hipcc here or on the GPU box) and runs forward calls of the BASELINE config on it.

    python tests/tools/select_phase_cycles.py [workload] [scenes]
"""
import ctypes, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neupan_amd import build as nb
import neupan_amd._lib as L

LIB = os.path.join(nb.HERE, "libneupan_amd_selprof.so")
NAMES = ["preamble (vectors to LDS, frame, margins)", "key pass (every point of the slice)", "bound + threshold", "window pass (compaction)",
         "candidate list / overflow decision", "exact keys of a long list (overflow)", "rows of the candidates (MLP on MFMA)", "audit tile",
         "rank + emit", "extraction after an overflow", "audit counters, end"]


def build_prof():
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) > os.path.getmtime(os.path.join(nb.CSRC, f)) for f in os.listdir(nb.CSRC) if not f.endswith(".o")):
        return
    objs = []
    for src in nb.SOURCES:
        if src != "dune.hip":
            objs.append(os.path.join(nb.CSRC, src.replace(".hip", ".o")))
            continue
        o = os.path.join(nb.CSRC, "dune.selprof.o")
        subprocess.check_call([nb.hipcc_path(), *nb.FLAGS, "-DNPA_SEL_PROF=1", f'-DNPA_HIPCC_VERSION="{nb.hipcc_version()}"',
                               "-c", os.path.join(nb.CSRC, src), "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([nb.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB])


if __name__ == "__main__":
    nb.build()
    build_prof()
    if "--build-only" in sys.argv:
        sys.exit(0)
    L.LIB_PATH = LIB
    import torch
    from gpu_helpers import make_gpu_pan
    from helpers import CONFIGS
    from neupan_amd.scenes import make_batch
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = argv[0] if argv else "diff_1k_T10_K10"
    B = int(argv[1]) if len(argv) > 1 else 256
    cfg = CONFIGS[name]
    batch = make_batch(cfg, 0, B)
    args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
    pan = make_gpu_pan(cfg)
    lib = ctypes.CDLL(LIB)
    buf = (ctypes.c_ulonglong * 16)()
    pan.forward_batch(*args); torch.cuda.synchronize()
    assert lib.npa_dbg_sel_prof(buf, 1) == 0
    pan.profile(True)
    pan.forward_batch(*args); torch.cuda.synchronize()
    pr = pan.profile_read()
    assert lib.npa_dbg_sel_prof(buf, 1) == 0
    v = np.array(list(buf), dtype=np.float64)
    waves = v[15]
    tot = v[:11].sum()
    print("%s, %d scenes, one forward call (K = %d) alone on the GPU: %d waves of select_geo_kernel; s_memtime cycles per wave, mean over the waves"
          % (name, B, cfg.iter_num, waves))
    for i, n in enumerate(NAMES):
        print("  %-55s %8.0f  %5.1f %%" % (n, v[i] / waves, 100 * v[i] / tot))
    print("  total %.0f cycles per wave (first stamp -> last: %.0f); the launches of this build: select %.4f ms, QP %.4f ms (HIP events)"
          % (tot / waves, v[14] / waves, pr["select_ms"], pr["nrmp_ms"]))
