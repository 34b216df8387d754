"""Where a QP solve spends its cycles: the -DNPA_QP_PROF build of the library (s_memtime stamps between the phases of an
interior-point iteration, accumulated per scene in qp_info[5..14]).  Builds the variant next to the product library
(neupan_amd/libneupan_amd_prof.so, hipcc here or on the GPU box) and runs one forward call of the BASELINE config on it.

    python tests/tools/qp_phase_cycles.py [workload] [scenes]
"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from neupan_amd import build as nb
import neupan_amd._lib as L

def prof_lib(mode):
    return os.path.join(nb.HERE, "libneupan_amd_prof%d.so" % mode)


def build_prof(mode):
    lib = prof_lib(mode)
    if os.path.exists(lib) and all(os.path.getmtime(lib) > os.path.getmtime(os.path.join(nb.CSRC, f)) for f in os.listdir(nb.CSRC) if not f.endswith(".o")):
        return
    objs = []
    for src in nb.SOURCES:
        if src != "nrmp_qp.hip":                     # the other objects of the product build
            objs.append(os.path.join(nb.CSRC, src.replace(".hip", ".o")))
            continue
        o = os.path.join(nb.CSRC, "nrmp_qp.prof%d.o" % mode)
        subprocess.check_call([nb.hipcc_path(), *nb.FLAGS, "-DNPA_QP_PROF=%d" % mode, f'-DNPA_HIPCC_VERSION="{nb.hipcc_version()}"',
                               "-c", os.path.join(nb.CSRC, src), "-o", o], stderr=subprocess.DEVNULL)
        objs.append(o)
    subprocess.check_call([nb.hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib])


NAMES = {
    1: ["setup (A/B/C, Phi, H, hinge rows, start)", "residuals, per-step sums, merit", "P_t recursion + band of C'DC",
        "rows of K'", "Cholesky + columns of L", "rhs weights + per-step sums (x2)", "Phi' + substitutions (x2)",
        "Phi dx, directions, step length (x2)", "update", "tail (solution, stop test)"],
    2: ["setup", "residual phase: Phi x", "  hinge rows", "  u rows + d rows", "  per-step sums (lane = t)", "  Phi' q + C' lam",
        "  3 wave reductions, merit, best iterate", "-", "rest of the iteration", "tail"],
    3: ["setup", "pass (x2): rhs weights of the rows", "  per-step sums (lane = t)", "  Phi' q + C' w", "  substitutions", "  Phi dx, dx_d",
        "  directions of the rows", "residuals + K' + Cholesky (+ reduction, centring of the pass)", "update", "tail"],
}


if __name__ == "__main__":
    nb.build()
    modes = [int(a[7:]) for a in sys.argv[1:] if a.startswith("--mode=")] or [1, 2, 3]
    for m in modes:
        build_prof(m)
    if "--build-only" in sys.argv:
        sys.exit(0)
    mode = modes[0]
    if len(modes) > 1:                                # one process per variant (the library is loaded once per process)
        for m in modes:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--mode=%d" % m] + [a for a in sys.argv[1:] if not a.startswith("--")])
        sys.exit(0)
    L.LIB_PATH = prof_lib(mode)
    from gpu_helpers import make_gpu_pan
    from helpers import CONFIGS
    from neupan_amd.scenes import make_batch
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    name = argv[0] if argv else "diff_1k_T10_K10"
    B = int(argv[1]) if len(argv) > 1 else 256
    cfg = CONFIGS[name]
    batch = make_batch(cfg, 0, B)
    args = [batch[k] for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")]
    names = NAMES[mode]
    for K in (1, cfg.iter_num):
        pan = make_gpu_pan(cfg, iter_num=K)
        pan.forward_batch(*args)
        info = pan.last_qp_info()
        cyc = info[:, 5:15]
        its = info[:, 4] + 1          # the last pass through the loop ends after the residual phase
        print("mode %d, K=%d: last QP of the call, %d scenes (one launch, alone on the GPU), iterations mean %.1f; s_memtime cycles per scene, mean over scenes"
              % (mode, K, B, info[:, 4].mean()))
        tot = cyc.sum(1).mean()
        for i, n in enumerate(names):
            if n == "-":
                continue
            per_it = "" if i in (0, 9) else "  per iteration %.0f" % (cyc[:, i] / its).mean()
            print("  %-62s %8.0f  %5.1f %%%s" % (n, cyc[:, i].mean(), 100 * cyc[:, i].mean() / tot, per_it))
        print("  total %.0f cycles" % tot)
