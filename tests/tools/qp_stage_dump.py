"""The fp64 solutions of the QP stage on seeded scenes of a workload, written to an .npz -- run once as is (the register-resident
instantiation of the workload's (T, M)) and once under NPA_QP_GENERIC=1 (the generic LDS instantiation: no DPP broadcasts, no
register rows): tests/test_gpu_parity.py::test_register_resident_qp_equals_the_generic_instantiation compares the two.

    python tests/tools/qp_stage_dump.py <workload> <scenes> <out.npz>
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    from helpers import CONFIGS
    from gpu_helpers import make_gpu_pan
    from neupan_amd.scenes import make_batch
    wl, n, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    cfg = CONFIGS[wl]
    pan = make_gpu_pan(cfg)
    b = make_batch(cfg, 0, n)
    t = {k: torch.from_numpy(b[k]).cuda() for k in ("nom_s", "nom_u", "ref_s", "ref_us", "points")}
    vel = torch.from_numpy(b["velocities"]).cuda() if b.get("velocities") is not None else None
    st = pan.dune_stage(t["nom_s"], t["points"], vel)
    r = pan.nrmp_stage(t["nom_s"], t["nom_u"], t["ref_s"], t["ref_us"], st)
    np.savez(out, x64=r["x64"].cpu().numpy(), info=r["info"].cpu().numpy(), opt_u=r["opt_u"].cpu().numpy(),
             mu=st["mu"].cpu().numpy(), count=st["count"].cpu().numpy())


if __name__ == "__main__":
    main()
