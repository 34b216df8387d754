"""Parity evidence per BASELINE config: HIP per-iteration controls vs an ensemble of oracle runs (tests/parity_tools.py).

    python tests/tools/parity_ensemble.py cpu  <workload> <scenes> <out.npz> [cores]   # oracle ensemble (no GPU)
    python tests/tools/parity_ensemble.py gpu  <workload> <scenes> <out.npz>           # HIP traces (GPU box)
    python tests/tools/parity_ensemble.py judge <hip.npz> <ens.npz> [report.json]      # verdicts A/B/C + distributions
"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    mode = sys.argv[1]
    if mode == "cpu":
        from parity_tools import run_ensemble
        wl, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
        cores = int(sys.argv[5]) if len(sys.argv) > 5 else (os.cpu_count() or 1)
        base, members, rate, cores = run_ensemble(wl, range(n), cores)
        np.savez_compressed(out, base=base, members=members, rate=rate, cores=cores, workload=wl)
        print(f"{wl}: {n} scenes, {members.shape[1]} members, base runs {rate:.1f} plans/s on {cores} cores")
    elif mode == "gpu":
        import torch
        from gpu_helpers import make_gpu_pan
        from helpers import CONFIGS
        from neupan_amd.scenes import make_batch
        wl, n, out = sys.argv[2], int(sys.argv[3]), sys.argv[4]
        cfg = CONFIGS[wl]
        pan = make_gpu_pan(cfg)
        batch = make_batch(cfg, 0, n)
        o = pan.forward_batch_trace(batch["nom_s"], batch["nom_u"], batch["ref_s"], batch["ref_us"], batch["points"],
                                    batch["velocities"])
        torch.cuda.synchronize()
        info = pan.last_qp_info()
        np.savez_compressed(out, trace_u=o["trace_u"].cpu().numpy(), opt_u=o["opt_u"].cpu().numpy(),
                            iters=o["iters"].cpu().numpy(), qp_info=info, workload=wl)
        print(f"{wl}: {n} scenes traced; last QP merit max {info[:, 1].max():.1e} status!=0: {(info[:, 3] != 0).sum()}")
    elif mode == "judge":
        from parity_tools import judge
        h, e = np.load(sys.argv[2]), np.load(sys.argv[3])
        n = min(h["trace_u"].shape[0], e["base"].shape[0])
        rep, hip, sp = judge(h["trace_u"][:n], e["base"][:n], e["members"][:n])
        rep["workload"] = str(e["workload"])
        print(json.dumps(rep, indent=1))
        if len(sys.argv) > 4:
            json.dump(rep, open(sys.argv[4], "w"), indent=1)
    else:
        raise SystemExit(__doc__)


if __name__ == "__main__":
    main()
