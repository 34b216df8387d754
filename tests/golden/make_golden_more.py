"""More pan_<case>.npz vectors, same recipe as make_golden.py (the reference's PAN.forward, unmodified, CvxpyLayer
call substituted by the oracle QP): decimation inside the loop, omni + moving points, a car-like robot in reverse
gear, the trapezoid robot with moving points.  Kept apart so that the first set of fixtures stays byte-identical.

    python tests/golden/make_golden_more.py          (build container only)
"""
import numpy as np

from make_golden import CKPT, CONFIGS, make_scene, pan_case  # noqa: F401  (imports the reference under stubs)

c2 = CONFIGS["diff_1k_T10_K10"]; dy = CONFIGS["dyna_4k_T10_K10"]; ak = CONFIGS["acker_2k_T20_K15"]
poly_robot = dict(kinematics="diff", vertices=[[-0.8, -1.0], [-1.8, 1.0], [1.8, 1.0], [0.8, -1.0]],
                  max_speed=[8, 3], max_acce=[8, 3], length=None, width=None)
omni_robot = dict(kinematics="omni", length=1.6, width=2.0, max_speed=[8, 6.28], max_acce=[3, 3])

if __name__ == "__main__":
    pan_case("decimate_n500_k3", c2, 13, 500, iter_num=3, dune_max_num=100)
    pan_case("omni_dyna_n80_k3", dy, 14, 80, iter_num=3, dune_max_num=80, robot=omni_robot)
    rev = next(b for b in range(100) if make_scene(ak, b, 10)["ref_us"][0] < 0)      # a reverse-gear scene
    print("reverse-gear scene", rev)
    pan_case(f"acker_reverse_n150_k4", ak, rev, 150, iter_num=4, dune_max_num=150)
    pan_case("polygon_dyna_n100_k3", dy, 15, 100, iter_num=3, dune_max_num=100, robot=poly_robot,
             dune_checkpoint=CKPT["polygon_robot"])
