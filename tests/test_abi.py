"""CPU tests of the drop-in boundary: the shared library builds/loads here (hipcc cross-compiles
without a GPU) and exports every symbol include/neupan_amd.h declares; struct layouts agree;
the host-side mirror refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from neupan_amd import build
    build.build(force=False, verbose=False)
    from neupan_amd import _lib
    return _lib.load()


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "neupan_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(npa_[a-z_]+)\s*\(", txt)))


def test_header_symbols_are_exported_and_bound(lib):
    from neupan_amd import _lib
    names = declared_symbols()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/neupan_amd.h but not exported"
        assert n in _lib.SYMBOLS, f"{n} has no ctypes prototype in neupan_amd/_lib.py"
    assert b"gfx950" in lib.npa_version()


def test_config_struct_layout_matches_header():
    from neupan_amd._lib import NpaConfig, NpaDuneWeights, NPA_MAX_E
    # 6 int32 + float (+pad) -> 32 bytes, 8 doubles, 7 floats, G[8][2], h[8]
    assert C.sizeof(NpaConfig) == 32 + 8 * 8 + 7 * 4 + NPA_MAX_E * 2 * 4 + NPA_MAX_E * 4 + 4
    assert NpaConfig.step_time.offset == 32 and NpaConfig.q_s.offset == 96
    assert C.sizeof(NpaDuneWeights) == 18 * 8
    hdr = open(os.path.join(ROOT, "include", "neupan_amd.h")).read()
    for name, val in (("NPA_MAX_T", 21), ("NPA_MAX_M", 32), ("NPA_MAX_E", 8)):
        assert re.search(rf"#define {name} {val}\b", hdr)


def test_forward_call_struct_layout_matches_header(tmp_path):
    """npa_forward_call as gcc lays the header's struct out against the ctypes mirror: size and every field offset."""
    import shutil
    import subprocess
    from neupan_amd._lib import NpaForwardCall
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    names = [f[0] for f in NpaForwardCall._fields_]
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "neupan_amd.h"\nint main(void) {\n  printf("%zu", sizeof(npa_forward_call));\n' +
                   "".join(f'  printf(" %zu", offsetof(npa_forward_call, {n}));\n' for n in names) + "  return 0;\n}\n")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[0] == C.sizeof(NpaForwardCall)
    assert got[1:] == [getattr(NpaForwardCall, n).offset for n in names]


def test_group_call_argument_validation_without_gpu(lib):
    from neupan_amd._lib import NpaForwardCall
    arr = (NpaForwardCall * 2)()
    assert lib.npa_forward_batch_group(0, arr, 0) == -1 and lib.npa_forward_batch_group(2, None, 0) == -1
    assert lib.npa_forward_batch_group(2, arr, 0) == -1            # null handles
    arr[0].h = arr[1].h = 0x1000
    arr[0].iter_num = arr[1].iter_num = 1
    assert lib.npa_forward_batch_group(2, arr, 0) == -1            # the same handle twice (refused before anything is touched)
    arr[1].h = 0x2000
    arr[1].iter_num = 0
    assert lib.npa_forward_batch_group(2, arr, 0) == -1            # iter_num < 1


def test_argument_validation_without_gpu(lib):
    from neupan_amd._lib import NpaConfig
    h = C.c_void_p()
    assert lib.npa_create(None, None, C.byref(h)) == -1            # NPA_E_ARG
    cfg = NpaConfig()
    cfg.receding, cfg.iter_num, cfg.nrmp_max_num, cfg.dune_max_num, cfg.edge_num = 64, 1, 10, 100, 4
    assert lib.npa_create(C.byref(cfg), None, C.byref(h)) == -3    # NPA_E_UNSUPPORTED (T > NPA_MAX_T)
    assert b"receding" in lib.npa_last_error()
    assert lib.npa_workspace_bytes(None, 4) == 0 and lib.npa_destroy(None) == 0


def test_qp_scene_block_fits_the_residency_the_design_counts_on(lib):
    """LDS per scene of the QP kernel (the launcher's own size function): 8 scenes per CU at (T, M) = (10, 10) -- the
    block is what limits how many solves a CU holds --, 4 at T = 20 (one wave per SIMD; round 5's 51 KB block
    held 3: a quarter of the SIMDs without a QP wave), and the largest generic problem still fits one CU."""
    import ctypes as C
    f = lib.npa_qp_shmem_bytes_path
    f.restype, f.argtypes = C.c_size_t, [C.c_int, C.c_int, C.c_int]
    cu = 160 * 1024
    assert f(10, 10, 1) <= cu // 8
    assert f(20, 10, 1) <= cu // 4
    assert f(21, 32, 0) <= cu and f(10, 0, 0) <= cu // 6
    assert f(10, 10, 0) > f(10, 10, 1)              # the generic instantiation keeps every row array in LDS


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from neupan_amd._lib import NeupanAmdError
    from neupan_amd.pan import PAN
    from neupan_amd.robot import Robot
    rb = Robot(10, 0.1, kinematics="diff", length=1.6, width=2.0, max_speed=[8, 1], max_acce=[8, 3])
    with pytest.raises(NeupanAmdError):
        PAN(10, 0.1, rb, dune_checkpoint=os.path.join(ROOT, "tests/golden/checkpoints/diff_robot_default_model_5000.pth"))


def test_robot_mirror_matches_reference_geometry():
    from helpers import golden
    from neupan_amd.robot import Robot, halfplanes_from_vertices
    geo = golden("geometry")
    rb = Robot(10, 0.1, kinematics="acker", length=4.6, width=1.6, wheelbase=3, max_speed=[8, 3], max_acce=[8, 0.5])
    np.testing.assert_array_equal(rb.G, geo["acker_G"]); np.testing.assert_array_equal(rb.h, geo["acker_h"])
    assert rb.speed_bound[1, 0] == 1.57 and abs(rb.acce_bound[1, 0] - 0.05) < 1e-15      # robot.py:63-69
    G, h = halfplanes_from_vertices(np.array([[-0.8, -1.0], [-1.8, 1.0], [1.8, 1.0], [0.8, -1.0]]).T)  # CW input
    np.testing.assert_array_equal(G, geo["polygon_G"]); np.testing.assert_array_equal(h, geo["polygon_h"])
    with pytest.raises(ValueError):
        halfplanes_from_vertices(np.array([[0, 0], [2, 0], [1, 0.2], [2, 2], [0, 2]]).T)   # non-convex
    with pytest.raises(ValueError):
        Robot(10, 0.1)                                                                      # robot.py:46-47


def test_hot_kernels_have_no_spills_and_no_scratch():
    """Register / scratch use of the hot kernels, read from the code objects inside the built library (AMDGPU metadata
    notes; tests/tools/kernel_resources.py).  A VGPR spill in the selection kernel was 7 MB of scratch traffic per launch
    (round-2 PMC), and register-starved builds of the QP kernel once came out of this compiler wrong: both
    are build properties, so they are checked on the build."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import kernel_resources as kr
    if not kr.tools_available():
        pytest.skip("ROCm LLVM tools not installed")
    res = kr.kernel_resources()
    sel = {n: r for n, r in res.items() if "select_geo_kernel" in n or "select_geo_group_kernel" in n}
    # E = 3 .. 8, plus the bf16 tiers of the rows and of the keys at E = 4 and 8; the merged-launch form (blockIdx.y = the call) for
    # E = 4 and 8, exact and with bf16 keys
    assert len(sel) == 10 + 4, sorted(sel)
    for n, r in sel.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (n, r)
        wide = "ILi7E" in n or "ILi8E" in n                     # (E > 6 is built for three waves per SIMD)
        assert r["vgpr"] + r["agpr"] <= (168 if wide else 128), (n, r)      # four waves per SIMD
    # EVERY QP instantiation of the product build, no exemptions: the experiments that spilled (the active-set launch, the
    # first form of the geometric selection) are not in it (NPA_EXPERIMENTS build only)
    qp = {n: r for n, r in res.items() if "nrmp_qp_kernel" in n or "nrmp_qp_group_kernel" in n}
    assert len(qp) >= 6, sorted(qp)
    for n, r in qp.items():
        assert r["vgpr_spill"] == 0 and r["scratch"] == 0, (n, r)
        assert r["vgpr"] + r["agpr"] <= 256, (n, r)             # two waves per SIMD
    from conftest import experiments_built
    if not experiments_built():
        assert not any("aset" in n or "select_kernelILi4ELb1" in n for n in res), "experiment kernels in the product build"
        assert os.path.getsize(kr.LIB) < 2 * 1024 * 1024, os.path.getsize(kr.LIB)

def test_no_kernel_has_instructions_that_only_run_with_exec_zero():
    """The root cause of the register-starved-build hazard (rounds 2 - 4: corrupted loop scalars in spilled QP builds, the
    faulting two-waves-per-SIMD scene kernel), found in round 5 under rocgdb: hipcc 7.2's register allocator puts live-range
    split copies and spill stores at the top of the block behind a divergent loop, in FRONT of the s_or_b64 that restores
    EXEC -- they execute with EXEC == 0, do nothing, and the restores behind the region bring back stale values
    (tests/tools/hw/README.md, profiles/r05_spilled_build_fault.txt).  Whether a build has such instructions is a static
    property of its machine code (tests/tools/kernel_resources.exec_zero_dead: a must-analysis of EXEC == 0 over the control
    flow): the product library must have none, in ANY kernel.  (The analysis is also unit-tested on a hand-written block.)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import kernel_resources as kr
    # the analysis itself: a divergent loop, then a copy and a spill store in front of the EXEC restore -- and a clean variant
    def block(after_loop):
        text = ["v_mov_b32_e32 v1, v2", "s_and_saveexec_b64 s[6:7], vcc", "s_cbranch_execz L_join",
                "L_loop:", "ds_write_b64 v6, v[8:9]", "s_andn2_b64 exec, exec, s[20:21]", "s_cbranch_execnz L_loop",
                "L_join:"] + after_loop + ["v_add_f32_e32 v3, v1, v1", "s_endpgm"]
        labels, ins, a = {}, [], 0x100
        for t in text:
            if t.endswith(":"):
                labels[t[:-1]] = a
            else:
                a += 4
        a = 0x100
        for t in text:
            if t.endswith(":"):
                continue
            mn, _, ops = t.partition(" ")
            tgt = labels[ops] - 0x100 if ops in labels else None
            ins.append((mn, ops, a, tgt))
            a += 4
        return ins
    bad = kr.exec_zero_dead(block(["v_writelane_b32 v254, s24, 21", "v_mov_b32_e32 v142, v136", "scratch_store_dwordx2 off, v[154:155], off offset:32",
                                   "s_or_b64 exec, exec, s[6:7]"]))
    assert [b[1].split()[0] for b in bad] == ["v_mov_b32_e32", "scratch_store_dwordx2"], bad          # (v_writelane ignores EXEC: fine)
    assert kr.exec_zero_dead(block(["s_or_b64 exec, exec, s[6:7]", "v_mov_b32_e32 v142, v136"])) == []
    if not kr.tools_available():
        pytest.skip("ROCm LLVM tools not installed")
    lost = kr.lost_instructions()
    assert lost == {}, {k: v[:4] for k, v in lost.items()}


def test_hand_written_dpp_instructions_have_their_wait_states():
    """Round 6: the T = 10 QP kernels broadcast a lane's double inside the multiply-add itself (v_fmac_f64_dpp ... row_newbcast, asm
    statements of nrmp_qp_device.h) -- the compiler schedules around such a statement but does not know its hazards: a VGPR written
    by a vector instruction may not be read through DPP within 2 wait states, one written by the statement not by v_readlane
    within 1.  The statements carry their own s_nop; what the register allocator puts around them is checked here on the MACHINE
    CODE of the built library (tests/tools/kernel_resources.dpp_hazards; the instruction itself is checked on the device:
    tests/tools/hw/dpp_f64_check.hip, bitwise fma in all 64 lanes for every row_newbcast lane)."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
    import kernel_resources as kr
    mk = lambda seq: [(m, o, 4 * i, None) for i, (m, o) in enumerate(seq)]
    dpp = ("v_fmac_f64_dpp", "v[8:9], v[4:5], -v[4:5] row_newbcast:3 row_mask:0xf bank_mask:0xf")
    assert len(kr.dpp_hazards(mk([("v_mul_f64", "v[4:5], v[0:1], v[2:3]"), ("s_nop", "0"), dpp]))) == 1
    assert kr.dpp_hazards(mk([("v_mul_f64", "v[4:5], v[0:1], v[2:3]"), ("s_nop", "1"), dpp])) == []
    assert kr.dpp_hazards(mk([("v_mul_f64", "v[4:5], v[0:1], v[2:3]"), ("v_add_f64", "v[0:1], v[0:1], v[2:3]"), ("s_nop", "0"), dpp])) == []
    assert len(kr.dpp_hazards(mk([dpp, ("v_readlane_b32", "s3, v9, 4")]))) == 1
    assert kr.dpp_hazards(mk([dpp, ("s_nop", "0"), ("v_readlane_b32", "s3, v9, 4")])) == []
    if not kr.tools_available():
        pytest.skip("ROCm LLVM tools not installed")
    rep = kr.dpp_hazard_report()
    assert len(rep) >= 2, list(rep)                     # the stand-alone and the group instantiation of T = 10
    assert all(v == [] for v in rep.values()), {k: v[:4] for k, v in rep.items() if v}


def test_build_refuses_an_unvalidated_compiler(monkeypatch):
    """neupan_amd.build fails (not warns) on a hipcc other than the validated one unless NPA_ALLOW_UNVALIDATED=1."""
    from neupan_amd import build as b
    calls = []
    monkeypatch.setattr(b, "hipcc_version", lambda: "9.9.99999-test")
    monkeypatch.setattr(b.subprocess, "check_call", lambda cmd, *a, **k: calls.append(cmd))

    class FakeProc:                                   # (the translation units are compiled side by side: Popen + wait)
        def __init__(self, cmd, *a, **k):
            calls.append(cmd)

        def wait(self):
            return 0
    monkeypatch.setattr(b.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("NPA_ALLOW_UNVALIDATED", raising=False)
    with pytest.raises(b.UnvalidatedCompiler):
        b.build(force=True, verbose=False)
    assert not calls
    monkeypatch.setenv("NPA_ALLOW_UNVALIDATED", "1")
    b.build(force=True, verbose=False)
    assert len(calls) == len(b.SOURCES) + 1          # every source compiled, one link
