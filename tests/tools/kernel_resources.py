"""Register / scratch / LDS use of every kernel in the built library, read from the code objects themselves.

    python tests/tools/kernel_resources.py [pattern]

libneupan_amd.so carries one offload bundle per source file in its .hip_fatbin section; each bundle holds a gfx950 code
object whose note section (AMDGPU metadata) lists, per kernel, the allocated VGPRs / SGPRs, what was spilled and the
scratch (private segment) bytes per lane.  tests/test_abi.py asserts on these: a spill in the selection or the QP kernel
is a performance bug (scratch traffic on the hot path) and, on this toolchain, was once a correctness bug.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.path.join(ROOT, "neupan_amd", "libneupan_amd.so")


def tools_available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf"))


def kernel_resources(lib=LIB):
    """{demangled-ish kernel name: dict(vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch, lds, name)} for every kernel."""
    out = {}
    with tempfile.TemporaryDirectory() as d:
        fat = os.path.join(d, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(d, "x")],
                              stderr=subprocess.DEVNULL)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            part = os.path.join(d, f"b{i}.bin")
            open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(d, f"b{i}.co")
            r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                                "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.PIPE)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], stdout=subprocess.PIPE, text=True).stdout
            cur = None
            for line in notes.splitlines():
                m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'\"")
                if k == "agpr_count" and line.lstrip().startswith("-"):
                    cur = {}                      # first key of a kernel record (keys are sorted alphabetically)
                if cur is None:
                    continue
                cur[k] = v
                if k == "wavefront_size":         # last key of a record
                    if "name" in cur:
                        out[cur["name"]] = dict(name=cur["name"], vgpr=int(cur.get("vgpr_count", 0)), agpr=int(cur.get("agpr_count", 0)),
                                                sgpr=int(cur.get("sgpr_count", 0)), vgpr_spill=int(cur.get("vgpr_spill_count", 0)),
                                                sgpr_spill=int(cur.get("sgpr_spill_count", 0)),
                                                scratch=int(cur.get("private_segment_fixed_size", 0)),
                                                lds=int(cur.get("group_segment_fixed_size", 0)))
                    cur = None
    return out


def _code_objects(lib, d):
    """Unbundle every gfx950 code object of the library into directory d; yields their paths."""
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", lib, os.path.join(d, "x")],
                          stderr=subprocess.DEVNULL)
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for i, s in enumerate(starts):
        part = os.path.join(d, f"h{i}.bin")
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(d, f"h{i}.co")
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", f"--input={part}",
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], stderr=subprocess.PIPE)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            yield co


def kernel_isa_hashes(lib=LIB):
    """{mangled kernel name: sha256[:16] of the kernel's MACHINE CODE (the bytes of its function symbol in .text) and of its
    64-byte kernel descriptor (register / LDS allocation, <name>.kd)}.  This is what a counter record is tied to: the counters
    were measured on these bytes, whatever the source files around them look like (a comment, another kernel added to the
    file or a refactoring that compiles to the same code leave them valid; another compiler or another register allocation
    do not)."""
    import hashlib
    out = {}
    with tempfile.TemporaryDirectory() as d:
        for co in _code_objects(lib, d):
            raw = open(co, "rb").read()
            secs = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-S", "--wide", co], stdout=subprocess.PIPE, text=True).stdout
            sec = {}
            for line in secs.splitlines():
                m = re.match(r"\s*\[\s*(\d+)\]\s+(\S+)\s+\S+\s+([0-9a-f]+)\s+([0-9a-f]+)\s+([0-9a-f]+)", line)
                if m:
                    sec[int(m.group(1))] = (m.group(2), int(m.group(3), 16), int(m.group(4), 16))      # name, address, file offset
            syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", co], stdout=subprocess.PIPE, text=True).stdout
            table = {}
            for line in syms.splitlines():
                f = line.split()
                if len(f) >= 8 and f[0].rstrip(":").isdigit() and f[6].isdigit():
                    table[f[7]] = (int(f[1], 16), int(f[2]), f[3], int(f[6]))                             # value, size, type, section index
            for name, (val, size, typ, shndx) in table.items():
                if typ != "FUNC" or size == 0 or (name + ".kd") not in table or shndx not in sec:
                    continue
                _, addr, off = sec[shndx]
                h = hashlib.sha256(raw[off + val - addr: off + val - addr + size])
                kv, ks, _, kx = table[name + ".kd"]
                if kx in sec:
                    _, ka, ko = sec[kx]
                    kd = bytearray(raw[ko + kv - ka: ko + kv - ka + ks])
                    # bytes 16..23 of the descriptor = kernel_code_entry_byte_offset, the distance from the descriptor to the
                    # code: it moves when ANOTHER kernel of the file grows, with not one instruction of this one changed
                    # (round 5: two kernels added next to the QP kernel, its assembly identical line by line, the hash moved)
                    kd[16:24] = b"\0" * 8
                    h.update(bytes(kd))
                out[name] = h.hexdigest()[:16]
    return out


# ---- instructions that can only execute with EXEC == 0 -----------------------------------------------------------------------
# Root cause of the "register-starved build" hazard of rounds 2 - 4 (tests/tools/hw/README.md): in a build
# that spills, hipcc 7.2's register allocator places live-range split copies and spill stores at the top of the block that
# FOLLOWS a divergent loop -- in front of the s_or_b64 that gives EXEC back.  On both ways into that block (the loop's exit: the
# s_cbranch_execnz back edge not taken; the s_cbranch_execz that skips the loop) EXEC is zero, so the copies and stores do
# nothing, and the reloads / restores behind the region bring back whatever the register or the scratch slot held before.  In
# round 4's two-waves-per-SIMD scene kernel that was `v_mov_b32 v142, v136` (a loop-invariant row offset parked while v136 served
# as a temporary): the restore handed every later hinge-row load the constant v142 held (0x19f4ec90 x 4 bytes past the buffer).
# The check below finds such instructions in any build: a must-analysis over the kernel's control flow (EXEC is known to be
# zero on the fall-through edge of s_cbranch_execnz and on the taken edge of s_cbranch_execz, until an instruction writes EXEC);
# a vector ALU / memory instruction reached ONLY with EXEC == 0 is lost work the compiler cannot have meant.
_LANE = ("v_readlane", "v_writelane", "v_readfirstlane")


def _parse_disassembly(lines):
    ins = []
    for l in lines:
        m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):\s*[0-9A-Fa-f ]+(?:<.*\+0x([0-9a-f]+)>)?", l)
        if m:
            ins.append((m.group(1), m.group(2), int(m.group(3), 16), int(m.group(4), 16) if m.group(4) else None))
    return ins


def exec_zero_dead(ins):
    """[(offset in the kernel, instruction)] of the vector instructions of one kernel that are reached only with EXEC == 0.
    ins: (mnemonic, operands, address, branch target offset or None) per instruction, in address order."""
    n = len(ins)
    if not n:
        return []
    base = ins[0][2]
    idx = {a: i for i, (_, _, a, _) in enumerate(ins)}
    preds = [[] for _ in range(n)]
    for j, (mn, ops, a, t) in enumerate(ins):
        if (mn.startswith("s_cbranch") or mn == "s_branch") and t is not None and base + t in idx:
            preds[idx[base + t]].append(("br", j))
        if j + 1 < n and mn not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            preds[j + 1].append(("ft", j))
    st = [True] * n
    st[0] = False

    def out(kind, j):
        mn, ops = ins[j][0], ins[j][1]
        if kind == "br":
            return True if mn == "s_cbranch_execz" else (False if mn == "s_cbranch_execnz" else st[j])
        if mn == "s_cbranch_execnz":
            return True
        if mn == "s_cbranch_execz" or "saveexec" in mn or (mn.startswith("s_") and re.match(r"\s*exec\b", ops)):
            return False
        return st[j]
    changed = True
    while changed:
        changed = False
        for i in range(1, n):
            v = all(out(k, j) for k, j in preds[i]) if preds[i] else False
            if v != st[i]:
                st[i], changed = v, True
    return [("%x" % (a - base), (mn + " " + ops).strip()) for i, (mn, ops, a, t) in enumerate(ins)
            if st[i] and ((mn.startswith("v_") and not mn.startswith(_LANE)) or mn.startswith(("scratch_", "global_", "buffer_", "flat_", "ds_")))]


def lost_instructions(lib=LIB):
    """{kernel symbol: [(offset, instruction)]} over every kernel of the library that has instructions reachable only with EXEC == 0."""
    hits = {}
    with tempfile.TemporaryDirectory() as d:
        for co in _code_objects(lib, d):
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True).stdout
            cur, block = None, []
            for l in text.splitlines() + ["0 <end>:"]:
                m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
                if m:
                    if cur:
                        bad = exec_zero_dead(_parse_disassembly(block))
                        if bad:
                            hits[cur] = bad
                    cur, block = m.group(1), []
                else:
                    block.append(l)
    return hits


# ---- wait states in front of the hand-written DPP instructions ---------------------------------------------------------------
# nrmp_qp_device.h issues v_fmac_f64_dpp (row_newbcast) from asm statements: the compiler schedules around them but does not see
# what they are, so the two hazards of such an instruction are the SOURCE's business (the s_nop inside the statements) -- and
# a copy the register allocator may put right in front of a statement is nobody's.  Checked here on the machine code:
#   a VGPR written by a vector instruction is not read through DPP within the next 2 wait states,
#   a VGPR written by a DPP instruction of ours is not read by v_readlane within the next 1 wait state
# (a wait state = one instruction issued, s_nop N = N + 1; a branch target in between ends the window: conservative = clean).
def _vregs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def dpp_hazards(ins):
    """[(offset, instruction, what)] for one kernel; ins as _parse_disassembly returns them."""
    if not ins:
        return []
    base = ins[0][2]
    bad = []
    for i, (mn, ops, a, _) in enumerate(ins):
        is_dpp = mn.endswith("_dpp") or "row_newbcast" in ops or "row_shr" in ops or "row_shl" in ops or "quad_perm" in ops or "row_bcast" in ops or "row_mirror" in ops or "row_half_mirror" in ops
        is_rl = mn.startswith("v_readlane")
        if not (is_dpp or is_rl):
            continue
        toks = [t for t in re.split(r",\s*", ops.split(" row_")[0].split(" quad_perm")[0])]
        if len(toks) < 2:
            continue
        src = _vregs(toks[1])                       # DPP applies to src0; v_readlane reads its vsrc0
        need = 2 if is_dpp else 1
        ws, j = 0, i - 1
        while j >= 0 and ws < need:
            pm, po = ins[j][0], ins[j][1]
            if pm == "s_nop":
                ws += int(po.strip() or 0, 0) + 1
                j -= 1
                continue
            if pm.startswith("v_") and not pm.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                wrote = _vregs(re.split(r",\s*", po)[0]) if po else set()
                if is_rl and not (pm.endswith("_dpp") and "f64" in pm):
                    wrote = set()                   # (the compiler's own instructions in front of a v_readlane: its business)
                if wrote & src:
                    bad.append(("%x" % (a - base), (mn + " " + ops).strip(), f"{pm} {po.strip()} only {ws} wait state(s) ahead"))
                    break
            ws += 1
            j -= 1
    return bad


def dpp_hazard_report(lib=LIB):
    """{kernel symbol: hazards} over every kernel of the library that contains a double-precision DPP instruction."""
    hits = {}
    with tempfile.TemporaryDirectory() as d:
        for co in _code_objects(lib, d):
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE, text=True).stdout
            cur, block = None, []
            for l in text.splitlines() + ["0 <end>:"]:
                m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
                if m:
                    if cur and any("v_fmac_f64_dpp" in x for x in block):
                        hits[cur] = dpp_hazards(_parse_disassembly(block))
                    cur, block = m.group(1), []
                else:
                    block.append(l)
    return hits


def demangle(name):
    try:
        return subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
    except Exception:
        return name


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--dpp":          # python tests/tools/kernel_resources.py --dpp [library ...]
        for lib in sys.argv[2:] or [LIB]:
            for k, v in dpp_hazard_report(lib).items():
                print(demangle(k)[:90], "hazards:", len(v))
                for x in v[:10]:
                    print("       ", x)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "--lost":         # python tests/tools/kernel_resources.py --lost [library ...]
        for lib in sys.argv[2:] or [LIB]:
            h = lost_instructions(lib)
            print(lib, "-- kernels with vector instructions that can only execute under EXEC = 0:", len(h))
            for k, v in h.items():
                print("  ", demangle(k)[:80], len(v))
                for x in v[:10]:
                    print("       ", x)
        sys.exit(0)
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    res = kernel_resources()
    print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch lds")
    for n, r in sorted(res.items()):
        dn = demangle(n)
        if pat and pat not in dn:
            continue
        print(f"{dn[:70]:70s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch']:7d} {r['lds']:5d}")
