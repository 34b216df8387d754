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


def demangle(name):
    try:
        return subprocess.run([os.path.join(LLVM, "llvm-cxxfilt"), name], stdout=subprocess.PIPE, text=True).stdout.strip().split("(")[0]
    except Exception:
        return name


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    res = kernel_resources()
    print(f"{'kernel':70s} vgpr agpr sgpr vspill sspill scratch lds")
    for n, r in sorted(res.items()):
        dn = demangle(n)
        if pat and pat not in dn:
            continue
        print(f"{dn[:70]:70s} {r['vgpr']:4d} {r['agpr']:4d} {r['sgpr']:4d} {r['vgpr_spill']:6d} {r['sgpr_spill']:6d} {r['scratch']:7d} {r['lds']:5d}")
