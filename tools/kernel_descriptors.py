#!/usr/bin/env python3
"""Kernel descriptors and static instruction counts of the built gfx950 code objects.

    python tools/kernel_descriptors.py [--json out.json] [--isa KERNEL_SUBSTRING]

Reads `.hip_fatbin` of zkstark_amd/libzkstark_amd.so (one clang offload bundle per translation unit),
extracts every gfx950 ELF and reports, per kernel: VGPRs, AGPRs, SGPRs, scratch (`.private_segment_fixed_size`),
static LDS, and the number of VALU / SALU / LDS / VMEM instructions in its body (llvm-objdump).  The
descriptors are what `tests/test_kernel_descriptors.py` pins (no kernel of the timed path may use scratch).
No GPU needed: this is a property of the binary.
"""
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zkstark_amd", "libzkstark_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def fatbin_bytes(lib=LIB):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, out])
        with open(out, "rb") as f:
            return f.read()


def code_objects(blob, arch="gfx950"):
    """Every device ELF for `arch` in the concatenated bundles."""
    elfs = []
    pos = blob.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", blob, p)
            ident = blob[p + 24:p + 24 + idlen].decode()
            p += 24 + idlen
            if arch in ident and size:
                elfs.append(blob[pos + off:pos + off + size])
        pos = blob.find(MAGIC, pos + len(MAGIC))
    return elfs


def demangle(names):
    if not names:
        return {}
    import shutil
    tool = shutil.which("c++filt") or shutil.which("llvm-cxxfilt")
    if not tool:
        return {n: n for n in names}
    out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True).stdout.splitlines()
    return dict(zip(names, out))


def notes(elf_path):
    """[{name, vgpr, agpr, sgpr, scratch, lds, wg}] from the AMDGPU metadata note."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", elf_path], capture_output=True, text=True).stdout
    kernels, cur = [], None
    for line in txt.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)$", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip().strip("'")
        if k == "agpr_count":                       # first key of a kernel record in llvm's emission order
            cur = {"agpr": int(v)}
            kernels.append(cur)
        elif cur is not None:
            if k == "name" and "name" not in cur and v.startswith("_Z"):
                cur["name"] = v
            elif k == "vgpr_count": cur["vgpr"] = int(v)
            elif k == "sgpr_count": cur["sgpr"] = int(v)
            elif k == "private_segment_fixed_size": cur["scratch"] = int(v)
            elif k == "group_segment_fixed_size": cur["lds"] = int(v)
            elif k == "max_flat_workgroup_size": cur["wg"] = int(v)
            elif k == "symbol": cur["symbol"] = v
    for k in kernels:
        if "name" not in k and "symbol" in k:
            k["name"] = k["symbol"].replace(".kd", "")
    return [k for k in kernels if "name" in k]


_DISASM = {}


def _disasm(elf_path):
    """llvm-objdump -d of one code object (cached per content: several queries walk the same text)."""
    with open(elf_path, "rb") as f:
        key = hash(f.read())
    if key not in _DISASM:
        _DISASM[key] = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", elf_path], capture_output=True, text=True).stdout
    return _DISASM[key]


def isa_counts(elf_path):
    """{mangled name: {valu, salu, lds, vmem, total}} from the disassembly."""
    txt = _disasm(elf_path)
    res, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = res.setdefault(m.group(1), {"valu": 0, "salu": 0, "ds": 0, "vmem": 0, "total": 0, "ops": {}})
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b", line)
        if not m or cur is None:
            continue
        op = m.group(1)
        if op.startswith("s_nop") or op == "s_endpgm" or op.startswith("s_code_end"):
            continue
        cur["total"] += 1
        if op.startswith("v_"):
            cur["valu"] += 1
            cur["ops"][op] = cur["ops"].get(op, 0) + 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("ds_"):
            cur["ds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            cur["vmem"] += 1
    return res


def loops(elf_path, mangled):
    """[(start offset, end offset, VALU instructions)] of every backward branch (loop) in one kernel; offsets are
    relative to the kernel's first instruction.  An innermost loop's count is what one trip executes."""
    txt = _disasm(elf_path)
    inside, rows, base = False, [], None
    for line in txt.splitlines():
        m = re.match(r"^([0-9a-f]+) <(\S+)>:", line)
        if m:
            inside = m.group(2) == mangled
            if inside:
                base = int(m.group(1), 16)
            continue
        if not inside:
            continue
        m = re.match(r"^\s+([a-z_0-9]+)\b.*//\s*([0-9A-Fa-f]+):", line)
        if m:
            tgt = re.search(r"<" + re.escape(mangled) + r"\+0x([0-9a-fA-F]+)>", line)
            rows.append((int(m.group(2), 16) - base, m.group(1), int(tgt.group(1), 16) if tgt else None))
    out = []
    for off, op, tgt in rows:
        if (op.startswith("s_cbranch") or op == "s_branch") and tgt is not None and tgt <= off:
            out.append((tgt, off, sum(1 for a, o, _ in rows if tgt <= a <= off and o.startswith("v_"))))
    return out


def collect(lib=LIB, want_isa=False):
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for i, elf in enumerate(code_objects(fatbin_bytes(lib))):
            path = os.path.join(td, f"co{i}.elf")
            with open(path, "wb") as f:
                f.write(elf)
            ks = notes(path)
            counts = isa_counts(path) if want_isa else {}
            for k in ks:
                k.update({kk: vv for kk, vv in counts.get(k["name"], {}).items()})
                rows.append(k)
    dm = demangle([r["name"] for r in rows])
    for r in rows:
        r["demangled"] = dm.get(r["name"], r["name"])
    return rows


def main():
    args = sys.argv[1:]
    want_isa = True
    rows = collect(want_isa=want_isa)
    isa_filter = args[args.index("--isa") + 1] if "--isa" in args else None
    for r in sorted(rows, key=lambda r: r["demangled"]):
        short = r["demangled"]
        if short.endswith(")"):                     # drop the parameter list, keep template arguments like (zk::NttMode)0
            depth, i = 0, len(short) - 1
            while i >= 0:
                depth += short[i] == ")"
                depth -= short[i] == "("
                if depth == 0:
                    break
                i -= 1
            short = short[:i]
        short = short[:110]
        print(f"{short:110s} vgpr {r.get('vgpr', -1):3d} agpr {r.get('agpr', 0):3d} sgpr {r.get('sgpr', -1):3d} scratch {r.get('scratch', -1):4d} "
              f"lds {r.get('lds', -1):6d}  valu {r.get('valu', 0):6d} salu {r.get('salu', 0):5d} ds {r.get('ds', 0):4d} vmem {r.get('vmem', 0):4d}")
        if isa_filter and isa_filter in r["demangled"]:
            for op, n in sorted(r.get("ops", {}).items(), key=lambda kv: -kv[1]):
                print(f"      {op:28s} {n}")
    if "--loops" in args:
        sub = args[args.index("--loops") + 1]
        with tempfile.TemporaryDirectory() as td:
            for i, elf in enumerate(code_objects(fatbin_bytes())):
                path = os.path.join(td, f"co{i}.elf")
                with open(path, "wb") as f:
                    f.write(elf)
                for k in notes(path):
                    dm = demangle([k["name"]])[k["name"]]
                    if sub in dm:
                        print(dm)
                        for a, b, n in loops(path, k["name"]):
                            print(f"   loop 0x{a:x} .. 0x{b:x}: {n} VALU instructions per trip")
    if "--json" in args:
        with open(args[args.index("--json") + 1], "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
