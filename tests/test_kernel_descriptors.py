"""Properties of the built gfx950 code objects (no GPU needed): registers, scratch and instruction counts of the
kernels, read from libzkstark_amd.so with llvm-readelf / llvm-objdump (tools/kernel_descriptors.py).  These are the
numbers DESIGN.md and bench.py quote; a compiler or source change that moves them fails here."""
import importlib.util
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def kd():
    mod = _load("kernel_descriptors", os.path.join(ROOT, "tools", "kernel_descriptors.py"))
    if not os.path.exists(os.path.join(mod.LLVM, "llvm-readelf")):
        pytest.skip("llvm tools not present")
    from zkstark_amd import _lib
    _lib.load()                                   # builds the library if needed
    return mod


@pytest.fixture(scope="module")
def rows(kd):
    return kd.collect(want_isa=True)


SCRATCH_ALLOWED = ()       # no kernel may use scratch (round 2 had one exception: the field-hash composition, 36 B)


def test_every_kernel_is_described(rows):
    names = [r["demangled"] for r in rows]
    assert len(names) >= 90
    for want in ("merkle_subtree_kernel", "merkle_wg_kernel", "ntt_pass_fast_kernel", "compose_kernel",
                 "fri_fold_kernel", "gather_kernel", "hash_chain_probe_kernel"):
        assert any(want in n for n in names), want


def test_no_kernel_uses_scratch(rows):
    bad = [(r["demangled"], r["scratch"]) for r in rows if r.get("scratch", 0) and not any(a in r["demangled"] for a in SCRATCH_ALLOWED)]
    assert not bad, bad
    for r in rows:                                 # the exceptions reserve the slots and never touch them
        if r.get("scratch", 0):
            assert r["scratch"] <= 64, r["demangled"]


def test_subtree_kernels_register_budget(rows):
    """512 VGPRs per SIMD lane.  SHA-256 subtree kernels: <= 64 per wave (the 40 KiB of LDS per workgroup, not the
    registers, sets their four waves per SIMD); field-hash ones (double precision: the state alone is 32 registers): <= 128,
    i.e. the same four waves per SIMD the LDS allows."""
    sub = [r for r in rows if "merkle_subtree_kernel<" in r["demangled"]]
    assert len(sub) == 16                       # leaf sources Plain / Fold / Compose / ComposeBlock / Interleave / the two batch forms + inner, x 2 hashes
    for r in sub:
        assert r["agpr"] == 0
        is_field = r["demangled"].split(">(")[0].rstrip().endswith("1")
        assert r["vgpr"] <= (128 if is_field else 64), (r["demangled"], r["vgpr"])


def test_latency_kernels_keep_one_workgroup_per_cu_free_of_spills(rows):
    for r in rows:
        if "merkle_wg_kernel<" in r["demangled"]:
            # one wave per SIMD per workgroup and at most a few workgroups per compute unit: what matters is that nothing spills
            # (round 5: the continuation phase keeps four more digests live; 176 registers for SHA-256, 236 for the field hash)
            assert r["vgpr"] <= 256 and r.get("scratch", 0) == 0, r["demangled"]


def _loops_of(kd, substring):
    import tempfile
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for i, elf in enumerate(kd.code_objects(kd.fatbin_bytes())):
            path = os.path.join(td, f"co{i}.elf")
            with open(path, "wb") as f:
                f.write(elf)
            for k in kd.notes(path):
                dm = kd.demangle([k["name"]])[k["name"]]
                if substring in dm:
                    out[dm] = kd.loops(path, k["name"])
    return out


def test_instruction_counts_quoted_by_bench(kd):
    """bench.py HASH_MODEL / kernels.hpp kSha*Ops, kField*Ops against the binary: the chain probe's loop body is one
    inner hash; the subtree kernel's loops are one leaf hash and one inner hash (plus a dozen address instructions)."""
    bench = _load("bench_mod", os.path.join(ROOT, "bench.py"))
    hm = bench.HASH_MODEL
    probes = _loops_of(kd, "hash_chain_probe_kernel")
    sha = [v for k, v in probes.items() if "<0>" in k][0]
    assert len(sha) == 1 and sha[0][2] == hm["sha256"]["probe_ops"], sha
    fh = sorted([v for k, v in probes.items() if "<1>" in k][0], key=lambda t: t[0])
    # loops in address order: the loop over PAIRS of partial rounds (10 trips: rounds 1 .. 20; rounds 0 and 21 are straight-line
    # code around it, csrc/fieldhash_f64.hpp) sits between the two full-round loops (4 trips each), all inside the outer loop
    def dynamic_count(loops):
        loops = sorted(loops, key=lambda t: t[0])
        outer = max(loops, key=lambda t: t[1] - t[0])
        inner = [l for l in loops if l is not outer and l[0] >= outer[0] and l[1] <= outer[1] and l[2] > 50]
        assert len(inner) == 3, loops
        full_a, partial, full_b = inner
        return (outer[2] - sum(l[2] for l in inner)) + 4 * full_a[2] + 10 * partial[2] + 4 * full_b[2]
    dynamic = dynamic_count(fh)
    assert abs(dynamic - hm["field"]["probe_ops"]) <= 0.01 * dynamic, (dynamic, fh)
    # the subtree kernel's two field-hash loops: one leaf hash, one inner hash
    subf = list(_loops_of(kd, "merkle_subtree_kernel<zk::PlainSrc, true, 1>").values())[0]
    big = sorted([l for l in subf if l[2] > 1000], key=lambda t: t[0])
    assert len(big) == 2, subf
    leaf_f = dynamic_count([l for l in subf if l[0] >= big[0][0] and l[1] <= big[0][1]])
    inner_f = dynamic_count([l for l in subf if l[0] >= big[1][0] and l[1] <= big[1][1] and l[2] > 50])
    assert abs(leaf_f - hm["field"]["leaf_ops"]) <= 0.01 * leaf_f and abs(inner_f - hm["field"]["inner_ops"]) <= 0.01 * inner_f, (leaf_f, inner_f)
    sub = _loops_of(kd, "merkle_subtree_kernel<zk::PlainSrc, true, 0>")
    loops = list(sub.values())[0]
    counts = sorted(l[2] for l in loops)
    leaf = [c for c in counts if abs(c - hm["sha256"]["leaf_ops"]) <= 25]
    inn = [c for c in counts if abs(c - hm["sha256"]["inner_ops"]) <= 25]
    assert leaf and inn, counts
    # kernels.hpp carries the same constants
    hpp = open(os.path.join(ROOT, "zkstark_amd", "csrc", "kernels.hpp")).read()
    for key, name in (("leaf_ops", "kShaLeafOps"), ("inner_ops", "kShaInnerOps")):
        assert float(re.search(name + r"\s*=\s*([0-9.]+)", hpp).group(1)) == hm["sha256"][key]
    for key, name in (("leaf_ops", "kFieldLeafOps"), ("inner_ops", "kFieldInnerOps")):
        assert float(re.search(name + r"\s*=\s*([0-9.]+)", hpp).group(1)) == hm["field"][key]


def test_four_lane_sha256_of_the_latency_levels(kd):
    """csrc/sha256_quad.hpp as compiled into merkle_wg_kernel: one hash is 48 x 16 + 16 x 12 + 64 x 11 round instructions,
    544 of them DPP additions (6 / 4 / 3 per round), and the order inside a round leaves no DPP operand closer than two
    instructions behind its producer, so no s_nop sits between the instructions of a round."""
    import tempfile
    text = None
    with tempfile.TemporaryDirectory() as td:
        for i, elf in enumerate(kd.code_objects(kd.fatbin_bytes())):
            path = os.path.join(td, f"co{i}.elf")
            with open(path, "wb") as f:
                f.write(elf)
            dis = kd._disasm(path)
            m = re.search(r"^[0-9a-f]+ <(_ZN2zk16merkle_wg_kernelINS_8PlainSrcELb0ELi0E[^>]*)>:\n(.*?)(?=^[0-9a-f]+ <)", dis, re.S | re.M)
            if m:
                text = m.group(2)
                break
    assert text, "merkle_wg_kernel<PlainSrc, false, 0> not found"
    ops = re.findall(r"^\s*([sv]_[a-z0-9_]+)", text, re.M)
    assert ops.count("v_add_u32_dpp") == 48 * 6 + 16 * 4 + 64 * 3
    # inside a round (between two DPP additions that are at most 3 instructions apart) there is never an s_nop
    idx = [i for i, o in enumerate(ops) if o == "v_add_u32_dpp"]
    for a, b in zip(idx, idx[1:]):
        if b - a <= 3:
            assert "s_nop" not in ops[a:b], ops[a:b + 1]
