"""The C-ABI library loads, exports every symbol include/zkstark_amd.h declares, and its host-only
entry points (field, channel, verifier, wire format) agree with the oracle.  No GPU, no compute calls."""
import ctypes as C
import hashlib
import json
import os
import re
import struct

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
P = 3221225473


def header_symbols():
    text = open(os.path.join(ROOT, "include", "zkstark_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zk_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported(zk):
    from zkstark_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 45
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/zkstark_amd.h but not exported"
    assert set(_lib.SYMBOLS) == set(syms), set(_lib.SYMBOLS) ^ set(syms)


def test_rust_surface_is_generated_from_the_header():
    """bindings/rust/zkstark_amd_sys.rs is generated from include/zkstark_amd.h (tools/gen_rust_sys.py): the
    committed file is current and declares every function of the header (the Rust drop-in of prover.rs:9)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_sys", os.path.join(ROOT, "tools", "gen_rust_sys.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    text = gen.generate()
    committed = open(gen.OUT).read()
    assert committed == text, "bindings/rust/zkstark_amd_sys.rs is out of date: run python tools/gen_rust_sys.py"
    assert gen.declared_functions(committed) == header_symbols()
    # the wrapper uses only declared functions, and the caller's channel is not discarded
    wrapper = open(os.path.join(ROOT, "bindings", "rust", "prover.rs")).read()
    used = set(re.findall(r"\b(zk_[a-z0-9_]+)\(", wrapper))
    assert used <= set(header_symbols()) and "zk_prove_channel" in used and "(_channel:" not in wrapper and "(channel: Channel)" in wrapper


def test_channel_import_adopts_state_and_data(zk):
    """zk_channel_import: the (state, data) of a Channel kept on the caller's side (channel.rs:6-9)."""
    import ctypes as C
    from zkstark_amd import _lib
    a = zk.Channel()
    a.commit(bytes(range(32)))
    a.get_u32()
    b = zk.Channel()
    _lib.check(_lib.load().zk_channel_import(b._h, a.state, a.data, len(a.data)))
    assert b.state == a.state and b.data == a.data
    assert a.get_u32() == b.get_u32() and a.state == b.state and a.data == b.data


def test_oracle_prefixed_prover_reduces_to_plain(orc):
    """orc_prove_prefixed with an empty prefix is orc_prove; a prefix changes every challenge."""
    base = orc.prove(6, 2, want_vectors=False)
    data, state = orc.prove_prefixed(b"", 6, 2)
    assert data == base.proof and state == base.state
    data2, state2 = orc.prove_prefixed(b"session 7", 6, 2)
    assert data2[:9] == b"session 7" and data2[9:9 + 32] == base.proof[:32]      # same f_eval root, committed after the prefix
    assert data2[9 + 32:9 + 36] != base.proof[32:36] and state2 != base.state      # alpha0 already differs


def test_product_does_not_touch_the_oracle():
    """The shipped package must not import, link or load anything under oracle/."""
    pkg = os.path.join(ROOT, "zkstark_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"
    import subprocess
    out = subprocess.run(["ldd", os.path.join(pkg, "libzkstark_amd.so")], capture_output=True, text=True).stdout
    assert "oracle" not in out


def test_field_matches_oracle(zk, orc):
    f = zk.field
    assert f.generator() == 5 == orc.generator()
    assert f.root_of_unity(10) == orc.gen_of_order_log(10) == f.pow(5, 3145728)     # prover.rs:48
    assert f.root_of_unity(13) == f.pow(5, 393216)                                   # prover.rs:49
    import random
    rnd = random.Random(1)
    for _ in range(2000):
        a, b = rnd.randrange(P), rnd.randrange(P)
        assert f.add(a, b) == orc.add(a, b) == (a + b) % P
        assert f.sub(a, b) == orc.sub(a, b) == (a - b) % P
        assert f.mul(a, b) == orc.mul(a, b) == a * b % P
        assert f.neg(a) == (-a) % P
    for a in (1, 2, 5, P - 1, 123456789):
        assert f.mul(a, f.inv(a)) == 1
    assert f.order(f.root_of_unity(10)) == 1024 and f.order(f.root_of_unity(13)) == 8192   # prover.rs:52-53
    assert f.order(5) == P - 1 and f.order(1) == 1 and f.order(P - 1) == 2
    for a in (2, 3, 7, 12345):
        assert f.pow(a, f.order(a)) == 1
    assert f.from_u32(3235878091) == 3235878091 - P        # field.rs:20-24: reduce raw u32 >= P
    assert f.add(P - 1, P - 1) == P - 2                    # P > 2^31: a + b overflows u32
    # field.rs:165-177 Div, :10-18 From<i32>, :89-94 Rem<u32>
    for _ in range(500):
        a, b, m = rnd.randrange(P), rnd.randrange(1, P), rnd.randrange(1, 2**32)
        q = f.div(a, b)
        assert q == orc.div(a, b) == a * pow(b, P - 2, P) % P and f.mul(q, b) == a
        assert f.rem(a, m) == orc.rem(a, m) == (a % m) % P
        v = rnd.randrange(-2**31, 2**31)
        assert f.from_i32(v) == orc.from_i32(v) == v % P
    assert f.from_i32(-1) == P - 1 and f.from_i32(0) == 0 and f.from_i32(-2**31) == (-2**31) % P and f.from_i32(2**31 - 1) == 2**31 - 1
    assert f.rem(P - 1, P - 1) == 0 and f.rem(5, 7) == 5 and f.div(0, 3) == 0 and f.div(7, 1) == 7
    lib = zk.load()
    assert lib.zk_field_div(1, 0) == 0 and b"division by zero" in lib.zk_last_error()      # the reference panics
    assert lib.zk_field_div(1, P) == 0                                                       # a raw divisor = 0 (mod P)
    assert lib.zk_field_rem(1, 0) == 0 and b"remainder by zero" in lib.zk_last_error()
    with pytest.raises(ZeroDivisionError):
        f.div(1, 0)
    with pytest.raises(ZeroDivisionError):
        f.rem(1, 0)


def test_trace_fibsq(zk, orc):
    a = zk.trace_fibsq(1023)
    assert a[1022] == 2338775057                            # prover.rs:42
    assert (a == orc.trace_fibsq(1023)).all()


def test_channel_matches_hashlib(zk):
    """channel.rs:11-37 with the bincode encodings of SURVEY Appendix B."""
    ch = zk.Channel()
    assert ch.state == bytes(32) and ch.data == b""
    root = bytes(range(32))
    ch.commit(root)
    st = hashlib.sha256(bytes(32) + root).digest()
    assert ch.state == st
    v = ch.get_u32()
    assert v == struct.unpack(">I", st[:4])[0]              # channel.rs:29 big-endian
    st = hashlib.sha256(st + struct.pack("<I", v)).digest()  # committed little-endian (bincode)
    assert ch.state == st
    path = [bytes([i]) * 32 for i in range(3)]
    ch.commit((7, path))
    enc = struct.pack("<I", 7) + struct.pack("<Q", 3) + b"".join(path)
    st = hashlib.sha256(st + enc).digest()
    assert ch.state == st
    assert ch.data == root + struct.pack("<I", v) + enc


def test_verifier_and_wire_format(zk, orc):
    canon = json.load(open(os.path.join(GOLD, "stark101_canonical.json")))["derived"]
    proof = zk.Proof(bytes.fromhex(canon["final_state"]), bytes.fromhex(canon["proof_hex"]))
    proof.verify()                                           # proof.rs:15
    assert proof.size() == 7884                              # proof.rs:151-154
    from zkstark_amd import _lib
    assert _lib.load().zk_proof_data_len(10, 3) == 7836
    for pos in (3, 33, 80, 500, 7000):
        bad = bytearray(proof.data)
        bad[pos] ^= 0x80
        with pytest.raises(zk.ZkError):
            zk.Proof(proof.state, bytes(bad)).verify()
    with pytest.raises(zk.ZkError):
        zk.Proof(proof.state, proof.data, public_last=1).verify()
    for row in json.load(open(os.path.join(GOLD, "prover_sizes.json"))):
        if row["log_n"] > 8:
            continue
        r = orc.prove(row["log_n"], row["log_blowup"], 1, row["a1"], want_vectors=False)
        zk.Proof(r.state, r.proof, row["log_n"], row["log_blowup"], row["public_last"]).verify()


def test_compute_root_from_path_host(zk, orc):
    nodes = orc.merkle_build([1, 2, 3, 4])
    path = orc.merkle_trace(nodes, 0)
    assert zk.compute_root_from_path(1, 0, [bytes(p) for p in path]) == bytes(nodes[0])   # merkle.rs:181


def test_compute_calls_fail_loudly_without_gpu(zk):
    """No CPU fallback: without a visible GPU every compute entry point raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(zk.ZkError):
        zk.Context(10, 3)
    with pytest.raises(zk.ZkError):
        zk.Merkle.new(4, [1, 2, 3, 4])
    with pytest.raises(zk.ZkError):
        zk.ntt([1, 2, 3, 4])
    with pytest.raises(zk.ZkError):
        zk.BatchContext(6, 2, 2)
    with pytest.raises(zk.ZkError):
        zk.trace_fibsq_batch([1, 1], [3, 4], 63)
    import ctypes as C
    from zkstark_amd import _lib
    h = C.c_void_p()
    assert _lib.load().zk_committer_create(0, C.byref(h)) < 0 and not h.value
    assert _lib.load().zk_dom_create(0, 6, 2, 5, 0, C.byref(h)) < 0 and not h.value


def test_context_argument_checks(zk):
    for log_n, log_b in ((1, 3), (10, 0), (10, 6), (28, 3), (3, 3)):
        with pytest.raises(zk.ZkError) as e:
            zk.Context(log_n, log_b)
        assert e.value.code == -1


def test_strict_verifier_replays_the_transcript(zk, orc):
    """challenges must be the transcript's, Proof.state must be the final state."""
    import struct as st
    r = orc.prove(6, 2, want_vectors=False)
    good = zk.Proof(r.state, r.proof, 6, 2, r.public_last)
    good.verify(strict=True)
    with pytest.raises(zk.ZkError):                      # wrong final state: the lax verifier does not notice
        zk.Proof(bytes(32), r.proof, 6, 2, r.public_last).verify(strict=True)
    zk.Proof(bytes(32), r.proof, 6, 2, r.public_last).verify()
    # a prover that picks its own alpha0: the reference verifier trusts it, the strict one does not
    forged = bytearray(r.proof)
    forged[32:36] = st.pack("<I", (st.unpack("<I", forged[32:36])[0] + 1) & 0xFFFFFFFF)
    with pytest.raises(zk.ZkError) as e:
        zk.Proof(r.state, bytes(forged), 6, 2, r.public_last).verify(strict=True)
    assert "transcript" in str(e.value)


def test_multi_query_proofs(zk, orc):
    """q queries per proof; q = 1 is the reference's byte format."""
    try:
        base = orc.prove(6, 2, want_vectors=False)
        for q in (2, 4):
            orc.set_queries(q)
            r = orc.prove(6, 2, want_vectors=False)
            assert r.rc == 0 and len(r.proof) == orc.proof_data_len(6, 2)
            assert r.proof[:len(base.proof) - (len(base.proof) - (32 + 12 + 32 + 6 * 36 + 4))] == base.proof[:32 + 12 + 32 + 6 * 36 + 4]
            assert orc.verify(r.proof, 6, 2, r.public_last) == 0
            p = zk.Proof(r.state, r.proof, 6, 2, r.public_last, queries=q)
            p.verify(strict=True)
            with pytest.raises(zk.ZkError):      # a q-query proof is not a (q-1)-query proof
                zk.Proof(r.state, r.proof, 6, 2, r.public_last, queries=q - 1).verify()
            bad = bytearray(r.proof)
            bad[-100] ^= 1                       # tamper with the LAST query's openings
            with pytest.raises(zk.ZkError):
                zk.Proof(r.state, bytes(bad), 6, 2, r.public_last, queries=q).verify()
    finally:
        orc.set_queries(1)


def test_shard_plan_is_the_one_layout(zk):
    """zk_shard_plan (no GPU): the layout both the native sharded prover and the torch.distributed mirror follow.
    The benchmark's weak-scaling configurations, the byte formula of DESIGN.md section 6, and the mirror's view of it."""
    # bench.py --gpus 2 / 4 / 8 (2^24 elements per GPU, production thresholds): GPU tests assert the same numbers on the prover's stats
    # (round 5: layers of >= 2^21 values stay distributed, >= 2^20 from 4 ranks on; the last rows are the strong shape of bench.py)
    for world, log_n, want_sharded, want_chunked in ((2, 22, 5, 3), (4, 23, 7, 2), (8, 24, 8, 1), (8, 21, 5, 0), (4, 21, 5, 0), (2, 21, 4, 2), (1, 21, 4, 0)):
        pl = zk.shard_plan(world, log_n, 3)
        assert (pl["sharded_layers"], pl["chunked_layers"]) == (want_sharded, want_chunked), (world, log_n, pl)
        assert pl["tail_rounds"] == log_n - want_sharded and pl["log_chunks"] == 2
        N = 1 << (log_n + 3)
        words = N + sum(N >> rho for rho in range(1, want_sharded))       # f and FRI layers 1 .. ns-1: one all-to-all each; cp (FRI
        assert pl["cp_from_f"] == (1 if world > 1 else 0)                 # layer 0) over a rank's block comes from the block of f it received
        if world == 1:
            words = 0
        assert pl["all_to_all_bytes"] == 4.0 * words / world * (world - 1) / world
        old = zk.shard_plan(world, log_n, 3, exchange_cp=True)            # rounds 1-4: cp exchanged as well, 12 N instead of 8 N
        assert old["cp_from_f"] == 0 and old["all_to_all_bytes"] == 4.0 * (N + sum(N >> rho for rho in range(want_sharded))) / world * (world - 1) / world
        assert pl["lde_commit_bytes"] == 4.0 * N / world * (world - 1) / world
        assert pl["piece_log"][0] == log_n + 3 - 2 * pl["log_world"]
    # explicit thresholds; at least one round stays in the replicated tail; a chunked layer needs pieces of >= 2^10 words
    pl = zk.shard_plan(2, 6, 3, min_layer_log=1, min_chunk_log=1)
    assert pl["sharded_layers"] == 5 and pl["tail_rounds"] == 1
    pl = zk.shard_plan(2, 9, 3, min_layer_log=1, min_chunk_log=3, overlap_min_log=3)
    assert pl["chunked_mask"] & 1 and pl["piece_log"][0] == 10 and not (pl["chunked_mask"] >> 2) & 1
    # zk_shard_options.plain_collectives (the fall-back rung of bench.py): the same distributed layers, nothing in chunks
    pl = zk.shard_plan(8, 24, 3, plain_collectives=True)
    assert pl["sharded_layers"] == 8 and pl["chunked_mask"] == 0 and pl["chunked_layers"] == 0 and pl["overlap_min_log"] == 99
    assert pl["all_to_all_bytes"] == zk.shard_plan(8, 24, 3)["all_to_all_bytes"]
    for bad in ((3, 12, 3), (16, 12, 3), (2, 3, 3), (4, 4, 2)):
        with pytest.raises(zk.ZkError) as e:
            zk.shard_plan(*bad)
        assert e.value.code == -1
    # the mirror (test infrastructure since round 4) holds no thresholds of its own, and the package holds no second prover
    src = open(os.path.join(ROOT, "tests", "sharded_mirror.py")).read()
    assert "shard_plan(" in src and "chunk_min_log" not in src
    pkg = open(os.path.join(ROOT, "zkstark_amd", "sharded.py")).read()
    assert "class ShardedProver" not in pkg and "def staged_transport" in pkg and "def device_transport" in pkg


def test_host_hash_mode_follows_the_cpu(zk):
    """zk_host_hash_mode: 0 portable, 1 x86 SHA extensions, 2 + AVX-512 (sixteen nodes at a time); it must agree with
    what the CPU advertises, because the provers hand tree tops to the host thread only when it returns >= 1."""
    flags = set()
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = set(line.split(":", 1)[1].split())
                    break
    except OSError:
        pytest.skip("no /proc/cpuinfo")
    want = 0
    if {"sha_ni", "sse4_1", "ssse3"} <= flags:
        want = 2 if "avx512f" in flags else 1
    names = ("portable", "sha-ni", "sha-ni + avx512 x16")
    assert zk.host_hash_mode() == names[want]
    # zk_host_set_hash_mode: never more than the CPU has, and the narrow setting survives switching the extensions on again
    lib = zk.load()
    try:
        for mode in (0, 1, 2):
            assert lib.zk_host_set_hash_mode(mode) == 0
            assert zk.host_hash_mode() == names[min(mode, want)]
        assert lib.zk_host_set_hash_mode(3) == -1
    finally:
        lib.zk_host_set_hash_mode(2)
    assert zk.host_hash_mode() == names[want]


def test_tuning_surface_is_calls_not_environment(zk):
    """Round 4 froze the tuning surface: the library reads two operational environment variables,
    everything else is a documented call, an option field or a build-time constant.  The setters that need no GPU are
    exercised here; the source is checked for stray getenv calls."""
    import re
    lib = zk.load()
    assert lib.zk_dev_set_merkle_latency_log(11) == -1 and lib.zk_dev_set_merkle_latency_log(25) == -1
    assert lib.zk_dev_set_merkle_latency_log(16) == 0 and lib.zk_dev_set_merkle_latency_log(0) == 0      # 0 = the build's default
    names = set()
    csrc = os.path.join(ROOT, "zkstark_amd", "csrc")
    for fn in os.listdir(csrc):
        with open(os.path.join(csrc, fn)) as f:
            names |= set(re.findall(r'getenv\("([A-Z_0-9]+)"\)', f.read()))
    # ZK_WG_TRACE_FILE exists in the DIAGNOSTIC build only (ZK_BUILD_DEFS="-DZK_WG_TRACE=1", tools/wg_trace.py): it must sit inside
    # that #ifdef, so a product build still reads exactly two variables
    kernels = open(os.path.join(csrc, "kernels.hip")).read()
    at = kernels.index('getenv("ZK_WG_TRACE_FILE")')
    assert kernels.rfind("#ifdef ZK_WG_TRACE", 0, at) > kernels.rfind("#endif", 0, at)
    names.discard("ZK_WG_TRACE_FILE")
    assert names == {"ZK_HOST_TIMING", "ZK_SHARD_TIMEOUT_S"}, names
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert all(n in doc for n in names)


# ---- ABI version and caller-allocated structs (include/zkstark_amd.h) ------------------------------------------------------
ABI_STRUCTS = {"zk_transcript_info": "TranscriptInfo", "zk_kernel_stat": "KernelStat", "zk_shard_options": "ShardOptions",
               "zk_shard_stats": "ShardStats", "zk_shard_plan_info": "ShardPlan", "zk_chain_probe": "ChainProbe"}


def test_abi_version_and_struct_sizes_agree_everywhere(zk, tmp_path):
    """One ABI version in the header, the library and the ctypes mirror; every caller-allocated struct starts with struct_size,
    and its size as gcc lays it out equals the ctypes mirror's and the library's version-6 minimum (csrc/internal.hpp: AbiMin)."""
    import subprocess
    from zkstark_amd import _lib
    header = open(os.path.join(ROOT, "include", "zkstark_amd.h")).read()
    ver = int(re.search(r"#define\s+ZK_ABI_VERSION\s+(\d+)u", header).group(1))
    assert _lib.load().zk_abi_version() == ver == _lib.ABI_VERSION
    src = tmp_path / "sizes.c"
    src.write_text('#include <stddef.h>\n#include <stdio.h>\n#include "zkstark_amd.h"\nint main(void) {\n' +
                   "".join(f'    printf("{n} %zu %zu\\n", sizeof({n}), offsetof({n}, struct_size));\n' for n in ABI_STRUCTS) +
                   "    zk_shard_options o; ZK_STRUCT_INIT(&o); return o.struct_size == sizeof o && o.timeout_s == 0 ? 0 : 1;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    internal = open(os.path.join(ROOT, "zkstark_amd", "csrc", "internal.hpp")).read()
    for line in out.splitlines():
        name, size, off = line.split()
        mirror = getattr(_lib, ABI_STRUCTS[name])
        assert int(off) == 0 and mirror._fields_[0][0] == "struct_size"
        assert int(size) == C.sizeof(mirror), (name, size, C.sizeof(mirror))
        amin = int(re.search(r"AbiMin<%s>\s*\{ static constexpr uint32_t v = (\d+);" % name, internal).group(1))
        assert amin <= int(size), name              # == today; a struct may only grow at its end
        assert mirror().struct_size == int(size)


def test_stale_or_unsized_structs_are_refused(zk):
    """A caller compiled against an older header passes a smaller struct (or one that never set struct_size): the library answers
    ZK_ERR_INVALID naming the struct and both sizes instead of reading / writing past it (round 5's zk_shard_stats grew in the
    middle).  A LARGER struct (a newer caller) is accepted and the bytes the library does not know stay untouched."""
    from zkstark_amd import _lib
    lib = _lib.load()
    plan_fn = lib.zk_shard_plan
    saved = plan_fn.argtypes
    plan_fn.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    try:
        good_opt, good_plan = _lib.ShardOptions(), _lib.ShardPlan()
        assert plan_fn(2, 22, 3, C.byref(good_opt), C.byref(good_plan)) == 0 and good_plan.sharded_layers == 5
        # output struct: size 0 (never set: the round-5 layout began with `world`), the round-5 size, an absurd size
        for bad in (0, C.sizeof(_lib.ShardPlan) - 8, 1 << 20):
            pl = _lib.ShardPlan()
            pl.struct_size = bad
            pl.world = 777
            assert plan_fn(2, 22, 3, None, C.byref(pl)) == -1
            msg = lib.zk_last_error().decode()
            assert "zk_shard_plan_info.struct_size" in msg and str(bad) in msg and str(C.sizeof(_lib.ShardPlan)) in msg, msg
            assert pl.world == 777                                     # nothing was written
        # input struct: the same rule
        for bad in (0, 21, 40):                                        # never set; round 5's first field (min_layer_log); too small
            opt = _lib.ShardOptions()
            opt.struct_size = bad
            assert plan_fn(2, 22, 3, C.byref(opt), C.byref(_lib.ShardPlan())) == -1
            assert "zk_shard_options.struct_size" in lib.zk_last_error().decode()
        # a caller compiled against the FIRST version-6 layout (48 bytes, before `peer_copy` was appended): accepted, and what lies
        # behind its struct is not read (here: a set peer_copy field, which would force plain collectives)
        opt = _lib.ShardOptions(peer_copy=1)
        assert C.sizeof(opt) > 48
        pl = _lib.ShardPlan()
        assert plan_fn(8, 24, 3, C.byref(opt), C.byref(pl)) == 0 and pl.overlap_min_log == 99
        opt.struct_size = 48
        assert plan_fn(8, 24, 3, C.byref(opt), C.byref(pl)) == 0 and pl.overlap_min_log == 21
        # a larger struct from a newer caller: fields the library knows are filled, the tail is left alone

        class BiggerPlan(C.Structure):
            _fields_ = [("base", _lib.ShardPlan), ("future", C.c_uint8 * 24)]
        big = BiggerPlan()
        C.memset(C.byref(big), 0xAB, C.sizeof(big))
        big.base.struct_size = C.sizeof(big)
        assert plan_fn(2, 22, 3, None, C.byref(big)) == 0
        assert big.base.sharded_layers == 5 and big.base.struct_size == C.sizeof(big) and bytes(big.future) == b"\xab" * 24
        assert bytes(big.base)[4:] == bytes(good_plan)[4:]

        class BiggerOpt(C.Structure):
            _fields_ = [("base", _lib.ShardOptions), ("future", C.c_uint32 * 4)]
        bo = BiggerOpt()
        bo.base.struct_size = C.sizeof(bo)
        bo.base.plain_collectives = 1
        bo.future[0] = 12345                                           # an option this library does not know: ignored
        pl = _lib.ShardPlan()
        assert plan_fn(8, 24, 3, C.byref(bo), C.byref(pl)) == 0 and pl.overlap_min_log == 99
    finally:
        plan_fn.argtypes = saved
    # the harness and the examples check the version and size their structs (what the r05i session lacked)
    for rel in ("tests/shard_threads_check.c", "examples/shard_c_abi.c", "examples/prove_c_abi.c", "examples/batch_c_abi.c"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "zk_abi_version() != ZK_ABI_VERSION" in text, rel
    harness = open(os.path.join(ROOT, "tests", "shard_threads_check.c")).read()
    assert "ZK_STRUCT_INIT(&args[r].opt)" in harness and "opt.timeout_s = timeout_s" in harness and "ZK_EXPECT_BUILD_HASH" in harness


def test_a_stale_harness_binary_refuses_to_run(zk, tmp_path):
    """tests/shard_threads_check.c compiled for another build of the library (what shipped to the GPU box in round 5's killed
    session: a git-ignored binary under tools/ that a script did not recompile) names the mismatch and exits before it touches
    the GPU; compiled by the one recipe (tools/build_shard_threads_check.sh) it gets as far as the missing GPU."""
    import subprocess
    inc = ["-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")]
    link = ["-L" + os.path.join(ROOT, "zkstark_amd"), "-lzkstark_amd", "-L/opt/rocm/lib", "-lamdhip64",
            "-Wl,-rpath," + os.path.join(ROOT, "zkstark_amd"), "-Wl,-rpath,/opt/rocm/lib"]
    stale = str(tmp_path / "stale")
    subprocess.check_call(["gcc", "-O1", "-pthread", '-DZK_EXPECT_BUILD_HASH="0123456789abcdef01234567"'] + inc +
                          [os.path.join(ROOT, "tests", "shard_threads_check.c")] + link + ["-o", stale])
    out = subprocess.run([stale, "2", "10", "3", "1", "1", "99"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "stale harness: compiled for library build 0123456789abcdef01234567" in out.stderr, out.stderr
    fresh = str(tmp_path / "fresh")
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_shard_threads_check.sh"), fresh], stdout=subprocess.DEVNULL)
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([fresh, "2", "10", "3", "1", "1", "99"], capture_output=True, text=True, timeout=60)
        assert out.returncode == 1 and "stale harness" not in out.stderr and "rank 0: -2" in out.stderr, out.stderr
