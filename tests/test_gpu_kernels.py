"""GPU parity: each HIP stage against the CPU oracle on the same seeded inputs (bit-exact).

All calls go through the C ABI (ctypes -> libzkstark_amd.so).
"""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P = 3221225473


def rand_field(rng, n):
    return rng.integers(0, P, size=n, dtype=np.uint64).astype(np.uint32)


# ---- Merkle (merkle.rs) -----------------------------------------------------------
def test_merkle_reference_vectors(zk):
    """merkle_test, merkle.rs:112-182: leaves [1,2,3,4], all 7 nodes, 4 paths, root from path."""
    m = zk.Merkle.new(4, [1, 2, 3, 4])
    i3 = "b40711a88c7039756fb8a73827eabe2c0fe5a0346ca7e0a104adc0fc764f528d"
    i4 = "433ebf5bc03dffa38536673207a21281612cef5faa9bc7a4d5b9be2fdb12cf1a"
    i5 = "88185d128d9922e0e6bcd32b07b6c7f20f27968eab447a1d8d1cdf250f79f7d3"
    i6 = "1bc5d0e3df0ea12c4d0078668d14924f95106bbe173e196de50fe13a900b0937"
    i1 = "be8dc357decb6e09c8e5ad874d3c4fa7fc09730bbb5e90f42c97dad20e0012d4"
    i2 = "6bed5b6d7ae093d1812ab9be5cbfa1ce787812a003d95c11448720a407b61727"
    i0 = "327cf213e1738de4206bfd14297c26c682961750cb56897ed5e8f519b0548ff2"
    assert [m[i].hex() for i in range(7)] == [i0, i1, i2, i3, i4, i5, i6]
    assert [h.hex() for h in m.trace(0)] == [i4, i2]
    assert [h.hex() for h in m.trace(1)] == [i3, i2]
    assert [h.hex() for h in m.trace(2)] == [i6, i1]
    assert [h.hex() for h in m.trace(3)] == [i5, i1]
    assert zk.compute_root_from_path(1, 0, m.trace(0)) == m[0]


@pytest.mark.parametrize("log_m", [0, 1, 2, 5, 10, 11, 12, 13, 16, 19, 20])
def test_merkle_matches_oracle(zk, orc, log_m):
    rng = np.random.default_rng(100 + log_m)
    vals = rand_field(rng, 1 << log_m)
    vals[0] = 0
    if log_m:
        vals[1] = P - 1
    got = zk.Merkle.new(1 << log_m, vals).nodes
    want = orc.merkle_build(vals)
    assert got.shape == want.shape
    assert hashlib.sha256(got.tobytes()).hexdigest() == hashlib.sha256(want.tobytes()).hexdigest()


def test_merkle_rejects_non_power_of_two(zk):
    with pytest.raises(zk.ZkError):      # merkle.rs:18 assert_eq!(i % 2, 0)
        zk.Merkle.new(6, [1, 2, 3, 4, 5, 6])
    with pytest.raises(zk.ZkError):
        zk.Merkle.new(0, [])


# ---- NTT ------------------------------------------------------------------------
@pytest.mark.parametrize("log_m", [1, 2, 3, 7, 8, 9, 10, 13, 14, 16, 17, 20, 21, 22])      # 21, 22: the 4096-word tile (round 4)
def test_ntt_forward_inverse_match_oracle(zk, orc, log_m):
    rng = np.random.default_rng(200 + log_m)
    x = rand_field(rng, 1 << log_m)
    root = orc.gen_of_order_log(log_m)
    fwd = zk.ntt(x)
    assert np.array_equal(fwd, orc.ntt(x, root))
    inv = zk.ntt(x, inverse=True)
    assert np.array_equal(inv, orc.intt(x, root))
    assert np.array_equal(zk.ntt(fwd, inverse=True), x)


# ---- LDE (polynomial.rs lagrange + solve; prover.rs:60-78) -------------------------
def test_lde_reference_checkpoints(zk):
    """prover.rs:42, :73-78."""
    a = zk.trace_fibsq(1023)
    assert a[1022] == 2338775057
    f = zk.lde(a, 10, 3)
    assert list(f[:3]) == [576067152, 3100214617, 2091264768]
    assert list(f[-3:]) == [800520420, 1199720174, 1076821037]


# (18, 3), (19, 3): LDE passes on the 4096-word tile; (21, 1): the inverse passes on it and coefficient blocks of half a tile in LDS;
# (16, 5): a first LDE pass the register-radix kernel does not take (separate coefficient sweep)
@pytest.mark.parametrize("log_n,log_b", [(2, 1), (4, 3), (5, 2), (8, 3), (9, 4), (10, 3), (13, 3), (14, 3), (17, 3), (16, 1), (9, 5), (16, 5), (20, 2),
                                         (18, 3), (19, 3), (21, 1), (12, 2), (15, 4)])
def test_lde_matches_oracle(zk, orc, log_n, log_b):
    rng = np.random.default_rng(300 + log_n * 8 + log_b)
    trace = rand_field(rng, (1 << log_n) - 1)   # arbitrary trace values, not only Fibonacci-square
    got = zk.lde(trace, log_n, log_b)
    assert np.array_equal(got, orc.lde(trace, log_n, log_b))


# ---- composition + fold (prover.rs:101-225) --------------------------------------------
@pytest.mark.parametrize("log_n,log_b", [(2, 1), (4, 3), (10, 3), (13, 2), (16, 3)])
def test_compose_and_fold_match_oracle(zk, orc, log_n, log_b):
    rng = np.random.default_rng(400 + log_n)
    N = 1 << (log_n + log_b)
    f = rand_field(rng, N)
    alphas = [int(rng.integers(0, 2**32)) for _ in range(3)]
    alphas[1] = 3235878091   # >= P: must be reduced (field.rs:20-24)
    with zk.Context(log_n, log_b) as ctx:
        trace = rand_field(rng, (1 << log_n) - 1)
        trace[0] = 1   # a[0]: prover.rs:101 uses a[0], the verifier the literal 1 (proof.rs:69)
        ctx.trace_upload(trace)
        ctx.layer_write(0, f)
        ctx.compose(alphas)
        cp = ctx.layer_read(1)
        assert np.array_equal(cp, orc.compose(f, log_n, log_b, alphas, int(trace[-1])))
        layer = cp
        for r in range(log_n):
            beta = int(rng.integers(0, 2**32)) if r != 1 else 4195595581
            ctx.fri_fold(r, beta)
            nxt = ctx.layer_read(2 + r)
            assert np.array_equal(nxt, orc.fri_fold_eval(layer, log_n, log_b, r, beta)), f"round {r}"
            layer = nxt


# ---- whole prover ---------------------------------------------------------------------
def test_prover_canonical_run(zk, orc):
    """generate_proof at the reference's literals: proof bytes equal to the oracle's, verifier accepts."""
    want = orc.prove(10, 3)
    with zk.Context(10, 3) as ctx:
        proof = ctx.prove(zk.trace_fibsq(1023))
        info = ctx.last_transcript()
        assert list(info.alpha_raw) == want.alpha_raw
        assert list(info.beta_raw)[:10] == want.beta_raw
        for t in range(12):
            assert bytes(info.roots[t]) == bytes(want.roots[t])
        assert info.free_term == want.free_term and info.query_raw == want.query_raw
    assert proof.data == want.proof and proof.state == want.state
    assert hashlib.sha256(proof.data).hexdigest() == "b956f69349dfb74d2facd9f886efa8b983fb61f17bd95cab2e3449fc57b4bb2e"
    proof.verify()
    assert proof.size() == 7884
    assert orc.verify(proof.data, 10, 3, 2338775057) == 0


def test_generate_proof_staged_equals_one_call(zk):
    """The prover.rs-shaped staged flow over the C ABI and the one-call C++ prover agree."""
    p1 = zk.generate_proof(zk.Channel())
    with zk.Context(10, 3) as ctx:
        p2 = ctx.prove(zk.trace_fibsq(1023))
    assert p1.data == p2.data and p1.state == p2.state
    p1.verify()


@pytest.mark.parametrize("log_n,log_b,a1", [(2, 1, 3141592), (4, 3, 7), (6, 2, 3141592), (12, 3, 99), (15, 3, 3141592), (17, 3, 5), (11, 5, 3), (18, 1, 3141592), (7, 4, 11)])
def test_prover_other_sizes(zk, orc, log_n, log_b, a1):
    want = orc.prove(log_n, log_b, 1, a1, want_vectors=False)
    assert want.rc == 0
    with zk.Context(log_n, log_b) as ctx:
        proof = ctx.prove(zk.trace_fibsq((1 << log_n) - 1, 1, a1))
    assert proof.data == want.proof and proof.state == want.state
    proof.verify()


def test_prover_rejects_bad_trace(zk):
    """A trace that breaks the recurrence must not yield a proof (reference: assert/panic)."""
    a = zk.trace_fibsq(1023)
    a[500] = (int(a[500]) + 1) % P
    with zk.Context(10, 3) as ctx:
        with pytest.raises(zk.ZkError):
            ctx.prove(a)


@pytest.mark.parametrize("log_n,log_b", [(6, 2), (10, 3), (14, 3), (18, 3)])
def test_reference_self_checks(zk, orc, log_n, log_b):
    """zk_ctx_set_checks: the reference's in-prover assertions run on the device data.  A good trace passes every
    checkpoint and yields the oracle's bytes; a trace corrupted at one of three positions is stopped at the FIRST
    checkpoint it violates -- the exact divisions / deg cp = n - 1 of prover.rs:148-159 and :169 -- not at the last FRI
    layer (prover.rs:238), which is where the unchecked prover notices."""
    n = 1 << log_n
    a = zk.trace_fibsq(n - 1)
    want = orc.prove(log_n, log_b, want_vectors=False)
    with zk.Context(log_n, log_b) as ctx:
        ctx.set_checks(True)
        proof = ctx.prove(a)
        assert proof.data == want.proof and proof.state == want.state
        for pos in (0, n // 2, n - 2):
            bad = a.copy()
            bad[pos] = (int(bad[pos]) + 1) % P
            with pytest.raises(zk.ZkError) as e:
                ctx.prove(bad)
            assert e.value.code == -7 and "prover.rs:148-159/:169" in str(e.value), str(e.value)
        ctx.set_checks(False)
        bad = a.copy()
        bad[n // 2] = (int(bad[n // 2]) + 1) % P
        with pytest.raises(zk.ZkError) as e:
            ctx.prove(bad)
        assert e.value.code == -7 and "prover.rs:238" in str(e.value)
        assert ctx.prove(a).data == want.proof            # and the context is still good


@pytest.mark.parametrize("host_levels,hash_name", [((0, 0), "sha256"), ((8, 9), "sha256"), ((5, 6), "sha256"), (None, "field")])
def test_self_checks_with_every_division_of_labour(zk, orc, host_levels, hash_name):
    """The checkpoints read the layers where they are: all on the device (host_levels (0, 0)), small FRI layers on the
    host thread (the default), other hand-over depths, and with the field hash.  A good trace passes all of them and
    the proof bytes do not change."""
    log_n, log_b = 13, 3
    a = zk.trace_fibsq((1 << log_n) - 1)
    with zk.Context(log_n, log_b, hash=hash_name, host_levels=host_levels) as ctx:
        plain = ctx.prove(a)
        ctx.set_checks(True)
        checked = ctx.prove(a)
        assert checked.data == plain.data and checked.state == plain.state
        bad = a.copy()
        bad[77] = (int(bad[77]) + 5) % P
        with pytest.raises(zk.ZkError) as e:
            ctx.prove(bad)
        assert e.value.code == -7 and "prover.rs:148-159/:169" in str(e.value)


def test_prove_channel_uses_the_callers_channel(zk, orc):
    """generate_proof(channel) (prover.rs:9): zk_prove_channel proves on the caller's Channel.  A fresh channel
    gives zk_prove's bytes; a channel with a committed prefix gives the oracle's bytes for the same prefix."""
    for log_n, log_b in ((10, 3), (6, 2), (13, 3)):
        a = zk.trace_fibsq((1 << log_n) - 1)
        with zk.Context(log_n, log_b) as ctx:
            ctx.trace_upload(a)
            plain = ctx.prove()
            p0 = ctx.prove_channel(zk.Channel())
            assert p0.data == plain.data and p0.state == plain.state
            for prefix in (b"x", b"session 7: " + bytes(range(40))):
                ch = zk.Channel()
                ch.commit(prefix)
                p1 = ctx.prove_channel(ch)
                want_data, want_state = orc.prove_prefixed(prefix, log_n, log_b)
                assert p1.data == want_data and p1.state == want_state
                assert p1.data[len(prefix):len(prefix) + 32] == plain.data[:32] and p1.data[len(prefix) + 32:] != plain.data[32:]
            # an imported channel (the Rust wrapper's path) behaves the same
            from zkstark_amd import _lib
            src = zk.Channel(); src.commit(b"x")
            imp = zk.Channel()
            _lib.check(_lib.load().zk_channel_import(imp._h, src.state, src.data, len(src.data)))
            assert ctx.prove_channel(imp).data == orc.prove_prefixed(b"x", log_n, log_b)[0]


# ---- BASELINE.json full sizes ------------------------------------------------------------
def test_config2_lde_commit_domain_2e20(zk, orc):
    """configs[1]: domain 2^20 LDE + Merkle commit, bit-exact vs the CPU oracle (values and root)."""
    a = zk.trace_fibsq((1 << 17) - 1)
    with zk.Context(17, 3) as ctx:
        ctx.trace_upload(a)
        ctx.lde()
        root = ctx.merkle_commit(0)
        f = ctx.layer_read(0)
    want = orc.lde(a, 17, 3)
    assert np.array_equal(f, want)
    assert root == bytes(orc.merkle_build(want)[0])


def test_config3_full_prover_domain_2e24(zk, orc):
    """configs[2] at full size, bit-exact against the CPU oracle: the proof bytes, the final channel state,
    every one of the 23 Merkle roots (each feeds the channel: prover.rs:81-85, :176-180, :214-224), every
    challenge, and the whole f_eval vector.  This is the size that reaches the three radix-128 NTT passes, the
    multi-launch subtree chain, indexing above 2^31 bytes and the host hand-over with 2^24 leaves."""
    log_n, log_b = 21, 3
    a = zk.trace_fibsq((1 << log_n) - 1)
    want = orc.prove(log_n, log_b, want_vectors=False, want_roots=True)
    assert want.rc == 0
    want_f = orc.lde(a, log_n, log_b)
    with zk.Context(log_n, log_b) as ctx:
        proof = ctx.prove(a)
        info = ctx.last_transcript()
        got_f = ctx.layer_read(0)
        # idempotence: a second proof from the resident trace is byte-identical
        proof2 = ctx.prove()
        last = ctx.layer_read(1 + log_n)
        # the same bytes with the whole tree built on the device (no host hand-over)
        ctx.set_host_levels(0, 0)
        proof3 = ctx.prove()
    assert np.array_equal(got_f, want_f)
    for t in range(log_n + 2):
        assert bytes(info.roots[t]) == bytes(want.roots[t]), f"root of tree {t}"
    assert list(info.alpha_raw) == want.alpha_raw and list(info.beta_raw)[:log_n] == want.beta_raw
    assert info.free_term == want.free_term and info.query_raw == want.query_raw
    assert proof.data == want.proof, "proof bytes differ from the CPU oracle at domain 2^24"
    assert proof.state == want.state
    assert proof2.data == proof.data and proof2.state == proof.state
    assert proof3.data == proof.data and proof3.state == proof.state
    assert len(set(int(v) for v in last)) == 1 and int(last[0]) == info.free_term
    assert len(proof.data) == 32 + 12 + 32 + 21 * 36 + 8 + 4 * (12 + 32 * 24) + sum(8 + 2 * (8 + 32 * (24 - i)) for i in range(21))
    proof.verify()
    assert orc.verify(proof.data, log_n, log_b, int(a[-1])) == 0
    bad = bytearray(proof.data)
    bad[-5] ^= 1
    with pytest.raises(zk.ZkError):
        zk.Proof(proof.state, bytes(bad), log_n, log_b, int(a[-1])).verify()


def test_merkle_full_size_root_property(zk, orc):
    """2^22 leaves: root(left half), root(right half) hash to the root (checksum of checksums)."""
    import hashlib as hl
    rng = np.random.default_rng(5)
    vals = rand_field(rng, 1 << 22)
    m = zk.Merkle.new(1 << 22, vals)
    assert m[0] == hl.sha256(m[1] + m[2]).digest()
    half = zk.Merkle.new(1 << 21, vals[:1 << 21])
    assert half[0] == m[1]
    for leaf in (0, 12345, (1 << 22) - 1):
        assert zk.compute_root_from_path(int(vals[leaf]), leaf, m.trace(leaf)) == m[0]


def test_ntt_linearity_and_roundtrip_2e22(zk):
    rng = np.random.default_rng(6)
    x, y = rand_field(rng, 1 << 22), rand_field(rng, 1 << 22)
    fx, fy = zk.ntt(x), zk.ntt(y)
    s = ((x.astype(np.uint64) + y) % P).astype(np.uint32)
    assert np.array_equal(zk.ntt(s), ((fx.astype(np.uint64) + fy) % P).astype(np.uint32))
    assert np.array_equal(zk.ntt(fx, inverse=True), x)


# ---- error behaviour at the boundary (the reference panics; the C ABI returns a status) ---------
def test_stage_order_and_argument_errors(zk):
    from zkstark_amd import _lib
    import ctypes as C
    lib = _lib.load()
    with zk.Context(6, 2) as ctx:
        with pytest.raises(zk.ZkError) as e:          # compose before lde
            ctx.compose([1, 2, 3])
        assert e.value.code == -4
        with pytest.raises(zk.ZkError) as e:          # prove without a trace
            ctx.prove()
        assert e.value.code == -4
        with pytest.raises(zk.ZkError) as e:          # wrong trace length (prover.rs:60 needs n-1 points)
            ctx.trace_upload(np.ones(64, dtype=np.uint32))
        assert e.value.code == -1
        with pytest.raises(zk.ZkError) as e:          # non-canonical residue
            ctx.trace_upload(np.full(63, P, dtype=np.uint32))
        assert e.value.code == -1
        a = zk.trace_fibsq(63)
        ctx.trace_upload(a)
        with pytest.raises(zk.ZkError):               # out-of-range layer / fold round
            ctx.layer_read(99)
        with pytest.raises(zk.ZkError):
            ctx.fri_fold(6, 1)
        # caller buffer too small: ZK_ERR_BUFFER and the needed length is reported
        buf = C.create_string_buffer(16)
        st = C.create_string_buffer(32)
        n = C.c_size_t()
        rc = lib.zk_prove_resident(ctx._h, buf, 16, C.byref(n), st)
        assert rc == -5 and n.value == lib.zk_proof_data_len(6, 2)
        proof = ctx.prove()                           # the context is still usable afterwards
        proof.verify(strict=True)


def test_merkle_index_and_path_api(zk, orc):
    """Index<usize> and trace() (merkle.rs:54-79) on a context-resident tree."""
    a = zk.trace_fibsq(255)
    with zk.Context(8, 2) as ctx:
        ctx.trace_upload(a)
        ctx.lde()
        root = ctx.merkle_commit(0)
        f = ctx.layer_read(0)
        nodes = orc.merkle_build(f)
        assert root == bytes(nodes[0])
        for idx in (0, 1, 2, 511, 1023, 2046):
            assert ctx.merkle_node(0, idx) == bytes(nodes[idx])
        for leaf in (0, 7, 1023):
            assert ctx.merkle_path(0, leaf) == [bytes(h) for h in orc.merkle_trace(nodes, leaf)]
        with pytest.raises(zk.ZkError):
            ctx.merkle_node(0, 2047)


def test_merkle_commit_pending_tree_top(zk, orc):
    """zk_merkle_commit returns with the root and leaves the device copy of the host-built tree top pending: a second commitment
    of the SAME tree supersedes it, a commitment of ANOTHER tree or a proof must not lose it, and every reader of the device
    arrays sees whole trees (merkle.rs:14-79)."""
    if zk.host_hash_mode() == "portable":
        pytest.skip("no host hand-over on this CPU: nothing is ever pending")
    a = zk.trace_fibsq((1 << 12) - 1)
    with zk.Context(12, 3) as ctx:
        ctx.trace_upload(a)
        ctx.lde()
        f = ctx.layer_read(0)
        r0 = ctx.merkle_commit(0)                              # top of tree 0 pending
        g = (f.astype(np.uint64) * 3 % P).astype(np.uint32)
        ctx.layer_write(1, g)
        r1 = ctx.merkle_commit(1)                              # another tree: tree 0's top must reach the device first
        n0, n1 = orc.merkle_build(f), orc.merkle_build(g)
        assert r0 == bytes(n0[0]) and r1 == bytes(n1[0])
        for idx in (0, 1, 2, 200, 254, 255, 511, 4000):         # top 8 levels (host-built) and below
            assert ctx.merkle_node(0, idx) == bytes(n0[idx]) and ctx.merkle_node(1, idx) == bytes(n1[idx])
        h = (f.astype(np.uint64) * 5 % P).astype(np.uint32)
        ctx.layer_write(1, h)
        r1b = ctx.merkle_commit(1)                             # the same tree again, twice: the older pending top is dropped
        r1c = ctx.merkle_commit(1)
        n2 = orc.merkle_build(h)
        assert r1b == r1c == bytes(n2[0])
        assert ctx.merkle_path(1, 12345) == [bytes(x) for x in orc.merkle_trace(n2, 12345)]
        assert ctx.merkle_path(0, 77) == [bytes(x) for x in orc.merkle_trace(n0, 77)]
        r0b = ctx.merkle_commit(0)                             # pending, then a whole proof on the same context
        assert r0b == r0
        want = orc.prove(12, 3, want_vectors=False)
        p = ctx.prove()
        assert p.data == want.proof and p.state == want.state
        assert ctx.merkle_node(0, 3) == bytes(n0[3])           # the proof rebuilt tree 0 from the same trace
        # the shape of bench.py's configs[1] loop -- lde; commit(0) over and over, the pending top of the previous
        # iteration dropped every time, an lde in between -- then the first reader must still see a whole tree; and the loop as
        # the benchmark runs it since round 6 (the top's copy ordered on the stream every iteration: Context.stream)
        for settle in (False, True):
            for _ in range(4):
                ctx.lde()
                rr = ctx.merkle_commit(0)
                if settle:
                    ctx.stream
            assert rr == bytes(n0[0])
            ctx.lde()                                          # a transform between the commitment and its first reader
            assert ctx.merkle_path(0, 4097) == [bytes(x) for x in orc.merkle_trace(n0, 4097)]
            for idx in (0, 1, 127, 254, 255, 256, 510, 511):   # both sides of the hand-over depth
                assert ctx.merkle_node(0, idx) == bytes(n0[idx]), idx


def test_caller_allocated_structs_carry_their_size(zk):
    """include/zkstark_amd.h, "ABI version and caller-allocated structs", on the entry points that need a context: a
    zk_kernel_stat array or a zk_transcript_info whose struct_size is unset (a caller compiled against round 5's header) is
    refused with ZK_ERR_INVALID before anything is written; sized ones work (every other test goes through them)."""
    import ctypes as C
    from zkstark_amd import _lib
    lib = _lib.load()
    with zk.Context(10, 3) as ctx:
        ctx.prove(zk.trace_fibsq(1023))
        arr = (_lib.KernelStat * len(_lib.KERNEL_CLASSES))()           # ctypes arrays do not run __init__: struct_size = 0
        assert lib.zk_kernel_stats(ctx._h, arr, len(arr), 0) == -1 and "zk_kernel_stat.struct_size is 0" in lib.zk_last_error().decode()
        assert lib.zk_dev_kernel_stats(arr, len(arr), 0) == -1
        info = _lib.TranscriptInfo()
        info.struct_size = C.sizeof(info) - 4
        info.free_term = 4242
        assert lib.zk_last_transcript(ctx._h, C.byref(info)) == -1 and info.free_term == 4242
        assert "zk_transcript_info.struct_size" in lib.zk_last_error().decode()
        good = ctx.last_transcript()
        assert good.struct_size == C.sizeof(good) and good.public_last == 2338775057
        pr = _lib.ChainProbe()
        pr.struct_size = 0
        assert lib.zk_probe_hash_chain(0, 0, 4, 1, 1, C.byref(pr)) == -1
        st = ctx.kernel_stats()
        assert set(st) == set(_lib.KERNEL_CLASSES)


@pytest.mark.parametrize("q", [2, 7, 64])
def test_prover_multi_query(zk, orc, q):
    """q decommitment queries: proof bytes equal to the oracle's, strict verifier accepts.
    (q = 7 and 64: the decommitment launch runs with several workgroups, the last of which raises the flag.)"""
    try:
        orc.set_queries(q)
        want = orc.prove(10, 3, want_vectors=False)
    finally:
        orc.set_queries(1)
    with zk.Context(10, 3, queries=q) as ctx:
        proof = ctx.prove(zk.trace_fibsq(1023))
    assert proof.data == want.proof and proof.state == want.state
    proof.verify(strict=True)


def test_trace_fibsq_batch_on_device(zk, orc):
    """batch trace generation, one lane per trace, equals the host recurrence (prover.rs:32-39)."""
    a1s = [3141592, 7, 99, P - 1, 0, 123456789] * 20
    got = zk.trace_fibsq_batch([1] * len(a1s), a1s, 1023)
    assert got[0][1022] == 2338775057                        # prover.rs:42
    for t in (0, 1, 3, 5, 119):
        assert np.array_equal(got[t], orc.trace_fibsq(1023, 1, a1s[t]))


def test_repeated_proofs_are_identical(zk):
    """Determinism under load (a latent LDS or hand-off race would show as a differing proof):
    many proofs from one resident trace, two contexts proving concurrently from two host threads."""
    import threading
    a = zk.trace_fibsq((1 << 15) - 1)
    with zk.Context(15, 3) as c1, zk.Context(15, 3) as c2:
        c1.trace_upload(a); c2.trace_upload(a)
        ref = c1.prove()
        out = {}

        def work(name, c):
            out[name] = [c.prove().data for _ in range(40)]
        th = [threading.Thread(target=work, args=(n, c)) for n, c in (("a", c1), ("b", c2))]
        [t.start() for t in th]
        [t.join() for t in th]
    assert all(d == ref.data for d in out["a"] + out["b"])
    b = zk.trace_fibsq((1 << 21) - 1)
    with zk.Context(21, 3) as c3:
        c3.trace_upload(b)
        first = c3.prove()
        assert all(c3.prove().data == first.data for _ in range(6))
    first.verify(strict=True)


def test_c_abi_from_plain_c(tmp_path):
    """The boundary from a C program (gcc, no Python/torch in the process): examples/prove_c_abi.c proves and
    verifies the reference-size instance and prints the reference's own outputs (main.rs:24-35)."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "prove_c_abi")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "prove_c_abi.c"),
                           "-L" + os.path.join(root, "zkstark_amd"), "-lzkstark_amd",
                           "-Wl,-rpath," + os.path.join(root, "zkstark_amd"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    canon = json.load(open(os.path.join(root, "tests", "golden", "stark101_canonical.json")))["derived"]
    assert "a[n-2] = 2338775057" in out.stdout                     # prover.rs:42
    assert "Proof size: 7884" in out.stdout                        # proof.rs:151-154
    head = " ".join(canon["proof_hex"][2 * i:2 * i + 2] for i in range(8))
    assert "proof head: " + head in out.stdout
    assert "final state: " + " ".join(canon["final_state"][2 * i:2 * i + 2] for i in range(8)) in out.stdout


def test_batch_c_abi_from_plain_c(tmp_path):
    """examples/batch_c_abi.c: 64 proofs of the reference's size in one zk_batch_prove, each accepted by the
    verifier; proof 0 (the reference's own trace) starts with the golden proof's bytes."""
    import json
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "batch_c_abi")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "batch_c_abi.c"),
                           "-L" + os.path.join(root, "zkstark_amd"), "-lzkstark_amd",
                           "-Wl,-rpath," + os.path.join(root, "zkstark_amd"), "-o", exe])
    out = subprocess.run([exe, "6"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    canon = json.load(open(os.path.join(root, "tests", "golden", "stark101_canonical.json")))["derived"]
    assert "a[n-2] of proof 0 = 2338775057" in out.stdout
    assert "all 64 proofs verified; proof size 7884" in out.stdout
    assert "proof 0 head: " + " ".join(canon["proof_hex"][2 * i:2 * i + 2] for i in range(8)) in out.stdout


@pytest.mark.parametrize("log_m", [4, 10, 14])
def test_ntt_edge_values(zk, orc, log_m):
    """Field edge cases through the butterflies: 0, 1, P-1 (a + b overflows u32 since P > 2^31), deltas, constants."""
    m = 1 << log_m
    root = orc.gen_of_order_log(log_m)
    vecs = [np.zeros(m, dtype=np.uint32), np.full(m, P - 1, dtype=np.uint32), np.ones(m, dtype=np.uint32)]
    d = np.zeros(m, dtype=np.uint32); d[1] = P - 1
    alt = np.where(np.arange(m) % 2 == 0, P - 1, 1).astype(np.uint32)
    half = np.full(m, (P - 1) // 2 + 1, dtype=np.uint32)      # 2 * half = P + 1: wraps exactly once
    for x in vecs + [d, alt, half]:
        assert np.array_equal(zk.ntt(x), orc.ntt(x, root))
        assert np.array_equal(zk.ntt(x, inverse=True), orc.intt(x, root))


def test_lde_edge_traces(zk, orc):
    """Traces of all zeros / all P-1 (the virtual point and coset scaling with extreme operands)."""
    for fill in (0, P - 1, 1):
        t = np.full(255, fill, dtype=np.uint32)
        assert np.array_equal(zk.lde(t, 8, 3), orc.lde(t, 8, 3))


def test_compose_and_fold_edge_challenges(zk, orc):
    """Raw challenges 0, P (== 0), P+1, 2^32-1 (field.rs:20-24 reduction) and extreme layer values."""
    log_n, log_b = 7, 3
    N = 1 << (log_n + log_b)
    rng = np.random.default_rng(11)
    for f in (np.zeros(N, dtype=np.uint32), np.full(N, P - 1, dtype=np.uint32), rand_field(rng, N)):
        for alphas in ([0, 0, 0], [P, P + 1, 2**32 - 1], [1, P - 1, 5]):
            with zk.Context(log_n, log_b) as ctx:
                trace = rand_field(rng, (1 << log_n) - 1); trace[0] = 1
                ctx.trace_upload(trace)
                ctx.layer_write(0, f)
                ctx.compose(alphas)
                cp = ctx.layer_read(1)
                assert np.array_equal(cp, orc.compose(f, log_n, log_b, alphas, int(trace[-1])))
                for r, beta in enumerate((0, P, 2**32 - 1, 1)):
                    ctx.fri_fold(r, beta)
                    nxt = ctx.layer_read(2 + r)
                    assert np.array_equal(nxt, orc.fri_fold_eval(cp if r == 0 else prev, log_n, log_b, r, beta))
                    prev = nxt


# ---- device / host division of the latency-bound end (zk_ctx_set_host_levels) ------------------
@pytest.mark.parametrize("log_n,log_b,levels", [
    (10, 3, (0, 0)), (10, 3, (8, 9)), (10, 3, (5, 5)), (10, 3, (6, 10)), (10, 3, (10, 10)), (10, 3, (8, 0)),
    (10, 3, (1, 1)), (10, 3, (2, 4)), (10, 3, (8, 7)),
    (5, 3, (8, 9)), (7, 2, (8, 9)), (4, 1, (3, 4)), (2, 1, (1, 1)),
    (15, 3, (8, 9)), (15, 3, (0, 0)), (16, 2, (7, 9)), (18, 1, (9, 9)), (11, 5, (8, 9)),
    # hand-over depths 9 and 10: a team of 2 / 4 threads reduces the tree tops; the host tail may be smaller than the top
    (15, 3, (10, 9)), (17, 3, (9, 9)), (18, 2, (10, 8)), (12, 3, (10, 0)), (9, 2, (10, 4))])
def test_prover_host_levels_identical(zk, orc, log_n, log_b, levels):
    """Whatever part of the tree tops / small FRI layers the host thread takes over, the proof is the
    oracle's, and the device arrays afterwards hold the complete trees and layers (merkle.rs:14-79)."""
    want = orc.prove(log_n, log_b, want_vectors=False)
    with zk.Context(log_n, log_b, host_levels=levels) as ctx:
        assert ctx.host_levels == levels
        proof = ctx.prove(zk.trace_fibsq((1 << log_n) - 1))
        assert proof.data == want.proof and proof.state == want.state
        assert ctx.prove().data == want.proof                         # staging buffers are reusable
        R = log_n
        for layer in sorted({0, 1, 2, max(1, R - 7), max(1, R - 6), max(1, R - 5), R, R + 1}):
            m = ctx.layer_size(layer)
            idx = sorted({0, 1, 2, 5, 30, 62, 63, 64, 126, 127, 254, 255, 256, 510, 511, 1022, 1023, 2046, m - 2, m - 1, m, 2 * m - 2} & set(range(2 * m - 1)))
            if m <= 1 << 18:
                vals = ctx.layer_read(layer)
                nodes = orc.merkle_build(vals)
                for i in idx:
                    assert ctx.merkle_node(layer, i) == bytes(nodes[i]), (layer, i)
                assert ctx.merkle_path(layer, m - 1) == [bytes(h) for h in orc.merkle_trace(nodes, m - 1)]
            else:
                # big layer (the oracle's tree would dominate the test time): the root is pinned by the proof bytes
                # above; here the region the host built and copied back must be consistent with the nodes below it
                # (merkle.rs:42-45) and a path from the last leaf must lead to that root (merkle.rs:82-110)
                for i in [j for j in idx if 2 * j + 2 < 2 * m - 1]:
                    assert ctx.merkle_node(layer, i) == hashlib.sha256(ctx.merkle_node(layer, 2 * i + 1) + ctx.merkle_node(layer, 2 * i + 2)).digest(), (layer, i)
                leaf = int(ctx.layer_read(layer, m - 1, 1)[0])
                assert zk.compute_root_from_path(leaf, m - 1, ctx.merkle_path(layer, m - 1)) == ctx.merkle_node(layer, 0)
                assert ctx.merkle_node(layer, 0) == bytes(ctx.last_transcript().roots[layer])
    proof.verify(strict=True)


def test_host_levels_argument_checks(zk):
    for bad in ((11, 11), (0, 5), (3, 12), (11, 0)):
        with pytest.raises(zk.ZkError):
            zk.Context(6, 2, host_levels=bad).close()
    # the field hash always builds on the device, whatever the setting
    with zk.Context(10, 3, hash="field", host_levels=(8, 9)) as ctx:
        p = ctx.prove(zk.trace_fibsq(1023))
    with zk.Context(10, 3, hash="field", host_levels=(0, 0)) as ctx:
        q = ctx.prove(zk.trace_fibsq(1023))
    assert p.data == q.data
    p.verify(strict=True)


def test_prove_many_contexts_at_once(zk, orc):
    """zk_prove_many: several contexts, one host thread each inside the library; every proof equals the
    oracle's for its own trace, and a failing context is reported by index."""
    sizes = [(12, 3, 99), (10, 3, 3141592), (15, 3, 7), (12, 3, 5)]
    ctxs = [zk.Context(ln, lb) for ln, lb, _ in sizes]
    try:
        for c, (ln, lb, a1) in zip(ctxs, sizes):
            c.trace_upload(zk.trace_fibsq((1 << ln) - 1, 1, a1))
        wants = [orc.prove(ln, lb, 1, a1, want_vectors=False) for ln, lb, a1 in sizes]
        for _ in range(3):
            proofs = zk.prove_many(ctxs)
            for p, want in zip(proofs, wants):
                assert p.data == want.proof and p.state == want.state
        proofs[1].verify(strict=True)
        # the same with every context's early launch on (the header advises against it when contexts share a GPU -- a stream waiting on
        # its gate holds a hardware queue -- but it must stay CORRECT and must not deadlock: a gate only ever waits for work that was
        # enqueued before it), eight contexts on more streams than the device has hardware queues
        more = [zk.Context(ln, lb) for ln, lb, _ in sizes]
        try:
            for c, (ln, lb, a1) in zip(more, sizes):
                c.trace_upload(zk.trace_fibsq((1 << ln) - 1, 1, a1))
            on = [c.set_early_launch(True) for c in ctxs + more]
            for _ in range(5):
                proofs = zk.prove_many(ctxs + more)
                for p, want in zip(proofs, wants + wants):
                    assert p.data == want.proof and p.state == want.state
            assert all(on) or not any(on)
            for c in ctxs:
                c.set_early_launch(False)
        finally:
            for c in more:
                c.close()
        bad = zk.trace_fibsq(4095, 1, 5)
        bad[77] = (int(bad[77]) + 1) % P
        ctxs[3].trace_upload(bad)
        with pytest.raises(zk.ZkError, match="context 3"):
            zk.prove_many(ctxs)
        with pytest.raises(zk.ZkError):
            zk.prove_many([ctxs[0], ctxs[0]])
    finally:
        for c in ctxs:
            c.close()


# ---- batched proving ----------------------------------------------------------
@pytest.mark.parametrize("log_n,log_b,log_batch", [(10, 3, 0), (10, 3, 1), (10, 3, 3), (6, 2, 4), (4, 1, 2), (2, 1, 3), (12, 3, 2),
                                                   (7, 4, 5), (5, 5, 1), (14, 2, 1), (9, 3, 6)])
def test_batch_prover_matches_oracle(zk, orc, log_n, log_b, log_batch):
    """Every proof of a batch equals the oracle's proof for its own trace (its own transcript and
    challenges); traces generated on the device from per-proof seeds (prover.rs:32-39)."""
    batch = 1 << log_batch
    a0s = [1] * batch                                        # prover.rs:105: the first constraint pins a[0] = 1
    a1s = [3141592 + 977 * p for p in range(batch)]
    with zk.BatchContext(log_n, log_b, log_batch) as bc:
        bc.gen_fibsq(a0s, a1s)
        proofs = bc.prove()
        again = bc.prove()                                   # resident traces, buffers reusable
    check = sorted({0, min(1, batch - 1), batch // 2, batch - 1})
    for p in range(batch):
        assert proofs[p].data == again[p].data
    for p in check:
        want = orc.prove(log_n, log_b, a0s[p], a1s[p], want_vectors=False)
        assert want.rc == 0
        assert proofs[p].data == want.proof and proofs[p].state == want.state, f"proof {p}"
        assert proofs[p].public_last == want.public_last
        proofs[p].verify(strict=True)
    for p in range(batch):
        proofs[p].verify()


def test_batch_prover_at_the_benchmark_domain_2e24(zk, orc):
    """Throughput mode at the metric's own domain (bench.py: batched_2e24): two 2^24 proofs in lockstep.  Proof 0 is compared with
    the ORACLE byte for byte (the oracle proves 2^24 in a few seconds), proof 1 with zk_prove of its own trace (that prover is
    itself oracle-checked at this size: test_config3_*), and both pass the strict verifier (prover.rs:9-293 per proof)."""
    log_n, log_b = 21, 3
    seeds = [3141592, 3141593]
    with zk.BatchContext(log_n, log_b, 1) as bc:
        bc.gen_fibsq([1, 1], seeds)
        proofs = bc.prove()
    want = orc.prove(log_n, log_b, 1, seeds[0], want_vectors=False)
    assert want.rc == 0 and proofs[0].data == want.proof and proofs[0].state == want.state
    with zk.Context(log_n, log_b) as ctx:
        one = ctx.prove(zk.trace_fibsq((1 << log_n) - 1, 1, seeds[1]))
    assert proofs[1].data == one.data and proofs[1].state == one.state and proofs[1].public_last == one.public_last
    for p in proofs:
        p.verify(strict=True)


def test_batch_setters_are_refused_while_a_prove_runs(zk):
    """One batch is used from one host thread at a time; a setter that arrives from another thread while zk_batch_prove runs
    is refused with ZK_ERR_STATE instead of replacing the pool / buffers under the running proof, and the
    proofs of that run are unharmed."""
    import threading
    from zkstark_amd import _lib
    lib = _lib.load()
    log_n, log_b, log_batch = 16, 3, 4
    a1s = [3141592 + p for p in range(1 << log_batch)]
    with zk.BatchContext(log_n, log_b, log_batch) as bc:
        bc.gen_fibsq([1] * len(a1s), a1s)
        first, _ = bc.prove_raw()
        seen, out, stop = [], {}, threading.Event()

        def prover():
            done = 0
            while done < 6:
                try:
                    out["last"] = bc.prove_raw()[0]
                    done += 1
                except zk.ZkError as e:                     # a setter of the other thread held the batch at that instant
                    assert e.code == -4
                    seen.append(-4)
            stop.set()

        t = threading.Thread(target=prover)
        t.start()
        while not stop.is_set():
            seen.append(lib.zk_batch_set_threads(bc._h, 3))
            seen.append(lib.zk_batch_set_queries(bc._h, 1))
        t.join()
        assert -4 in seen                                   # ZK_ERR_STATE at least once while a prove was in flight
        assert set(seen) <= {0, -4}
        assert (out["last"] == first).all()
        assert lib.zk_batch_set_threads(bc._h, 3) == 0      # idle again: accepted
        again, _ = bc.prove_raw()
        assert (again == first).all()


@pytest.mark.parametrize("hash_name,q", [("sha256", 3), ("field", 1), ("field", 2)])
def test_batch_prover_queries_and_field_hash(zk, orc, hash_name, q):
    """The batch honours the same settings as a context: q decommitment queries, field-native Merkle hash."""
    log_n, log_b, log_batch = 8, 3, 2
    a1s = [11 + p for p in range(1 << log_batch)]
    with zk.BatchContext(log_n, log_b, log_batch, hash=hash_name, queries=q) as bc:
        bc.gen_fibsq([1] * len(a1s), a1s)
        proofs = bc.prove()
    try:
        orc.set_hash(1 if hash_name == "field" else 0)
        orc.set_queries(q)
        for p in (0, 3):
            want = orc.prove(log_n, log_b, 1, a1s[p], want_vectors=False)
            assert want.rc == 0 and proofs[p].data == want.proof and proofs[p].state == want.state
    finally:
        orc.set_hash(0)
        orc.set_queries(1)
    for pr in proofs:
        pr.verify(strict=True)


def test_batch_prover_host_traces_and_single_context(zk):
    """Traces handed over from the host; each proof equals Context.prove() of the same trace."""
    log_n, log_b, log_batch = 8, 3, 3
    traces = np.stack([zk.trace_fibsq((1 << log_n) - 1, 1, 5 + p) for p in range(1 << log_batch)])
    with zk.BatchContext(log_n, log_b, log_batch) as bc:
        bc.set_traces(traces)
        proofs = bc.prove()
        assert list(bc.public_last()) == [int(t[-1]) for t in traces]
    with zk.Context(log_n, log_b) as ctx:
        for p in (0, 3, 7):
            one = ctx.prove(traces[p])
            assert one.data == proofs[p].data and one.state == proofs[p].state


def test_batch_prover_rejects_a_bad_trace(zk):
    traces = np.stack([zk.trace_fibsq(255, 1, 9 + p) for p in range(4)])
    traces[2, 100] = (int(traces[2, 100]) + 1) % P
    with zk.BatchContext(8, 2, 2) as bc:
        bc.set_traces(traces)
        with pytest.raises(zk.ZkError, match="proof 2"):
            bc.prove()
        with pytest.raises(zk.ZkError):
            bc.set_traces(traces[:3])
    with pytest.raises(zk.ZkError):
        zk.BatchContext(10, 3, 11)
    with pytest.raises(zk.ZkError):
        zk.BatchContext(10, 0, 2)
    # the batch accepts exactly the sizes a context accepts (everything it proves must be verifiable)
    for log_n, log_b in ((1, 1), (1, 3), (3, 3), (28, 3)):
        with pytest.raises(zk.ZkError) as e1:
            zk.BatchContext(log_n, log_b, 1)
        with pytest.raises(zk.ZkError) as e2:
            zk.Context(log_n, log_b)
        assert e1.value.code == e2.value.code == -1


def test_maximum_domain_2e30(zk):
    """The largest domain the field admits (n * B must divide 2^30; SURVEY 8: N <= 2^30): 225 GB of layers and
    full trees on one MI355X.  No oracle at this size: two proofs from one resident trace are identical and the
    transcript-replaying verifier accepts (proof.rs:15-149 generalised)."""
    import torch
    torch.cuda.empty_cache()
    free, total = torch.cuda.mem_get_info()
    if free < 232 * 10**9:
        pytest.skip(f"needs 232 GB of free device memory, {free / 1e9:.0f} GB available")
    log_n = 27
    try:
        ctx = zk.Context(log_n, 3)
    except zk.ZkError as e:
        if e.code == -3:
            pytest.skip("device memory exhausted")
        raise
    with ctx:
        ctx.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        p = ctx.prove()
        q = ctx.prove()
        assert ctx.device_bytes > 200 * 10**9
    assert p.data == q.data and p.state == q.state
    assert len(p.data) == 34968
    p.verify(strict=True)
    with pytest.raises(zk.ZkError):
        zk.Context(28, 3)                                    # n * B = 2^31 does not divide P - 1 = 3 * 2^30


@pytest.mark.parametrize("log_n,log_b,hash_name", [(4, 3, "sha256"), (10, 3, "sha256"), (13, 2, "sha256"), (17, 3, "sha256"), (12, 3, "field"), (18, 1, "sha256")])
def test_early_launch_gives_the_same_proof(zk, orc, log_n, log_b, hash_name):
    """zk_ctx_set_early_launch: the next FRI round's launches enqueued before the current commitment is waited for, released by one
    store once the challenge is drawn (a command-processor wait on a host word; the round's constant read from pinned host memory).
    Same bytes as the oracle, proof after proof, with the switch flipped between proofs and with several queries."""
    if hash_name == "field":
        orc.set_hash(orc.HASH_FIELD)
    try:
        want = orc.prove(log_n, log_b, want_vectors=False)
    finally:
        orc.set_hash(orc.HASH_SHA256)
    with zk.Context(log_n, log_b, hash=hash_name) as ctx:
        ctx.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        supported = ctx.set_early_launch(True)
        if not supported:
            pytest.skip("the device has no stream memory operations")
        for _ in range(3):
            p = ctx.prove()
            assert p.data == want.proof and p.state == want.state
        ctx.set_early_launch(False)
        p = ctx.prove()
        assert p.data == want.proof
        ctx.set_early_launch(True)
        p = ctx.prove()
        assert p.data == want.proof and p.state == want.state
        p.verify(strict=True) if hash_name == "sha256" else p.verify()


def test_early_launch_survives_a_failed_proof(zk, orc):
    """A proof that ends with an error while the next round's launches sit behind the gate (a trace that breaks the recurrence:
    prover.rs:238 fails at the last layer) must release them: the context proves correctly afterwards and nothing hangs."""
    want = orc.prove(12, 3, want_vectors=False)
    good = zk.trace_fibsq((1 << 12) - 1)
    bad = good.copy()
    bad[1000] = (int(bad[1000]) + 1) % P
    with zk.Context(12, 3) as ctx:
        if not ctx.set_early_launch(True):
            pytest.skip("the device has no stream memory operations")
        for _ in range(2):
            with pytest.raises(zk.ZkError):
                ctx.prove(bad)
            p = ctx.prove(good)
            assert p.data == want.proof and p.state == want.state
        ctx.sync()
