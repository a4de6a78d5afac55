"""The native sharded prover (zk_shard_*, csrc/shard.hip) on real hardware, through the C ABI.

* world = 1 with the built-in RCCL transport and the collectives forced: grouped ncclSend/ncclRecv all-to-all,
  ncclAllGather, the chunked exchange on the side stream -- everything a one-GPU box lets RCCL do;
* world = 2 / 4 as processes sharing the one GPU, with a caller-supplied transport (gloo, host-staged): the same
  orchestration code, every rank's proof bit-exact against the CPU oracle;
* BASELINE.json configs[3]: domain 2^26 evaluated by 2 ranks + all-to-all transpose + commit, root anchored on
  the oracle (orc.lde + orc.merkle_build).
"""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize("log_n,log_b,opts", [
    (10, 3, dict(min_layer_log=1, min_chunk_log=4)),
    (13, 3, dict(min_layer_log=1, min_chunk_log=6, overlap_min_log=10)),      # chunked exchange + commit_finish
    (16, 2, dict(min_layer_log=12, min_chunk_log=6, overlap_min_log=12)),
    (18, 3, dict()),                                                           # the production thresholds
])
def test_shard_world1_rccl_matches_oracle(zk, orc, log_n, log_b, opts):
    """One rank, RCCL transport, collectives forced: proof bytes == oracle; the byte counters see no peer."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    want = orc.prove(log_n, log_b, want_vectors=False, want_roots=True)
    uid = zk.shard_unique_id()
    with zk.ShardContext(log_n, log_b, 0, 1, uid, force_collectives=True, **opts) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        info = sp.last_transcript()
        st = sp.stats()
        again = sp.prove()
        ch = zk.Channel(); ch.commit(b"prefix")
        pref = sp.prove_channel(ch)
    assert proof.data == want.proof and proof.state == want.state
    assert again.data == proof.data
    for t in range(log_n + 2):
        assert bytes(info.roots[t]) == bytes(want.roots[t]), t
    assert list(info.beta_raw)[:log_n] == want.beta_raw and info.free_term == want.free_term
    assert st["native_rccl"] == 1 and st["sent_bytes"] == 0 and st["sharded_layers"] >= 1
    assert st["selftest_ok"] == 1                            # zk_shard_create's known-pattern exchange through RCCL
    if "overlap_min_log" in opts:
        assert st["chunked_layers"] >= 2
        assert st["communicators"] == 2                      # the chunked exchanges run on their own communicator
    else:
        assert st["communicators"] == (2 if zk.shard_plan(1, log_n, log_b, force_collectives=True, **opts)["chunked_mask"] else 1)
    assert (pref.data, pref.state) == orc.prove_prefixed(b"prefix", log_n, log_b)
    proof.verify(strict=True)


@pytest.mark.parametrize("opts,want_comms", [
    (dict(), 2),                                  # chunked exchange on its own communicator (the default)
    (dict(single_communicator=True), 1),          # ... sharing the main one, ordered by the collective events alone
    (dict(plain_collectives=True), 1),            # the fall-back rung of bench.py: nothing in chunks, roots by all-gather
    (dict(single_build_stream=True), 2),
])
def test_shard_options_profiling_and_self_test(zk, orc, opts, want_comms):
    """zk_shard_options' operational fields (round 4: they replace the ZK_SHARD_* environment switches), the exchange
    timing of zk_shard_set_profiling and zk_shard_self_test, one rank over RCCL; the proof bytes never change."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log_n, log_b = 14, 3
    want = orc.prove(log_n, log_b, want_vectors=False)
    with zk.ShardContext(log_n, log_b, 0, 1, zk.shard_unique_id(), force_collectives=True, min_layer_log=1, min_chunk_log=6,
                         overlap_min_log=10, timeout_s=15.0, **opts) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        st0 = sp.stats()
        sp.set_profiling(True)
        again = sp.prove()
        st1 = sp.stats()
        sp.self_test()
        sp.set_profiling(False)
        last = sp.prove()
        root = sp.lde_commit()
    assert proof.data == want.proof == again.data == last.data and proof.state == want.state
    assert st0["communicators"] == want_comms and st0["selftest_ok"] == 1 and st0["selftest_ms"] > 0
    plain = bool(opts.get("plain_collectives"))
    assert (st0["chunked_layers"] == 0) == plain and st0["root_board"] == (0 if plain else 1)
    assert st0["exchange_ms"] == 0 and st0["exchanges"] == 0                     # profiling off: nothing timed
    assert st1["exchanges"] >= st1["sharded_layers"] + 1 and st1["exchange_ms"] > 0 and st1["tail_ms"] > 0
    assert 0 <= st1["exposed_exchange_ms"] <= st1["exchange_ms"] * 1.5 + 1.0
    assert root == bytes(orc.prove(log_n, log_b, want_vectors=False, want_roots=True).roots[0])


def test_shard_self_test_catches_a_transport_that_loses_data(zk):
    """A transport whose all-to-all reports success without moving anything: zk_shard_create must fail in its self-test,
    naming the exchange, instead of producing proofs from a zero-filled receive buffer."""
    from zkstark_amd import _lib
    lazy = _lib.ShardTransport(None, _lib.ALL_TO_ALL_FN(lambda *a: 0), _lib.ALL_GATHER_FN(lambda *a: 0))
    with pytest.raises(zk.ZkError) as e:
        zk.ShardContext(12, 3, 0, 1, os.urandom(128), transport=lazy, force_collectives=True, min_layer_log=1, min_chunk_log=5)
    assert e.value.code == -2 and "self-test" in str(e.value)


@pytest.mark.parametrize("hash_name,queries", [("field", 1), ("sha256", 3), ("field", 4)])
def test_shard_world1_rccl_hash_and_queries(zk, orc, hash_name, queries):
    """zk_shard_set_hash / zk_shard_set_queries (the sharded prover was SHA-256 and single-query only): proof bytes ==
    the oracle's with the same settings, the verifier with those settings accepts."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log_n, log_b = 13, 3
    orc.set_hash(orc.HASH_FIELD if hash_name == "field" else orc.HASH_SHA256)
    orc.set_queries(queries)
    try:
        want = orc.prove(log_n, log_b, want_vectors=False)
    finally:
        orc.set_hash(orc.HASH_SHA256)
        orc.set_queries(1)
    with zk.ShardContext(log_n, log_b, 0, 1, zk.shard_unique_id(), force_collectives=True, min_layer_log=1, min_chunk_log=6,
                         overlap_min_log=10, hash=hash_name, queries=queries) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        st = sp.stats()
    assert proof.data == want.proof and proof.state == want.state
    assert st["native_rccl"] == 1 and st["rccl_nranks"] == 1
    proof.verify(strict=(hash_name == "sha256" and queries == 1))


def test_shard_failed_rank_refuses_and_aborts_its_communicator(zk):
    """A rank that has left a proof with an error refuses further proofs and tears its RCCL communicator down with
    ncclCommAbort (zk_shard_destroy), not ncclCommDestroy."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sp = zk.ShardContext(12, 3, 0, 1, zk.shard_unique_id(), force_collectives=True, min_layer_log=1, min_chunk_log=5)
    sp.trace_upload(zk.trace_fibsq((1 << 12) - 1))
    good = sp.prove()
    assert sp.inject_failure(-4) == -4
    with pytest.raises(zk.ZkError) as e:
        sp.prove()
    assert e.value.code == -4 and "earlier proof" in str(e.value)
    sp.close()                                            # ncclCommAbort path
    with zk.ShardContext(12, 3, 0, 1, zk.shard_unique_id(), force_collectives=True, min_layer_log=1, min_chunk_log=5) as sp2:
        sp2.trace_upload(zk.trace_fibsq((1 << 12) - 1))   # a fresh communicator works after the abort
        assert sp2.prove().data == good.data


def test_shard_world1_local_no_collectives(zk, orc):
    """world = 1 without forcing: no transport call at all (RCCL is not even loaded for it)."""
    from zkstark_amd import _lib
    boom = _lib.ShardTransport(None, _lib.ALL_TO_ALL_FN(lambda *a: 7), _lib.ALL_GATHER_FN(lambda *a: 7))
    want = orc.prove(12, 3, want_vectors=False)
    with zk.ShardContext(12, 3, 0, 1, None, transport=boom, min_layer_log=1, min_chunk_log=5) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << 12) - 1))
        proof = sp.prove()
    assert proof.data == want.proof and proof.state == want.state


def test_shard_argument_checks(zk):
    from zkstark_amd import _lib
    t = _lib.ShardTransport(None, _lib.ALL_TO_ALL_FN(lambda *a: 0), _lib.ALL_GATHER_FN(lambda *a: 0))
    for world, rank, log_n, log_b in ((3, 0, 12, 3), (16, 0, 12, 3), (2, 2, 12, 3), (2, 0, 3, 3), (4, 0, 4, 2)):
        with pytest.raises(zk.ZkError) as e:
            zk.ShardContext(log_n, log_b, rank, world, None, transport=t)
        assert e.value.code == -1
    with pytest.raises(zk.ZkError):                       # the RCCL transport needs the shared id
        zk.ShardContext(12, 3, 0, 1, None)
    with zk.ShardContext(12, 3, 0, 1, None, transport=t, min_layer_log=1, min_chunk_log=5) as sp:
        with pytest.raises(zk.ZkError) as e:              # prove without a trace
            sp.prove()
        assert e.value.code == -4
        with pytest.raises(zk.ZkError):
            sp.trace_upload(np.ones(5, dtype=np.uint32))


def _worker(rank, world, port, log_n, log_b, opts, q, mode, uid):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    __import__("mp_util").init_pg("gloo", port, rank, world)
    try:
        import zkstark_amd as zk
        from sharded_testlib import gloo_transport
        torch.cuda.set_device(0)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # the caller's transport (gloo, host-staged), or -- peer_copy -- the library's own peer copies through IPC handles
        tp = None if opts.get("peer_copy") else gloo_transport()
        with zk.ShardContext(log_n, log_b, rank, world, uid, transport=tp, timeout_s=30.0, **opts) as sp:
            sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
            if mode == "fail":                             # the last rank leaves the proof with an error: nobody may hang
                import time
                if rank == world - 1:
                    sp.inject_failure()
                t0 = time.time()
                try:
                    sp.prove()
                    q.put((rank, "proved", "", time.time() - t0))
                except zk.ZkError as e:
                    q.put((rank, "error", str(e), time.time() - t0))
            elif mode == "prove":
                proof = sp.prove()
                info = sp.last_transcript()
                q.put((rank, proof.data, proof.state, [bytes(r) for r in info.roots[:log_n + 2]], sp.stats()))
            else:
                root = sp.lde_commit()
                again = sp.lde_commit()
                q.put((rank, root, again, sp.layer_read(0, 0, 4).tolist(), sp.stats()))
    finally:
        dist.destroy_process_group()


def _run(world, log_n, log_b, opts, mode, timeout=240):
    import torch.multiprocessing as mp
    port = _free_port()
    uid = os.urandom(128)                                  # names the shared-memory root board (no RCCL here)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, log_b, opts, q, mode, uid)) for r in range(world)]
    for p in procs:
        p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results
    out = sorted(gather_results(q, procs, world, timeout), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return out


def test_shard_two_ranks_fieldhash_three_queries(orc):
    """Two ranks on one GPU with hash = field and three queries (zk_shard_set_hash / _set_queries) against the oracle."""
    log_n, log_b, world = 12, 3, 2
    orc.set_hash(orc.HASH_FIELD)
    orc.set_queries(3)
    try:
        want = orc.prove(log_n, log_b, want_vectors=False)
    finally:
        orc.set_hash(orc.HASH_SHA256)
        orc.set_queries(1)
    out = _run(world, log_n, log_b, dict(min_layer_log=1, min_chunk_log=6, hash="field", queries=3), "prove")
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank}"


def test_shard_failure_of_one_rank_is_contained(threads_check):
    """Four ranks (threads of one process); rank 2 leaves the proof with an error (zk_shard_inject_failure).  Every other
    rank must come back with an error within seconds -- not hang in a collective, a board exchange or a barrier."""
    import subprocess
    out = subprocess.run([threads_check, "4", "14", "3", "1", "5", "99", "0", "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 3, out.stdout + out.stderr
    assert "failure contained: every rank returned an error" in out.stdout
    assert "injected failure on rank 2" in out.stderr


@pytest.mark.parametrize("world,log_n,log_b,opts", [
    (2, 12, 3, dict(min_layer_log=1, min_chunk_log=6)),
    (4, 14, 3, dict(min_layer_log=1, min_chunk_log=6)),
    (2, 16, 3, dict(min_layer_log=1, min_chunk_log=6, overlap_min_log=10)),          # chunked exchange, 2 ranks
    (4, 16, 2, dict(min_layer_log=10, min_chunk_log=6, overlap_min_log=10)),         # world = blow-up: local blow-up 1
    (2, 14, 3, dict(min_layer_log=1, min_chunk_log=6, no_root_board=True)),          # roots by all-gather
    (2, 16, 3, dict(min_layer_log=1, min_chunk_log=6, overlap_min_log=10, exchange_cp=True)),   # cp exchanged like every other layer (rounds 1-4)
    (4, 14, 3, dict(min_layer_log=1, min_chunk_log=6, exchange_cp=True)),
])
def test_shard_multirank_one_gpu_matches_oracle(orc, world, log_n, log_b, opts):
    """Every rank's proof, state and all R + 2 roots equal the CPU oracle's; the exchanged volume is what
    DESIGN.md section 6 states: 4 bytes per element per EXCHANGED layer -- f and every distributed FRI layer but cp, which is
    recomputed over the rank's block from the received block of f (exchange_cp: cp too) --, a share (G-1)/G of it to peers."""
    want = orc.prove(log_n, log_b, want_vectors=False, want_roots=True)
    out = _run(world, log_n, log_b, opts, "prove")
    N = 1 << (log_n + log_b)
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank}"
        assert roots == [bytes(r) for r in want.roots], f"rank {rank}"
        ns = st["sharded_layers"]
        assert ns >= 2 and st["root_board"] == (0 if opts.get("no_root_board") else 1) and st["native_rccl"] == 0
        words = N + sum(N >> rho for rho in range(ns))              # f and FRI layers 0 .. ns-1, one all-to-all each ...
        if not opts.get("exchange_cp"):
            words -= N                                                # ... but none for FRI layer 0 = cp
        assert st["all_to_all_bytes"] == 4.0 * words / world * (world - 1) / world
        halo = 0 if opts.get("exchange_cp") else 2 * (1 << log_b)    # the positions after every block: one small all-gather
        gathers = st["sent_bytes"] - st["all_to_all_bytes"]
        assert gathers >= 4.0 * halo * (world - 1)
        if "overlap_min_log" in opts:
            assert st["chunked_layers"] >= 2
        # what ran is what zk_shard_plan announced (the one layout, shared with the mirror)
        import zkstark_amd as zk
        pl = zk.shard_plan(world, log_n, log_b, **{k: v for k, v in opts.items() if k in ("min_layer_log", "min_chunk_log", "overlap_min_log", "exchange_cp")})
        assert pl["cp_from_f"] == (0 if opts.get("exchange_cp") else 1)
        assert (st["sharded_layers"], st["chunked_layers"], st["all_to_all_bytes"]) == (pl["sharded_layers"], pl["chunked_layers"], pl["all_to_all_bytes"])


@pytest.mark.parametrize("world,log_n,log_b,opts", [
    (2, 12, 3, dict(min_layer_log=1, min_chunk_log=6, peer_copy=True)),
    (4, 14, 3, dict(min_layer_log=1, min_chunk_log=6, overlap_min_log=8, peer_copy=True)),     # overlap_min_log is overridden: nothing in chunks
    (2, 18, 3, dict(peer_copy=True)),                                                             # production thresholds
])
def test_shard_peer_copy_transport_matches_oracle(orc, world, log_n, log_b, opts):
    """The built-in PEER-COPY transport (zk_shard_options.peer_copy, csrc/peer.hpp: IPC handles on a shared page, device-to-device
    pulls, no RCCL and no caller transport) with 2 / 4 processes sharing the GPU: every rank's proof, state and roots equal the
    oracle's; plain collectives only (roots by all-gather), the byte counters as planned."""
    want = orc.prove(log_n, log_b, want_vectors=False, want_roots=True)
    out = _run(world, log_n, log_b, opts, "prove")
    import zkstark_amd as zk
    pl = zk.shard_plan(world, log_n, log_b, **{k: v for k, v in opts.items() if k in ("min_layer_log", "min_chunk_log", "overlap_min_log", "peer_copy")})
    assert pl["chunked_layers"] == 0 and pl["overlap_min_log"] == 99
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank}"
        assert roots == [bytes(r) for r in want.roots], f"rank {rank}"
        assert st["peer_copy"] == 1 and st["native_rccl"] == 0 and st["root_board"] == 0 and st["chunked_layers"] == 0 and st["selftest_ok"] == 1
        assert (st["sharded_layers"], st["all_to_all_bytes"]) == (pl["sharded_layers"], pl["all_to_all_bytes"])


def test_shard_peer_copy_failure_of_one_rank_is_contained():
    """Four processes on the peer-copy transport; the last one leaves the proof with an error (zk_shard_inject_failure).  The others
    must come back with an error naming it within seconds -- the abort word on the transport's page ends their waits -- not after
    timeout_s (30 s here) and not never."""
    out = _run(4, 14, 3, dict(min_layer_log=1, min_chunk_log=6, peer_copy=True), "fail", timeout=200)
    for rank, what, msg, dt in out:
        assert what == "error", (rank, what, msg)
        assert dt < 15.0, (rank, dt, msg)
        if rank != 3:                                      # the rank that left, or a peer that had already left because of it
            assert "left the proof with error" in msg, msg
    assert any("rank 3 left" in msg for rank, what, msg, dt in out if rank != 3), out


def test_shard_peer_copy_one_rank_and_a_peer_that_never_comes(zk, orc):
    """One rank with the collectives forced goes through the peer-copy code with itself as the only peer; a rank whose peer never
    arrives gets an error naming what it waited for within timeout_s (not a hang)."""
    want = orc.prove(12, 3, want_vectors=False)
    uid = os.urandom(128)
    with zk.ShardContext(12, 3, 0, 1, uid, force_collectives=True, peer_copy=True, min_layer_log=1, min_chunk_log=4) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << 12) - 1))
        proof = sp.prove()
        assert sp.stats()["peer_copy"] == 1
    assert proof.data == want.proof and proof.state == want.state
    import time
    t0 = time.time()
    with pytest.raises(zk.ZkError) as e:
        zk.ShardContext(12, 3, 0, 2, os.urandom(128), peer_copy=True, timeout_s=2.0, min_layer_log=1, min_chunk_log=4)   # rank 1 does not exist
    assert "peer-copy transport" in str(e.value) and time.time() - t0 < 20


def test_shard_two_ranks_production_sizes_2e25(orc):
    """Two ranks at the per-rank size of the weak-scaling benchmark (2^24 elements each, domain 2^25) with the
    production thresholds: f and the two FRI layers after cp go through the chunked exchange (pieces of 2^23 ... 2^21 words; cp
    itself is recomputed from the received block of f), the 2^21-value layer through a plain one, the rest through the replicated tail.  Every byte of the proof and all 24 roots against the oracle."""
    log_n, world = 22, 2
    want = orc.prove(log_n, 3, want_vectors=False, want_roots=True)
    assert want.rc == 0
    out = _run(world, log_n, 3, {}, "prove", timeout=240)
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank}"
        assert roots == [bytes(r) for r in want.roots], f"rank {rank}"
        assert st["sharded_layers"] == 5 and st["chunked_layers"] == 3 and st["root_board"] == 1
    # the same size on the library's peer-copy transport (round 6): shards whose tree array is 3.2 GB -- the size at which the first
    # version of that transport, which exported the caller's own allocations, never came back from mapping one (docs/LOG.md)
    out = _run(world, log_n, 3, dict(peer_copy=True), "prove", timeout=240)
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank} (peer copy)"
        assert st["sharded_layers"] == 5 and st["chunked_layers"] == 0 and st["peer_copy"] == 1


def test_shard_four_ranks_production_sizes_2e26(orc):
    """Four ranks, 2^24 elements each (domain 2^26, the size of BASELINE.json configs[3]), production thresholds:
    chunked exchange for the pieces of 2^22 and 2^21 words (f and the FRI layer after cp), single exchanges below, replicated
    tail from 2^19 values (layers of >= 2^20 values stay distributed from 4 ranks on)."""
    log_n, world = 23, 4
    want = orc.prove(log_n, 3, want_vectors=False, want_roots=True)
    assert want.rc == 0
    out = _run(world, log_n, 3, {}, "prove", timeout=240)
    N = 1 << (log_n + 3)
    for rank, data, state, roots, st in out:
        assert data == want.proof and state == want.state, f"rank {rank}"
        assert roots == [bytes(r) for r in want.roots], f"rank {rank}"
        assert st["sharded_layers"] == 7 and st["chunked_layers"] == 2 and st["root_board"] == 1
        words = N + sum(N >> rho for rho in range(1, 7))                                # f and FRI layers 1 .. 6; cp comes from the block of f
        assert st["all_to_all_bytes"] == 4.0 * words / world * (world - 1) / world      # 8 N-ish bytes in total, (G-1)/G of it to peers


def test_config4_native_sharded_lde_transpose_commit_2e26(zk, config4_expected):
    """configs[3]: domain 2^26, each of 2 ranks evaluates its cosets (no communication), one all-to-all
    transposes to natural order (chunked, overlapped with the hashing), subtrees + host top.  The root is the
    CPU oracle's (orc.lde + orc.merkle_build), the shards are the oracle's values at i = rank (mod 2)."""
    log_n, world = config4_expected["log_n"], 2
    want_root, want_head = config4_expected["root"], config4_expected["head"]
    out = _run(world, log_n, 3, {}, "lde_commit", timeout=240)
    for rank, root, again, head, st in out:
        assert root == want_root and again == want_root, f"rank {rank}"
        assert head == [int(want_head[rank + world * j]) for j in range(4)]
        assert st["chunked_layers"] == 1 and st["all_to_all_bytes"] == 4.0 * (1 << 26) / world / world


def test_shard_from_plain_c(tmp_path, orc):
    """The sharded prover reached from a C program (gcc + pthreads, no Python/torch in the process): RCCL is loaded by
    the library at run time; the proof equals zk_prove's and the oracle's."""
    import subprocess
    exe = str(tmp_path / "shard_c_abi")
    subprocess.check_call(["gcc", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "shard_c_abi.c"),
                           "-L" + os.path.join(ROOT, "zkstark_amd"), "-lzkstark_amd", "-Wl,-rpath," + os.path.join(ROOT, "zkstark_amd"), "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe, "1", "12", "3"], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    want = orc.prove(12, 3, want_vectors=False)
    assert f"world 1: {len(want.proof)} proof bytes on every rank, equal to zk_prove" in out.stdout
    assert "native rccl 1" in out.stdout
    assert "proof head: " + " ".join(f"{b:02x}" for b in want.proof[:8]) in out.stdout
    # the same program on the library's peer-copy transport (no RCCL): one rank, then two and four ranks as THREADS sharing the GPU
    # (ranks of one process hand each other raw pointers instead of IPC handles)
    for world in ("1", "2", "4"):
        out = subprocess.run([exe, world, "12", "3", "peer"], capture_output=True, text=True, timeout=240, env=dict(env, ZK_EXAMPLE_SHARE_GPU="1"))
        assert out.returncode == 0, out.stdout + out.stderr
        assert f"world {world}: {len(want.proof)} proof bytes on every rank, equal to zk_prove" in out.stdout
        assert "native rccl 0, peer copy 1" in out.stdout


@pytest.fixture(scope="module")
def threads_check(tmp_path_factory):
    import subprocess
    exe = str(tmp_path_factory.mktemp("shard") / "shard_threads_check")
    # one recipe for tests and session scripts: stamped with the library's build hash, so a stale binary refuses to run
    subprocess.check_call(["bash", os.path.join(ROOT, "tools", "build_shard_threads_check.sh"), exe], stdout=subprocess.DEVNULL)
    return exe


@pytest.mark.parametrize("world,log_n,log_b,thresholds", [
    (8, 14, 3, (1, 5, 8)),        # lg = 3: local blow-up 1, three-digest top paths, chunked f / cp / first FRI layer
    (8, 17, 3, (16, 6, 12)),      # ... with a replicated tail that starts early
    (16, 14, 4, (1, 4, 9)),       # lg = 4
    (2, 13, 3, (1, 5, 9)),
    (4, 15, 2, (1, 5, 9)),
    (32, 12, 5, (1, 2, 99)),      # lg = 5 = log2 of the blow-up: 32 ranks, the halo of cp is 2 words per (rank, block)
    (8, 4, 3, (1, 1, 99)),        # n = 2 G: a block is exactly the 2B taps long, pieces of two words
    (4, 4, 2, (1, 1, 99)),
    (2, 2, 3, (1, 1, 99)),
])
def test_shard_ranks_as_threads_of_one_process(threads_check, orc, world, log_n, log_b, thresholds):
    """world = 8 and 16 natively on one GPU: the ranks are threads of one C process (the model of
    examples/shard_c_abi.c) with a device-to-device transport; every rank's proof equals zk_prove's (checked inside the
    harness) and the oracle's."""
    import subprocess
    out = subprocess.run([threads_check, str(world), str(log_n), str(log_b)] + [str(t) for t in thresholds],
                         capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    want = orc.prove(log_n, log_b, want_vectors=False)
    assert f"threads ok: world {world}, {len(want.proof)} proof bytes on every rank equal zk_prove" in out.stdout
    assert "board 1" in out.stdout


@pytest.mark.parametrize("world,log_n,log_b,thresholds", [
    (8, 14, 3, (1, 5, 8)),        # chunked exchanges on three streams, nothing waits on the host
    (4, 20, 3, (0, 0, 0)),        # production thresholds at 2^21 elements per rank
])
def test_shard_ranks_as_threads_with_a_stream_ordered_transport(threads_check, orc, world, log_n, log_b, thresholds):
    """The same harness with a transport that behaves like RCCL (ZK_HARNESS_ASYNC=1): no device synchronisation, peers' pieces
    ordered by HIP events only.  The host-synchronous transports of the other multi-rank tests would hide a missing stream
    dependency inside the library (a chunk hashed before it arrived, a receive buffer reused too early); this one would not."""
    import subprocess
    env = dict(os.environ, ZK_HARNESS_ASYNC="1")
    out = subprocess.run([threads_check, str(world), str(log_n), str(log_b)] + [str(t) for t in thresholds] + ["2"],
                         capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    want = orc.prove(log_n, log_b, want_vectors=False)
    assert f"threads ok: world {world}, {len(want.proof)} proof bytes on every rank equal zk_prove" in out.stdout
    assert "[stream-ordered transport]" in out.stdout


def test_shard_eight_ranks_at_the_benchmark_size_2e27(threads_check, zk):
    """The exact configuration `bench.py --gpus 8` proves: domain 2^27, eight ranks of 2^24 elements each, production
    thresholds (0 = defaults) -- here as eight threads on one GPU.  Every rank's bytes equal the single-GPU prover's at
    2^27 (itself oracle-pinned up to 2^24 and by the strict verifier here); 8 distributed layers (down to 2^20 values: round 5); f (pieces of 2^21 words) in chunks, cp from the received block of f."""
    import subprocess
    import torch
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * 10**9:
        pytest.skip("needs 80 GB of free device memory")
    out = subprocess.run([threads_check, "8", "24", "3", "0", "0", "0"], capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stdout + out.stderr
    from zkstark_amd import _lib
    plen = _lib.load().zk_proof_data_len(24, 3)
    assert f"threads ok: world 8, {plen} proof bytes on every rank equal zk_prove; sharded layers 8, chunked 1, board 1" in out.stdout
    # 8 N-ish bytes in total: f and FRI layers 1..7 (cp = layer 0 comes from the received block of f), 4 bytes per element,
    # (G-1)/G of it to peers
    N = 1 << 27
    words = N + sum(N >> rho for rho in range(1, 8))
    assert f"all-to-all bytes per rank {4.0 * words / 8 * 7 / 8:.0f}" in out.stdout
