"""The N > 1 path on CPU: world_size-2 and -4 gloo process groups run the sharded prover
(tests/sharded_mirror.py: the torch.distributed MIRROR of the native zk_shard_* protocol; the native prover itself
needs a GPU for every kernel and therefore has NO CPU test -- its tests are tests/test_gpu_shard_native.py) with the CPU test double as
compute backend; the proof bytes must equal the single-process oracle's.  No GPU."""
import hashlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, log_n, log_b, min_chunk_log, q, min_layer_log=None, use_board=True):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    __import__("mp_util").init_pg("gloo", port, rank, world)
    try:
        import oracle
        from sharded_testlib import OracleBackend
        import sharded_mirror as sharded
        be = OracleBackend()
        sp = sharded.ShardedProver(log_n, log_b, sharded.Comm(), be, min_chunk_log=min_chunk_log, min_layer_log=min_layer_log,
                                   use_board=use_board)
        sp.trace_upload(oracle.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        proof2 = sp.prove()                          # a second proof on the same prover: exchange numbers keep running
        assert proof2.data == proof.data
        calls = dict(be.calls)
        calls["board_exchanges"] = sp.n_exchanges if sp.board is not None else -1
        q.put((rank, proof.data, proof.state, sp.n_sharded, calls, [r.hex() for r in sp.transcript["roots"]]))
        sp.close()
    finally:
        dist.destroy_process_group()


def _run(world, log_n, log_b, min_chunk_log, min_layer_log=None, use_board=True):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, log_b, min_chunk_log, q, min_layer_log, use_board)) for r in range(world)]
    for p in procs:
        p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results
    out = gather_results(q, procs, world, 240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(out)


@pytest.mark.parametrize("world,log_n,log_b,min_chunk_log", [(2, 6, 3, 2), (4, 6, 3, 1), (2, 7, 2, 3), (2, 5, 1, 2), (8, 7, 3, 0)])
def test_sharded_prover_matches_oracle(orc, world, log_n, log_b, min_chunk_log):
    want = orc.prove(log_n, log_b)
    assert want.rc == 0
    res = _run(world, log_n, log_b, min_chunk_log)
    for rank, data, state, n_sharded, calls, roots in res:
        assert n_sharded >= 2, "the test must exercise sharded FRI layers"
        assert roots == [bytes(r).hex() for r in want.roots], f"rank {rank}: roots differ"
        assert data == want.proof and state == want.state, f"rank {rank}: proof differs from the oracle"
        assert calls["lde"] == 2 and calls["compose"] == 2 and calls["fold"] == 2 * log_n          # two proofs
        assert calls["compose_block"] == 2, "cp must be committed from the received block of f, not exchanged"
        assert calls["board_exchanges"] == 2 * (n_sharded + 1), "subtree roots must travel through the shared-memory board"
    assert orc.verify(res[0][1], log_n, log_b, want.public_last) == 0


@pytest.mark.parametrize("world,log_n,log_b", [(8, 4, 3), (4, 4, 2), (2, 2, 3)])
def test_sharded_cp_from_f_at_the_smallest_blocks(orc, world, log_n, log_b):
    """cp over a rank's block from the received block of f, at the edge of what the layout allows: n = 2 G (a block is
    exactly the 2B taps long: every f(g x), f(g^2 x) of its last B positions comes from the gathered halo), pieces of two words."""
    want = orc.prove(log_n, log_b)
    assert want.rc == 0
    for rank, data, state, n_sharded, calls, roots in _run(world, log_n, log_b, 1, 1):
        assert calls["compose_block"] == 2
        assert roots == [bytes(r).hex() for r in want.roots], f"rank {rank}: roots differ"
        assert data == want.proof and state == want.state, f"rank {rank}"


def test_sharded_without_root_board(orc):
    """The collective fallback (ranks on different nodes): all-gather of the subtree roots, same proof."""
    want = orc.prove(6, 3)
    for rank, data, state, n_sharded, calls, roots in _run(2, 6, 3, 2, use_board=False):
        assert calls["board_exchanges"] == -1
        assert data == want.proof and state == want.state


@pytest.mark.parametrize("world,log_n,log_b,min_chunk_log,min_layer_log,want_sharded", [
    (2, 7, 2, 2, 7, 3),        # layers of 2^9, 2^8, 2^7 values sharded, the rest replicated
    (2, 7, 2, 2, 30, 1),       # total-size threshold above the domain: only f and cp are sharded
    (4, 6, 3, 1, 8, 2)])
def test_sharded_layer_size_threshold(orc, world, log_n, log_b, min_chunk_log, min_layer_log, want_sharded):
    """Where the proof switches from sharded to replicated layers is a cost decision only: same bytes."""
    want = orc.prove(log_n, log_b)
    res = _run(world, log_n, log_b, min_chunk_log, min_layer_log)
    for rank, data, state, n_sharded, calls, roots in res:
        assert n_sharded == want_sharded
        assert data == want.proof and state == want.state, f"rank {rank}"


def _chunk_worker(rank, world, port, log_n, log_b, lists, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    __import__("mp_util").init_pg("gloo", port, rank, world)
    try:
        import oracle
        from sharded_testlib import ChunkingOracleBackend
        import sharded_mirror as sharded
        be = ChunkingOracleBackend()
        sp = sharded.ShardedProver(log_n, log_b, sharded.Comm(lists=lists), be, min_chunk_log=3, overlap_min_log=3)
        sp.trace_upload(oracle.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        q.put((rank, proof.data, proof.state, be.calls.get("merkle_chunk", 0), be.calls.get("merkle_finish", 0)))
        sp.close()
    finally:
        dist.destroy_process_group()


# sizes: a layer is exchanged in chunks when a (rank, peer) piece has >= 2^10 words (zk_shard_plan), whatever overlap_min_log says
@pytest.mark.parametrize("world,log_n,log_b,lists", [(2, 9, 3, True), (4, 12, 2, True), (8, 13, 3, True), (2, 9, 3, False), (4, 12, 2, False)])
def test_chunked_exchange_multirank(orc, world, log_n, log_b, lists):
    """The chunked commitment of the N > 1 path with real collectives between CPU ranks: hashing chunk c while
    chunk c+1 is exchanged, either as list exchanges over slices of the layer (what RCCL runs; emulated with
    send/recv pairs on gloo) or packed.  Every rank must produce the oracle's proof."""
    want = orc.prove(log_n, log_b, want_vectors=False)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    procs = [ctx.Process(target=_chunk_worker, args=(r, world, port, log_n, log_b, lists, q)) for r in range(world)]
    for p in procs:
        p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results
    out = sorted(gather_results(q, procs, world, 240))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, data, state, n_chunk, n_finish in out:
        # f goes through the chunked exchange (4 chunk builds + 1 finish); cp is recomputed from the received chunks of f
        assert n_chunk >= 4 and n_finish >= 1, "the test must go through the chunked path"
        assert data == want.proof and state == want.state, f"rank {rank}"


def test_sharded_requires_world_dividing_blowup(zk):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_mirror as sharded

    class FakeComm:
        rank, world = 0, 4
    with pytest.raises(zk.ZkError):
        sharded.ShardedProver(6, 1, FakeComm(), backend=None)


def test_path_nodes_and_host_top(orc):
    """merkle.rs:54-71 index walk and the host top-of-tree agree with the oracle's full tree."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_mirror as sharded
    import numpy as np
    vals = np.arange(16, dtype=np.uint32) * 7 + 1
    nodes = orc.merkle_build(vals)
    for leaf in range(16):
        want = [bytes(h) for h in orc.merkle_trace(nodes, leaf)]
        got = [bytes(nodes[j]) for j in sharded.path_nodes(16, leaf)]
        assert got == want
    # 4 subtrees of 4 leaves: top of their roots is the root of the whole tree
    subroots = [bytes(orc.merkle_build(vals[4 * p:4 * p + 4])[0]) for p in range(4)]
    top = sharded.host_merkle_top(subroots)
    assert top[0] == bytes(nodes[0]) and top[1] == bytes(nodes[1]) and top[3:] == subroots
