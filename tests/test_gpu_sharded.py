"""The torch.distributed MIRROR of the sharded prover (tests/sharded_mirror.py: test infrastructure, NOT the product --
the product is the native zk_shard_* prover, covered by tests/test_gpu_shard_native.py) on real hardware:
HipBackend (C-ABI device primitives on torch tensors).
world = 1 in-process, and world = 2 as two processes sharing the one GPU of the test box with the
collectives staged through gloo (RCCL needs one GPU per rank; the exchange logic is the same)."""
import os
import socket
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("log_n,log_b", [(10, 3), (13, 3)])
def test_sharded_world1_matches_oracle(zk, orc, log_n, log_b):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_mirror as sharded
    want = orc.prove(log_n, log_b, want_vectors=False)
    be = sharded.HipBackend(0)
    sp = sharded.ShardedProver(log_n, log_b, sharded.LocalComm(), be, min_chunk_log=4)
    sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    proof = sp.prove()
    assert proof.data == want.proof and proof.state == want.state
    proof.verify()
    sp.close()


def _worker(rank, world, port, log_n, log_b, q, lat):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    if lat:
        from zkstark_amd import _lib
        _lib.check(_lib.load().zk_dev_set_merkle_latency_log(lat))     # chunk builds hand over at this depth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    __import__("mp_util").init_pg("gloo", port, rank, world)
    try:
        import zkstark_amd as zk
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sharded_mirror as sharded
        torch.cuda.set_device(0)
        be = sharded.HipBackend(0)
        sp = sharded.ShardedProver(log_n, log_b, sharded.Comm(staged=True), be, min_chunk_log=6, overlap_min_log=8)
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        q.put((rank, proof.data, proof.state, sp.n_sharded))
        sp.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,log_n,log_b,lat", [(2, 12, 3, 0), (4, 14, 3, 0), (2, 16, 3, 12), (4, 16, 2, 13)])
def test_sharded_multirank_one_gpu(orc, world, log_n, log_b, lat):
    import torch.multiprocessing as mp
    want = orc.prove(log_n, log_b, want_vectors=False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    procs = [ctx.Process(target=_worker, args=(r, world, port, log_n, log_b, q, lat)) for r in range(world)]
    for p in procs:
        p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results, fresh_store
    out = sorted(gather_results(q, procs, world, 240))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, data, state, n_sharded in out:
        assert n_sharded >= 3
        assert data == want.proof and state == want.state, f"rank {rank}"


@pytest.mark.parametrize("log_n,log_b,rank_exp", [(6, 0, 5), (7, 1, 3), (9, 2, 1), (10, 3, 0), (13, 0, 7), (16, 0, 3)])
def test_shard_domain_primitives_match_definition(zk, orc, log_n, log_b, rank_exp):
    """zk_dev_lde / zk_dev_compose / zk_dev_fri_fold on a shard domain (shift = w * h_global^r, blow-up
    B/G down to 1) against the formulas written out directly in the CPU test double."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from sharded_testlib import OracleBackend
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import sharded_mirror as sharded
    P = 3221225473
    hglob = pow(5, (P - 1) >> (log_n + 3), P)
    shift = 5 * pow(hglob, rank_exp, P) % P
    hb, ob = sharded.HipBackend(0), OracleBackend()
    dh, do = hb.domain(log_n, log_b, shift), ob.domain(log_n, log_b, shift)
    n, N = 1 << log_n, 1 << (log_n + log_b)
    rng = np.random.default_rng(log_n)
    trace = rng.integers(0, P, size=n - 1, dtype=np.uint64).astype(np.uint32)
    trace[0] = 1
    tr0 = np.concatenate([trace, np.zeros(1, dtype=np.uint32)])
    th, to = hb.upload(tr0), ob.upload(tr0)
    fh, fo = hb.empty(N), ob.empty(N)
    hb.lde(dh, th, hb.empty(2 * n), fh)
    ob.lde(do, to, ob.empty(n), fo)
    assert np.array_equal(hb.to_host(fh), ob.to_host(fo))
    alphas = [int(rng.integers(0, 2**32)) for _ in range(3)]
    ch, co = hb.empty(N), ob.empty(N)
    hb.compose(dh, fh, ch, 1, int(trace[-1]), alphas)
    ob.compose(do, fo, co, 1, int(trace[-1]), alphas)
    assert np.array_equal(hb.to_host(ch), ob.to_host(co))
    cur_h, cur_o = ch, co
    for rnd in range(min(log_n, 4)):
        beta = int(rng.integers(0, 2**32))
        m_log = log_n + log_b - rnd
        nh, no = hb.empty(1 << (m_log - 1)), ob.empty(1 << (m_log - 1))
        hb.fold(dh, cur_h, nh, m_log, rnd, beta)
        ob.fold(do, cur_o, no, m_log, rnd, beta)
        assert np.array_equal(hb.to_host(nh), ob.to_host(no)), rnd
        cur_h, cur_o = nh, no
    hb.close()


def _nccl_worker(port, log_n, log_b, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").init_pg("nccl", port, 0, 1, device_id=torch.device("cuda", 0))
    try:
        import zkstark_amd as zk
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sharded_mirror as sharded
        be = sharded.HipBackend(0)
        sp = sharded.ShardedProver(log_n, log_b, sharded.Comm(force=True), be, min_chunk_log=6, overlap_min_log=8)
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        proof = sp.prove()
        dist.barrier()
        q.put((proof.data, proof.state))
        sp.close()
    finally:
        dist.destroy_process_group()


def test_sharded_collectives_over_rccl_single_rank(orc):
    """The exact torch.distributed calls of the N > 1 path (all_to_all_single, all_gather_into_tensor on
    int32 device tensors, barrier) over the nccl backend = RCCL, with the one rank a one-GPU box allows."""
    import torch.multiprocessing as mp
    want = orc.prove(12, 3, want_vectors=False)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    p = ctx.Process(target=_nccl_worker, args=(port, 12, 3, q))
    p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results, fresh_store
    data, state = gather_results(q, [p], 1, 240)[0]
    p.join(timeout=120)
    assert p.exitcode == 0
    assert data == want.proof and state == want.state


def _config4_worker(rank, world, port, log_n, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    __import__("mp_util").init_pg("gloo", port, rank, world)
    try:
        import zkstark_amd as zk
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import sharded_mirror as sharded
        torch.cuda.set_device(0)
        be = sharded.HipBackend(0)
        sp = sharded.ShardedProver(log_n, 3, sharded.Comm(staged=True), be)
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        root = sp.lde_commit()
        # the cyclic shard itself: element j of rank r is f_eval[r + world * j]
        shard_head = be.to_host(sp._layer(0)[:4]).tolist()
        q.put((rank, root, shard_head))
        sp.close()
    finally:
        dist.destroy_process_group()


def test_config4_sharded_lde_and_transpose_domain_2e26(zk, config4_expected):
    """configs[3] through the torch.distributed mirror (tests/sharded_mirror.py): domain 2^26 evaluated by 2 ranks
    (each its cosets, no communication), all-to-all transpose to natural order, subtree commitment; the root and
    the shards equal the CPU oracle's (orc.lde + orc.merkle_build)."""
    import torch.multiprocessing as mp
    log_n, world = config4_expected["log_n"], 2
    want_root, head = config4_expected["root"], config4_expected["head"]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mctx = mp.get_context("spawn")
    q = mctx.Queue()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    __import__("mp_util").fresh_store(port)
    procs = [mctx.Process(target=_config4_worker, args=(r, world, port, log_n, q)) for r in range(world)]
    for p in procs:
        p.start()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from mp_util import gather_results, fresh_store
    out = sorted(gather_results(q, procs, world, 240))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, root, shard_head in out:
        assert root == want_root, f"rank {rank}"
        assert shard_head == [int(head[rank + world * j]) for j in range(4)]
