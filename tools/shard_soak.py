"""Soak of the native sharded prover: `world` ranks sharing one GPU (host-staged transport), many proofs back to back
with the chunked exchange, the root board and lde_commit interleaved; every proof must equal the first one."""
import os, sys, socket
sys.path.insert(0, '.')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(rank, world, port, log_n, reps, uid, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch, torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import zkstark_amd as zk
    from zkstark_amd import sharded
    torch.cuda.set_device(0)
    tp = sharded.staged_transport()
    with zk.ShardContext(log_n, 3, rank, world, uid, transport=tp, min_layer_log=1, min_chunk_log=6, overlap_min_log=9) as sp:
        sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        first = sp.prove()
        root = sp.lde_commit()
        bad = 0
        for i in range(reps):
            p = sp.prove()
            bad += p.data != first.data or p.state != first.state
            if i % 7 == 0:
                bad += sp.lde_commit() != root
        q.put((rank, bad, first.data[:8].hex(), sp.stats()["chunked_layers"]))
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    log_n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    uid = os.urandom(128)
    procs = [ctx.Process(target=worker, args=(r, world, port, log_n, reps, uid, q)) for r in range(world)]
    [p.start() for p in procs]
    out = sorted(q.get(timeout=1000) for _ in range(world))
    [p.join(timeout=60) for p in procs]
    print(out)
    assert all(b == 0 for _, b, _, _ in out) and all(p.exitcode == 0 for p in procs)
    print(f"shard soak ok: {world} ranks x {reps} proofs at 2^{log_n + 3}")
