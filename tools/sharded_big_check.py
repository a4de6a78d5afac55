"""One rank, forced RCCL collectives, at the domain an 8-GPU weak-scaling run uses (2^27): the sharded
orchestration must give the same bytes as the one-call prover."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import zkstark_amd as zk
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import sharded_mirror as sharded   # the torch.distributed mirror (test infrastructure)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
a = zk.trace_fibsq((1 << log_n) - 1)
be = sharded.HipBackend(0)
for ov in ([int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [None]):
    sp = sharded.ShardedProver(log_n, 3, sharded.Comm(force=True), be, **({} if ov is None else {"overlap_min_log": ov}))
    sp.trace_upload(a)
    p = sp.prove()
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        p = sp.prove()
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("sharded(1 rank, RCCL, overlap_min_log=%s) 2^%d: %.2f ms" % (ov, log_n + 3, best * 1e3), flush=True)
    sp.close(); del sp
torch.cuda.empty_cache()
with zk.Context(log_n, 3) as ctx:
    ctx.trace_upload(a)
    q = ctx.prove()
    t0 = time.perf_counter(); q = ctx.prove(); print("one-call prover: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
assert p.data == q.data and p.state == q.state
p.verify(strict=True)
print("identical and verified")
dist.destroy_process_group()
