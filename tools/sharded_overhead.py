"""Times the sharded orchestration (world = 1, no collectives) against the one-call C++ prover at the
same size: the difference is Python + unfused-kernel overhead of tests/sharded_mirror.py."""
import sys, time
sys.path.insert(0, '.')
import torch
import zkstark_amd as zk
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import sharded_mirror as sharded   # the torch.distributed mirror (test infrastructure)
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
a = zk.trace_fibsq((1 << log_n) - 1)
be = sharded.HipBackend(0)
sp = sharded.ShardedProver(log_n, 3, sharded.LocalComm(), be)
sp.trace_upload(a)
for _ in range(2): p = sp.prove()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): p = sp.prove()
torch.cuda.synchronize(); t1 = time.perf_counter()
print("sharded world=1: %.2f ms/proof" % ((t1 - t0) / 5 * 1e3))
with zk.Context(log_n, 3) as ctx:
    ctx.trace_upload(a)
    for _ in range(2): q = ctx.prove()
    t0 = time.perf_counter()
    for _ in range(5): q = ctx.prove()
    t1 = time.perf_counter()
    print("zk_prove_resident: %.2f ms/proof" % ((t1 - t0) / 5 * 1e3))
assert p.data == q.data
