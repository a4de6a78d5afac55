"""For rocprofv3 --kernel-trace --stats: a few one-rank RCCL proofs with cp from the block of f, then with cp exchanged."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
mode = sys.argv[2] if len(sys.argv) > 2 else "new"
trace = zk.trace_fibsq((1 << log_n) - 1)
uid = zk.shard_unique_id()
with zk.ShardContext(log_n, 3, 0, 1, uid, force_collectives=True, timeout_s=20.0, exchange_cp=(mode == "old")) as sp:
    sp.trace_upload(trace)
    for _ in range(8):
        sp.prove()
