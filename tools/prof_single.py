"""For rocprofv3 --kernel-trace: a few one-call proofs at domain 2^(log_n + 3) on one GPU (zk_prove_resident)."""
import sys
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
with zk.Context(log_n, 3) as c:
    c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    for _ in range(8):
        c.prove()
