import sys, os
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
with zk.Context(log_n, 3) as ctx:
    ctx.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    for _ in range(3): ctx.prove()
    os.environ["X"] = "1"
    ctx.prove()
