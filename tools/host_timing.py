import sys, os
sys.path.insert(0, '.')
import zkstark_amd as zk
with zk.Context(21, 3) as ctx:
    ctx.trace_upload(zk.trace_fibsq((1 << 21) - 1))
    for _ in range(3): ctx.prove()
    os.environ["X"] = "1"
    ctx.prove()
