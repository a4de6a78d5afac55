"""BASELINE.json configs[1] alone (domain 2^20: LDE + Merkle commit, trace resident -> root on host), for
rocprofv3 --kernel-trace --stats and ZK_HOST_TIMING laps.  Prints the mean time of `reps` iterations."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
a = zk.trace_fibsq((1 << log_n) - 1)
with zk.Context(log_n, 3) as ctx:
    ctx.trace_upload(a)
    for _ in range(5):
        ctx.lde(); ctx.merkle_commit(0)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.lde(); root = ctx.merkle_commit(0)
    dt = (time.perf_counter() - t0) / reps
print(f"domain 2^{log_n + 3}: LDE + Merkle commit {dt * 1e6:.1f} us per iteration ({reps} iterations), root {root.hex()[:16]}")
