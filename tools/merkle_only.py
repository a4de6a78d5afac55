"""One Merkle commitment of 2^log_m random leaves (zk.Merkle-free: device-resident layer of a context), repeated,
for PMC passes on the throughput kernels.  ZK_MERKLE_MAX_K selects the levels per subtree launch."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
with zk.Context(log_n, 3) as ctx:
    ctx.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    ctx.lde()
    ctx.merkle_commit(0)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.merkle_commit(0)
    dt = (time.perf_counter() - t0) / reps
print(f"merkle commit of 2^{log_n + 3} leaves: {dt * 1e6:.1f} us")
