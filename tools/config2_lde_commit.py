"""BASELINE.json configs[1]: domain 2^20 LDE + Merkle commit on one GPU (time from trace resident to root on host)."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for log_n in (17, 21):
    a = zk.trace_fibsq((1 << log_n) - 1)
    with zk.Context(log_n, 3) as ctx:
        ctx.trace_upload(a)
        for _ in range(3):
            ctx.lde(); ctx.merkle_commit(0)
        ctx.sync()
        t0 = time.perf_counter()
        K = 20
        for _ in range(K):
            ctx.lde(); root = ctx.merkle_commit(0)
        dt = (time.perf_counter() - t0) / K
        N = 1 << (log_n + 3)
        print(f"domain 2^{log_n+3}: LDE + Merkle commit {dt*1e6:.1f} us = {N/dt/1e9:.2f} G field-elements/s; "
              f"algorithmic 73.5N bytes -> {73.5*N/dt/1e12:.2f} TB/s ({73.5*N/dt/8e12*100:.1f} % of 8 TB/s)")
