"""ZK_HOST_TIMING laps of zk_batch_prove for the small batches VERDICT r01 flagged (1 x 2^13, 4 x 2^17)."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for log_n, lb in ((10, 0), (10, 2), (14, 0), (14, 2), (14, 4)):
    batch = 1 << lb
    with zk.BatchContext(log_n, 3, lb) as bc:
        bc.gen_fibsq([1] * batch, [3141592 + p for p in range(batch)])
        for _ in range(3):
            bc.prove_raw()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); bc.prove_raw(); ts.append((time.perf_counter() - t0) * 1e6)
        print(f"== batch {batch} x 2^{log_n + 3}: " + " ".join(f"{t:.0f}" for t in ts) + " us", file=sys.stderr, flush=True)
    if log_n + 3 <= 17 and lb == 0:
        with zk.Context(log_n, 3) as ctx:
            ctx.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
            for _ in range(3):
                ctx.prove()
            t0 = time.perf_counter()
            for _ in range(8):
                ctx.prove()
            print(f"== zk_prove 2^{log_n + 3}: {(time.perf_counter() - t0) / 8 * 1e6:.0f} us", file=sys.stderr, flush=True)
