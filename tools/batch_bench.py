"""Batched proving at the reference's own size (configs[0]: trace 1023, domain 8192): time per batch and per proof."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n, log_b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 3)
for lb in (0, 2, 4, 6, 8, 10):
    if log_n + log_b + lb > 28:
        break
    batch = 1 << lb
    with zk.BatchContext(log_n, log_b, lb) as bc:
        bc.gen_fibsq([1] * batch, [3141592 + p for p in range(batch)])
        bc.prove_raw()
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            bc.prove_raw()
        dt = (time.perf_counter() - t0) / reps
    print("batch %5d x 2^%d: %8.3f ms per batch, %8.2f us per proof, %.3e field-elements/s" % (
        batch, log_n + log_b, dt * 1e3, dt * 1e6 / batch, batch * (1 << (log_n + log_b)) / dt), flush=True)
