"""Throughput mode at domain 2^24: T threads, each with its own batch of 2^lb proofs in lockstep (zk_batch_*), so that the
latency-bound phases of one batch overlap the hashing of another.  Prints ms per proof for every (T, lb)."""
import sys, threading, time
sys.path.insert(0, '.')
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
combos = [(1, 2), (1, 3), (2, 1), (2, 2), (3, 1), (2, 3)]
for T, lb in combos:
    nb = 1 << lb
    bcs = []
    try:
        for t in range(T):
            bc = zk.BatchContext(log_n, 3, lb)
            bc.gen_fibsq([1] * nb, [3141592 + 16 * t + p for p in range(nb)])
            bc.prove_raw()
            bcs.append(bc)
        reps = 4
        def work(bc):
            for _ in range(reps):
                bc.prove_raw()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(bc,)) for bc in bcs]
        [t.start() for t in th]; [t.join() for t in th]
        dt = time.perf_counter() - t0
        print("threads %d x batch %d x 2^%d: %.3f ms per proof (%.2f GB resident)" % (T, nb, log_n + 3, dt * 1e3 / (T * nb * reps), sum(b.device_bytes for b in bcs) / 1e9), flush=True)
    finally:
        for bc in bcs:
            bc.close()
