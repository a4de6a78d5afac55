"""Time per proof for several divisions of the latency-bound end between device and host (zk_ctx_set_host_levels)."""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for log_n in (21, 10, 17):
    a = zk.trace_fibsq((1 << log_n) - 1)
    ref = None
    for levels in ((8, 9), (9, 9), (10, 9), (10, 10), (9, 8), (10, 8), (0, 0)):
        with zk.Context(log_n, 3, host_levels=levels) as ctx:
            ctx.trace_upload(a)
            for _ in range(5):
                p = ctx.prove()
            reps = 30 if log_n >= 20 else 100
            t0 = time.perf_counter()
            for _ in range(reps):
                p = ctx.prove()
            dt = (time.perf_counter() - t0) / reps
            # stand-alone commit (configs[1] shape) with the same hand-over
            ctx.lde(); ctx.merkle_commit(0)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.lde(); ctx.merkle_commit(0)
            dc = (time.perf_counter() - t0) / reps
        ref = ref or p.data
        assert p.data == ref
        print(f"domain 2^{log_n + 3} host_levels {levels}: {dt * 1e6:9.1f} us per proof, LDE + commit {dc * 1e6:8.1f} us", flush=True)
