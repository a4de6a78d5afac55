"""A/B of zk_ctx_set_early_launch inside one process and one build: proofs at domain 2^(log_n + 3) with the switch off / on, alternating,
`reps` timed proofs per leg after a few warm-up ones; every proof's bytes are compared with the first one's.
    python tools/ab_early_launch.py [log_n ...]"""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk

for log_n in [int(a) for a in sys.argv[1:]] or [21, 17, 10]:
    reps = {21: 40, 17: 200, 10: 500}.get(log_n, 100)
    with zk.Context(log_n, 3) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        ref = c.prove()
        out = []
        for leg in range(6):
            on = bool(leg & 1)
            got = c.set_early_launch(on)
            for _ in range(5):
                p = c.prove()
            t0 = time.perf_counter()
            for _ in range(reps):
                p = c.prove()
            dt = (time.perf_counter() - t0) / reps
            assert p.data == ref.data and p.state == ref.state
            out.append((on and got, dt * 1e6))
        off = [u for o, u in out if not o]
        on_ = [u for o, u in out if o]
        print(f"domain 2^{log_n + 3}: early launch off {' / '.join(f'{u:.1f}' for u in off)} us per proof, on {' / '.join(f'{u:.1f}' for u in on_)} us "
              f"(mean {sum(off) / len(off):.1f} -> {sum(on_) / max(len(on_), 1):.1f})", flush=True)
