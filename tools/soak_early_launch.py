"""Soak of zk_ctx_set_early_launch: thousands of proofs with the switch on at three sizes (and four contexts proving at once at the
smallest), every proof's bytes compared with the first one's.  python tools/soak_early_launch.py [seconds_per_leg]"""
import sys, time, threading
sys.path.insert(0, '.')
import zkstark_amd as zk

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
for log_n in (10, 17, 21):
    with zk.Context(log_n, 3) as c:
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        ref = c.prove()
        if not c.set_early_launch(True):
            print("early launch is not supported on this device"); sys.exit(0)
        t0, k, bad = time.time(), 0, 0
        while time.time() - t0 < secs:
            p = c.prove(); k += 1
            bad += p.data != ref.data or p.state != ref.state
        print(f"domain 2^{log_n + 3}: {k} proofs with early launch on, {bad} differing", flush=True)
        assert bad == 0
ctxs = [zk.Context(10, 3) for _ in range(4)]
for c in ctxs:
    c.trace_upload(zk.trace_fibsq(1023)); c.set_early_launch(True)
ref = ctxs[0].prove()
t0, k, bad = time.time(), 0, 0
while time.time() - t0 < secs:
    for p in zk.prove_many(ctxs):
        k += 1; bad += p.data != ref.data
print(f"4 contexts at once, domain 2^13: {k} proofs with early launch on, {bad} differing", flush=True)
assert bad == 0
for c in ctxs:
    c.close()
print("early-launch soak ok")
