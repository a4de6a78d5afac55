"""Largest domains one MI355X holds (N <= 2^30: n*B must divide 2^30): prove twice, compare,
verify with the transcript replay.  python tools/big_domain_check.py [log_n ...]   (log_blowup 3)"""
import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for log_n in [int(x) for x in sys.argv[1:]] or [23, 24]:
    t0 = time.time()
    a = zk.trace_fibsq((1 << log_n) - 1)
    with zk.Context(log_n, 3) as ctx:
        ctx.trace_upload(a)
        t1 = time.time()
        p = ctx.prove()
        t2 = time.time()
        q = ctx.prove()
        t3 = time.time()
        print("domain 2^%d: setup+trace %.2f s, first proof %.1f ms, second %.1f ms, %d proof bytes, %.1f GB on the device" % (
            log_n + 3, t1 - t0, (t2 - t1) * 1e3, (t3 - t2) * 1e3, len(p.data), ctx.device_bytes / 1e9), flush=True)
    assert p.data == q.data
    p.verify(strict=True)
    print("identical twice, verified (strict)", flush=True)
