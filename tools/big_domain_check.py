import sys, time
sys.path.insert(0, '.')
import zkstark_amd as zk
for log_n in (23, 24):
    t0=time.time()
    a = zk.trace_fibsq((1<<log_n)-1)
    with zk.Context(log_n, 3) as ctx:
        ctx.trace_upload(a)
        t1=time.time()
        p = ctx.prove()
        t2=time.time()
        p = ctx.prove()
        t3=time.time()
        print(log_n, "setup+trace %.2fs first %.1f ms second %.1f ms" % (t1-t0, (t2-t1)*1e3, (t3-t2)*1e3), len(p.data), ctx.device_bytes/1e9, "GB", flush=True)
    p.verify(strict=True)
    print("verified", flush=True)
