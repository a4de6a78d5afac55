"""Soak: many proofs through every host/device hand-off (mailbox, last-workgroup copy, scatter, batch pool);
any race shows as a differing proof or a mailbox timeout."""
import sys, time, threading
sys.path.insert(0, '.')
import zkstark_amd as zk

t_end = time.time() + float(sys.argv[1]) if len(sys.argv) > 1 else time.time() + 60
count = {"big": 0, "small": 0, "batch": 0, "mid": 0, "sharded": 0, "field": 0, "checked": 0}
err = []

def loop(name, make, prove):
    try:
        obj = make()
        ref = prove(obj)
        while time.time() < t_end:
            if prove(obj) != ref:
                err.append(name); return
            count[name] += 1
        obj.close()
    except Exception as e:          # noqa: BLE001
        err.append(f"{name}: {e}")

def mk_ctx(log_n, levels=None, hash="sha256", checks=False):
    def make():
        c = zk.Context(log_n, 3, host_levels=levels, hash=hash)
        c.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
        if checks:
            c.set_checks(True)
        return c
    return make

def mk_batch():
    b = zk.BatchContext(10, 3, 8)
    b.gen_fibsq([1] * 256, [7 + p for p in range(256)])
    return b

def mk_sharded():
    import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import sharded_mirror as sharded
    sp = sharded.ShardedProver(17, 3, sharded.LocalComm(), sharded.HipBackend(0), min_chunk_log=8)
    sp.trace_upload(zk.trace_fibsq((1 << 17) - 1))
    return sp

th = [threading.Thread(target=loop, args=("sharded", mk_sharded, lambda sp: sp.prove().data)),
      threading.Thread(target=loop, args=("big", mk_ctx(21), lambda c: c.prove().data)),
      threading.Thread(target=loop, args=("mid", mk_ctx(16, (7, 8)), lambda c: c.prove().data)),
      threading.Thread(target=loop, args=("small", mk_ctx(10), lambda c: c.prove().data)),
      threading.Thread(target=loop, args=("field", mk_ctx(15, hash="field"), lambda c: c.prove().data)),      # row-of-16 hash in the latency levels
      threading.Thread(target=loop, args=("checked", mk_ctx(13, checks=True), lambda c: c.prove().data)),     # zk_ctx_set_checks on every proof
      threading.Thread(target=loop, args=("batch", mk_batch, lambda b: b.prove_raw()[0].tobytes()))]
[t.start() for t in th]
[t.join() for t in th]
print("proofs:", count, "errors:", err)
sys.exit(1 if err else 0)
