"""One rank through the native sharded prover with the collectives forced through RCCL (the N > 1 code path on a one-GPU box):
ms per 2^(log_n + 3) proof for several min_layer_log.  What an extra sharded layer costs with a real (stream-ordered) transport,
where the ranks-as-threads harness pays host barriers and device synchronisations instead."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import zkstark_amd as zk
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
trace = zk.trace_fibsq((1 << log_n) - 1)
with zk.Context(log_n, 3) as c:
    want = c.prove(trace)
for ml in (0, 21, 20, 19, 18, 17):
    uid = zk.shard_unique_id()
    with zk.ShardContext(log_n, 3, 0, 1, uid, min_layer_log=ml, force_collectives=True, timeout_s=20.0) as sp:
        sp.trace_upload(trace)
        p = sp.prove()
        assert p.data == want.data and p.state == want.state
        for _ in range(3):
            sp.prove()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            sp.prove()
        dt = (time.perf_counter() - t0) / reps
        sp.set_profiling(True); sp.prove(); st = sp.stats(); sp.set_profiling(False)
        print("min_layer_log %2d: %.3f ms per proof; sharded layers %d, chunked %d, exchanges %d, exchange_ms %.3f (exposed %.3f), tail_ms %.3f, decommit_ms %.3f" % (
            ml or 22, dt * 1e3, st["sharded_layers"], st["chunked_layers"], st["exchanges"], st["exchange_ms"], st["exposed_exchange_ms"], st["tail_ms"], st["decommit_ms"]), flush=True)
