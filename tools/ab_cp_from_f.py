"""A/B of the cp commitment of the sharded prover on one GPU: one rank with the collectives forced through RCCL, cp recomputed from
the received block of f (default) against cp exchanged like every other layer (exchange_cp, rounds 1-4).  Per proof: wall time, the
exchange timing of zk_shard_set_profiling, and the device time of every kernel class (zk_dev_set_profiling).  At ONE rank the
exchange of a rank with itself goes through RCCL's transport kernel (~0.14 TB/s), so the wall-time difference overstates what G > 1
ranks gain per exchange; the kernel classes show what the block form of the composition costs inside the leaf hashing."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import zkstark_amd as zk
from zkstark_amd import _lib
lib = _lib.load()
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 21
trace = zk.trace_fibsq((1 << log_n) - 1)
with zk.Context(log_n, 3) as c:
    want = c.prove(trace)


def dev_stats():
    arr = _lib.kernel_stat_array()
    _lib.check(lib.zk_dev_kernel_stats(arr, len(arr), 1))
    return {name: (int(a.launches), a.ms) for name, a in zip(_lib.KERNEL_CLASSES, arr)}


for rep in range(2):
    for exchange_cp in (False, True):
        uid = zk.shard_unique_id()
        with zk.ShardContext(log_n, 3, 0, 1, uid, force_collectives=True, timeout_s=20.0, exchange_cp=exchange_cp) as sp:
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
            _lib.check(lib.zk_dev_set_profiling((1 << len(_lib.KERNEL_CLASSES)) - 1))
            dev_stats()
            sp.prove()
            ks = dev_stats()
            _lib.check(lib.zk_dev_set_profiling(0))
            print("cp %-9s: %.3f ms per proof; sharded %d, chunked %d, exchanges %d, exchange_ms %.3f (exposed %.3f), all-to-all bytes %.0f | %s" % (
                "exchanged" if exchange_cp else "from f", dt * 1e3, st["sharded_layers"], st["chunked_layers"], st["exchanges"], st["exchange_ms"],
                st["exposed_exchange_ms"], st["all_to_all_bytes"], "  ".join("%s %dx %.3f" % (k, v[0], v[1]) for k, v in ks.items() if v[0])), flush=True)
