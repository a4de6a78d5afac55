"""Soak of the native sharded prover over the built-in RCCL transport with the one rank a one-GPU box allows (collectives
forced): many proofs back to back through both communicators (chunked exchange on the exchange stream), with the exchange
profiling toggled, lde_commit and the transport self-test interleaved; every proof must equal the first one."""
import os, sys, time
sys.path.insert(0, '.')
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import zkstark_amd as zk

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
t0 = time.time()
with zk.ShardContext(log_n, 3, 0, 1, zk.shard_unique_id(), force_collectives=True, min_layer_log=1, min_chunk_log=6, overlap_min_log=10,
                     timeout_s=15.0) as sp:
    sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    first = sp.prove()
    root = sp.lde_commit()
    st = sp.stats()
    assert st["communicators"] == 2 and st["chunked_layers"] >= 1 and st["selftest_ok"] == 1, st
    bad = 0
    for i in range(reps):
        if i % 10 == 0:
            sp.set_profiling(i % 20 == 0)
        p = sp.prove()
        bad += p.data != first.data or p.state != first.state
        if i % 7 == 0:
            bad += sp.lde_commit() != root
        if i % 50 == 49:
            sp.self_test()
print(f"rccl soak: {reps} proofs at 2^{log_n + 3}, {bad} differing, {time.time() - t0:.1f} s")
sys.exit(1 if bad else 0)
