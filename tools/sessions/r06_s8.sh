#!/bin/bash
# round 6, session 8: why the peer-copy rung did not finish in 40 / 60 s with ranks sharing the GPU at the benchmark's size
O=gpurun_out/r06i; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
cat > /tmp/peer_big.py <<'PY'
import os, sys, time, faulthandler
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch.multiprocessing as mp
def worker(rank, world, log_n, uid, q):
    faulthandler.dump_traceback_later(50, exit=False)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch, zkstark_amd as zk
    torch.cuda.set_device(0)
    t0 = time.time()
    sp = zk.ShardContext(log_n, 3, rank, world, uid, peer_copy=True, timeout_s=30.0)
    t1 = time.time()
    sp.trace_upload(zk.trace_fibsq((1 << log_n) - 1))
    p = sp.prove(); t2 = time.time()
    for _ in range(3): p = sp.prove()
    t3 = time.time()
    q.put((rank, round(t1 - t0, 2), round(t2 - t1, 2), round((t3 - t2) / 3 * 1e3, 2), sp.stats()["setup_ms"], sp.stats()["selftest_ms"]))
    sp.close()
if __name__ == "__main__":
    world, log_n = int(sys.argv[1]), int(sys.argv[2])
    ctx = mp.get_context("spawn"); q = ctx.Queue(); uid = os.urandom(128)
    ps = [ctx.Process(target=worker, args=(r, world, log_n, uid, q)) for r in range(world)]
    [p.start() for p in ps]
    t0 = time.time()
    out = []
    while len(out) < world and time.time() - t0 < 150:
        try: out.append(q.get(timeout=1.0))
        except Exception: pass
    print("world", world, "log_n", log_n, "results (rank, create s, first proof s, ms per proof, setup_ms, selftest_ms):", sorted(out), flush=True)
    for p in ps:
        p.join(timeout=5)
        if p.is_alive(): p.terminate()
PY
for cfg in "2 18" "2 20" "2 22" "4 23"; do timeout -k 10 200 python /tmp/peer_big.py $cfg > $O/peer_$(echo $cfg | tr ' ' '_').txt 2>&1; echo "$cfg rc=$?"; grep -v "amdgpu.ids" $O/peer_$(echo $cfg | tr ' ' '_').txt | tail -12 | cut -c1-300; done
