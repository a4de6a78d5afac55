#!/usr/bin/env python3
"""bench_multi.py -- the N > 1 side of bench.py: the launcher (`--gpus N` without torchrun), the per-rank supervisor with its
transport ladder and watchdog, the sharded run itself (one proof over N GPUs: weak-scaling headline, strong-scaling leg,
configs[3]) and `--plan-only` (the layout and a written-down estimate, no GPU).  bench.py stays the entry point the driver runs;
nothing here is imported for a plain N = 1 run.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
ENTRY = os.path.join(ROOT, "bench.py")          # children are started through the entry point, with the same arguments


# ---- N > 1 without a launcher: one child per GPU, started before this process touches torch or the GPU ----------
class _Terminated(Exception):
    """SIGTERM / SIGINT reached this process (an outer `timeout`, the launcher stopping the other ranks)."""

    def __init__(self, signum):
        super().__init__(f"signal {signum}")
        self.signum = signum


def _raise_on_signals():
    """SIGTERM and SIGINT raise _Terminated in the main thread, so that `finally` blocks run and children are ended."""
    import signal

    def handler(signum, frame):
        raise _Terminated(signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, handler)


def _end_child(p, grace_s=5.0):
    """Ends exactly the child process `p` (and nothing else): SIGTERM, a grace period, SIGKILL."""
    if p is None or p.poll() is not None:
        return
    p.terminate()
    try:
        p.wait(timeout=grace_s)
    except subprocess.TimeoutExpired:
        p.kill()
        p.wait()


def spawn_ranks(args):
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    _raise_on_signals()                                     # an outer `timeout` ends the ranks too, not only this launcher
    try:
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            # N ranks on one host: torch's CPU ops (the control plane; the host-staged rehearsal transport) must not start one
            # OpenMP thread per logical CPU each -- torchrun sets 1 for the same reason (rehearsal n2: 3.4 s per step without, 0.09 s with)
            env.setdefault("OMP_NUM_THREADS", "4")
            # rank 0 inherits stdout (the one JSON line); the other ranks' stdout goes to stderr
            procs.append(subprocess.Popen([sys.executable, ENTRY] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else sys.stderr))
        rc = 0
        live = set(range(args.gpus))
        while live:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0 and rc == 0:
                    rc = code
                    print(f"[bench] rank {r} exited with {code}: stopping the other ranks", file=sys.stderr, flush=True)
                    for q in live:
                        procs[q].terminate()               # exactly the children started above: each is a supervisor that
                                                           # ends its worker and removes its files on SIGTERM
            time.sleep(0.05)
        return rc
    except _Terminated as e:
        print(f"[bench] launcher: {e}: stopping the ranks", file=sys.stderr, flush=True)
        for p_ in procs:
            if p_.poll() is None:
                p_.terminate()
        for p_ in procs:
            _end_child(p_, 8.0)
        return 128 + e.signum


# ---- N > 1: every rank is a SUPERVISOR that runs the real work in a fresh child process -----------------------------
# The multi-GPU run is the driver's (one shot, 600 s limit), and two of the things that can go wrong in it cannot be
# handled inside a process: ncclCommInitRank that never returns, and a collective that waits for a peer for ever.
# A hung RCCL call cannot be cancelled, so the worker that runs it is killed by its own watchdog (os._exit(7): a plain
# exit, never an exec) and the supervisor -- which never touches torch or the GPU -- starts a FRESH worker on the next
# rung of the transport ladder.  Budget: every rung has its own deadline, the whole ladder prints a line inside ~300 s.
# (transport, plain collectives): RCCL inside the library with the chunked exchange and the root board; RCCL with plain
# collectives; NO RCCL at all -- the library's peer-copy transport (csrc/peer.hpp: IPC handles on a shared page, device-to-device
# pulls; round 6), so that a node where no communicator can be formed still yields a measured, verified line; and torch.distributed's
# own RCCL communicator as an independent way to the same wire
LADDER = (("native", False), ("native", True), ("peer", True), ("torch", True))
RUNG_BUDGET_S = (50.0, 40.0, 40.0, 40.0)   # rendezvous + communicator(s) + self-test + first verified proof, per rung
if os.environ.get("ZK_BENCH_RUNG_BUDGET_S"):                 # rehearsals shorten the deadlines (tests/test_bench_cli.py)
    RUNG_BUDGET_S = tuple(float(x) for x in os.environ["ZK_BENCH_RUNG_BUDGET_S"].split(","))
RUN_BUDGET_S = 150.0                   # the timed proofs of the headline (after the first proof)
LEG_BUDGET_S = 60.0                    # every secondary leg after the headline (parity, lde_commit, strong leg, configs[3]): a SOFT
                                       # deadline -- the line is printed without a leg that hangs
if os.environ.get("ZK_BENCH_LEG_BUDGET_S"):                  # rehearsals shorten it (tests/test_bench_cli.py)
    LEG_BUDGET_S = float(os.environ["ZK_BENCH_LEG_BUDGET_S"])
RENDEZVOUS_BUDGET_S = 150.0            # gloo rendezvous of the workers: no RCCL in it, but the ranks' first `import torch` on a fresh
                                       # box can finish a minute apart, and a rank that gives up early would split the generations
SHARD_TIMEOUT_S = 20.0                 # zk_shard_options.timeout_s: every host-side wait on a peer inside the library


def _rendezvous_tag():
    return f"{os.getppid()}_{os.environ.get('MASTER_PORT', '0')}"


# Worker generations.  Every supervisor starts a fresh worker whenever its own worker exits, so the ranks must AGREE on
# which generation of workers is meeting (one gloo rendezvous file per generation).  Counting deaths locally is not
# enough -- a rank whose worker dies once more than its peers' (a crash inside the rendezvous) would wait in a store the
# others never open -- so the generation lives in one shared file per run: a supervisor that starts a worker takes
# max(shared, its own last + 1) under a lock, and a worker waiting for its peers leaves (exit code 8) as soon as the
# shared number has moved past its own.  Whoever opens a new generation removes what an earlier, killed run may have
# left under the same names.
STALE_GENERATION = 8


def _gen_paths(tag, gen, world):
    return f"/tmp/zkbench_store_{tag}_{gen}", [f"/tmp/zkbench_join_{tag}_{gen}_{r}" for r in range(world)]


def _bump_generation(tag, at_least, world):
    import fcntl
    path = f"/tmp/zkbench_gen_{tag}"
    with open(path + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            with open(path) as f:
                cur = int(f.read().strip())
        except (OSError, ValueError):
            cur = -1
        if at_least > cur:                                   # this supervisor opens a new generation
            store, joins = _gen_paths(tag, at_least, world)
            for stale in [store] + joins:
                try:
                    os.unlink(stale)
                except OSError:
                    pass
            with open(path + ".tmp", "w") as f:
                f.write(str(at_least))
            os.replace(path + ".tmp", path)
            cur = at_least
        return cur


def _current_generation(tag):
    try:
        with open(f"/tmp/zkbench_gen_{tag}") as f:
            return int(f.read().strip())
    except (OSError, ValueError):
        return -1


def join_generation(tag, gen, rank, world, budget_s):
    """Worker side, before the gloo rendezvous: wait until every rank's worker of THIS generation is here.  Returns
    'ok', 'stale' (the run has moved on to a later generation: leave at once) or 'timeout'."""
    _, joins = _gen_paths(tag, gen, world)
    with open(joins[rank], "w") as f:
        f.write(str(os.getpid()))
    t0 = time.time()
    while True:
        if all(os.path.exists(j) for j in joins):
            return "ok"
        if _current_generation(tag) > gen:
            return "stale"
        if time.time() - t0 > budget_s:
            return "timeout"
        time.sleep(0.05)


def supervise():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    tag = _rendezvous_tag()
    status = f"/tmp/zkbench_status_{tag}_{rank}"
    first = os.environ.get("ZK_BENCH_TRANSPORT", "native")
    rung = {"native": 0, "peer": 2, "torch": 3}.get(first, 0)
    if os.environ.get("ZK_BENCH_STAGED") == "1":
        rung = 0                                             # one rung only: the host-staged rehearsal transport
    t_start = time.time()
    attempt = 0                                              # workers of this rank that ran a rung (ladder progress)
    stale_restarts = 0
    gen = 0
    gens_used = set()
    code = 1
    p = None
    _raise_on_signals()                                      # SIGTERM (the launcher, an outer `timeout`) ends the worker too
    try:
        while attempt < len(LADDER) + 1:
            gen = _bump_generation(tag, gen, world)
            gens_used.add(gen)
            store, _ = _gen_paths(tag, gen, world)
            env = dict(os.environ, ZK_BENCH_WORKER="1", ZK_BENCH_RUNG=str(rung), ZK_BENCH_ATTEMPT=str(attempt), ZK_BENCH_STATUS=status,
                       ZK_BENCH_STORE=store, ZK_BENCH_GEN=str(gen), ZK_BENCH_TAG=tag, ZK_BENCH_T0=repr(t_start))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            try:
                os.unlink(status)
            except OSError:
                pass
            p = subprocess.Popen([sys.executable, ENTRY] + sys.argv[1:], env=env)   # stdout inherited: the worker prints the line
            code = p.wait()
            p = None
            if code == 0:
                return 0
            if code == STALE_GENERATION and stale_restarts < 8:
                # the peers had already moved on when this worker arrived: join them, the ladder does not advance
                stale_restarts += 1
                os.environ["ZK_BENCH_TEST_STALE_SEEN"] = str(stale_restarts)      # (only read by the scripted worker of the tests)
                gen = max(gen + 1, _current_generation(tag))
                continue
            try:
                with open(status) as f:
                    last = int(f.read().strip())
            except (OSError, ValueError):
                last = rung - 1                               # died before it could say anything: the same rung again
            # 3: not enough GPUs, 4: a proof that differs (parity): nothing another transport would change
            if code in (3, 4) or os.environ.get("ZK_BENCH_STAGED") == "1" or last + 1 >= len(LADDER):
                return code if code > 0 else 1
            where = f"rung {last} ({LADDER[last][0]}{' + plain' if LADDER[last][1] else ''})" if last >= 0 else "the rendezvous"
            print(f"[bench] rank {rank}: worker exited with {code} on {where} after "
                  f"{time.time() - t_start:.0f} s; starting a fresh worker on rung {last + 1}", file=sys.stderr, flush=True)
            rung = last + 1
            attempt += 1
            gen += 1
        return code if code > 0 else 1
    except _Terminated as e:
        print(f"[bench] rank {rank}: supervisor: {e}: ending the worker", file=sys.stderr, flush=True)
        return 128 + e.signum
    finally:
        _end_child(p)                                        # never leave the process that holds the GPU behind
        mine = [status, status + ".tmp"]
        for g in gens_used:
            store, joins = _gen_paths(tag, g, world)
            mine.append(joins[rank])
            if rank == 0:
                mine.append(store)
        if rank == 0:
            mine += [f"/tmp/zkbench_gen_{tag}", f"/tmp/zkbench_gen_{tag}.lock", f"/tmp/zkbench_gen_{tag}.tmp"]
        for path in mine:
            try:
                os.unlink(path)
            except OSError:
                pass


class Watchdog:
    """Ends the process when an armed deadline passes: the only way out of an RCCL call that never returns.  A HARD deadline
    (the transport ladder: communicators, self-test, first proof) exits with code 7 and the supervisor starts a fresh worker
    on the next rung.  A SOFT deadline guards a secondary leg that runs after the headline has been measured: `on_late` prints
    the line with what has been measured so far (rank 0), then every rank exits with code 0 -- a leg that hangs costs that
    leg, never the measurement."""

    def __init__(self, rank):
        import threading
        self.rank, self.deadline, self.what, self.on_late = rank, None, "", None
        self._lock = threading.Lock()
        threading.Thread(target=self._run, daemon=True).start()

    def arm(self, seconds, what, on_late=None):
        with self._lock:
            self.deadline, self.what, self.on_late = time.time() + seconds, what, on_late

    def disarm(self):
        with self._lock:
            self.deadline = None

    def _run(self):
        while True:
            time.sleep(0.25)
            with self._lock:
                late = self.deadline is not None and time.time() > self.deadline
                what, on_late = self.what, self.on_late
            if late and on_late is not None:
                print(f"[bench] rank {self.rank}: WATCHDOG: secondary leg '{what}' did not finish in time; the line is printed without it",
                      file=sys.stderr, flush=True)
                code = 0
                try:
                    code = on_late(what) or 0             # e.g. 4 when the headline proof had already failed its parity check
                finally:
                    os._exit(code)
            if late:
                print(f"[bench] rank {self.rank}: WATCHDOG: '{what}' did not finish in time; this worker exits (7) and the supervisor "
                      f"starts a fresh one on the next rung", file=sys.stderr, flush=True)
                os._exit(7)


def worker_rendezvous(R):
    """Worker side, first thing: the watchdog, the meeting of this worker generation, the gloo control plane.  The control plane
    carries the unique id, agreement rounds and the max over ranks of the time; the data path is inside the library (RCCL, or the
    peer-copy transport)."""
    import datetime
    dist, rank, world = R.dist, R.rank, R.world
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    R.wd = wd = Watchdog(rank)
    start_rung = int(os.environ.get("ZK_BENCH_RUNG", "0"))
    if os.environ.get("ZK_BENCH_STATUS"):                 # a worker that dies before its first rung is retried on the SAME rung
        with open(os.environ["ZK_BENCH_STATUS"], "w") as f:
            f.write(str(start_rung - 1))
    wd.arm(RENDEZVOUS_BUDGET_S, "rendezvous of the control plane (gloo)")
    store = os.environ.get("ZK_BENCH_STORE")
    if store:
        # every rank's worker of THIS generation is here before the rendezvous file is touched; a worker whose peers have
        # moved on to a later generation leaves at once and its supervisor joins them (supervise())
        how = join_generation(os.environ["ZK_BENCH_TAG"], int(os.environ.get("ZK_BENCH_GEN", "0")), rank, world, RENDEZVOUS_BUDGET_S - 5.0)
        if how == "stale":
            print(f"[bench] rank {rank}: the other ranks are already in a later worker generation; rejoining", file=sys.stderr, flush=True)
            os._exit(STALE_GENERATION)
        if how == "timeout":
            print(f"[bench] rank {rank}: the other ranks' workers did not arrive within {RENDEZVOUS_BUDGET_S:.0f} s", file=sys.stderr, flush=True)
            os._exit(7)
        dist.init_process_group("gloo", init_method=f"file://{store}", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=RENDEZVOUS_BUDGET_S))
    else:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=90))
    dist.barrier()                                        # every rank is here: the rung deadlines start together


# ---- --plan-only: the layout of the sharded run and an estimate written down BEFORE the run (no GPU) -------------------------
# One-GPU proxies behind the estimate (every figure an ESTIMATE until a multi-GPU node has run the line):
#   * per-rank device work with the ranks as threads of one process sharing ONE MI355X, total / G
#     (profiles/r06_shard_threads_timing.txt: weak shape, 2^24 elements per rank; profiles/r06_shard_threads_strong.txt: one
#     2^24 proof over G ranks); the harness's exchanges are device copies, so these hold NO link time;
#   * xGMI: 7 links x ~153 GB/s per GPU, point to point (MI355X_MICROARCH.md): in an all-to-all every pair has its own link,
#     so a rank's exchange of `piece` bytes per peer takes piece / 153 GB/s however many peers there are; ~25 us of latency
#     per collective (RCCL launch + handshake; not measured here);
#   * the replicated tail, the decommitment and the size-n inverse transform do not shrink with G.
PROXY = {
    "source": ["profiles/r06_shard_threads_timing.txt", "profiles/r06_shard_threads_strong.txt", "profiles/r05_shard_min_layer.txt"],
    # summed device work of all ranks / G, ms: a LOWER bound of a rank's critical path (no link time, no peer skew)
    "weak_ms_per_rank": {2: 5.7, 4: 5.7, 8: 5.7}, "single_gpu_ms_2e24": 5.7,
    "strong_ms_per_rank": {2: 3.2, 4: 2.0, 8: 1.3},
    "replicated_tail_ms": {2: 0.79, 4: 0.63, 8: 0.53}, "decommit_ms": 0.045,
    "xgmi_link_GBps": 153.0, "collective_latency_us": 25.0,
}


def plan_only(args):
    """`bench.py [--gpus N] --plan-only`: zk_shard_plan for N (or 2, 4, 8) at the weak shape (domain 2^24 * N) and the strong
    shape (domain 2^24), the bytes every rank puts on its links, and the estimated critical path per rank -- so that the one
    multi-GPU run the driver makes can be read against a prediction written down beforehand."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import zkstark_amd as zk
    log_b = args.log_blowup
    worlds = [args.gpus] if args.gpus > 1 else [2, 4, 8]
    out = {"what": "sharded layout (zk_shard_plan) and ESTIMATED per-rank critical path; no GPU was used", "proxy": PROXY, "runs": []}
    for world in worlds:
        lg = world.bit_length() - 1
        for shape, log_n in (("weak", args.log_n + lg), ("strong", args.log_n)):
            for label, kw in (("rungs 0 (chunked exchange, root board)", {}), ("rungs 1-3 (plain collectives)", {"plain_collectives": True})):
                try:
                    pl = zk.shard_plan(world, log_n, log_b, **kw)
                except zk.ZkError as e:
                    out["runs"].append({"world": world, "shape": shape, "error": str(e)})
                    continue
                N = 1 << (log_n + log_b)
                layers = []
                link_s = 0.0
                n_coll = 0
                for lid in range(pl["sharded_layers"] + 1):
                    if lid == 1 and pl["cp_from_f"]:
                        layers.append({"layer": "cp (FRI layer 0)", "exchange": "none: recomputed from the received block of f + a 2B-word all-gather"})
                        n_coll += 1
                        continue
                    piece = 4 << pl["piece_log"][lid]
                    chunked = bool(pl["chunked_mask"] >> lid & 1)
                    layers.append({"layer": "f" if lid == 0 else f"FRI layer {lid - 1}", "values": N >> max(lid - 1, 0), "piece_bytes_per_peer": piece,
                                   "chunked": chunked, "link_us": piece / (PROXY["xgmi_link_GBps"] * 1e9) * 1e6})
                    link_s += piece / (PROXY["xgmi_link_GBps"] * 1e9)
                    n_coll += (1 << pl["log_chunks"]) if chunked else 1
                    n_coll += 0 if not kw else 1                       # plain: the subtree roots travel by all-gather too
                compute = PROXY["weak_ms_per_rank" if shape == "weak" else "strong_ms_per_rank"].get(world)
                est = None
                if compute is not None and args.log_n == 21 and log_b == 3:
                    exposed = link_s * 1e3 * (0.25 if not kw else 1.0)     # chunked: only the first quarter of a layer is exposed (if the link keeps up)
                    est = {"compute_ms_per_rank_lower_bound": compute, "link_ms_if_fully_exposed": link_s * 1e3,
                           "link_ms_exposed_estimate": exposed, "collective_latency_ms": n_coll * PROXY["collective_latency_us"] * 1e-3,
                           "ms_per_proof_ESTIMATE": compute + exposed + n_coll * PROXY["collective_latency_us"] * 1e-3,
                           "single_gpu_ms": PROXY["single_gpu_ms_2e24"],
                           "amdahl_terms_ms": {"replicated_tail": PROXY["replicated_tail_ms"].get(world), "decommitment": PROXY["decommit_ms"],
                                               "size_n_inverse_transform": 0.027}}
                    if shape == "strong":
                        est["speedup_over_single_gpu_ESTIMATE"] = PROXY["single_gpu_ms_2e24"] / est["ms_per_proof_ESTIMATE"]
                    else:
                        est["weak_efficiency_ESTIMATE"] = PROXY["single_gpu_ms_2e24"] / est["ms_per_proof_ESTIMATE"]
                out["runs"].append({"world": world, "shape": shape, "transport": label, "domain_log2": log_n + log_b,
                                    "plan": {k: pl[k] for k in ("sharded_layers", "tail_rounds", "chunked_layers", "chunked_mask", "log_chunks", "min_layer_log",
                                                                "min_chunk_log", "overlap_min_log", "cp_from_f", "all_to_all_bytes", "lde_commit_bytes")},
                                    "bytes_per_element_on_the_links": pl["all_to_all_bytes"] * world / N, "collectives_per_proof": n_coll,
                                    "layers": layers, "estimate": est})
    print(json.dumps(out))
    return 0


def run_sharded(R):
    """The sharded run of one worker process (rank R.rank of R.world): transport ladder, headline, secondary legs, the line.
    R: the namespace bench.main() builds (args, rank, world, the loaded modules, barrier(), dev_stats(), emit_line())."""
    args, rank, local_rank, world = R.args, R.rank, R.local_rank, R.world
    zk, _lib, lib, torch, dist = R.zk, R._lib, R.lib, R.torch, R.dist
    log_b, staged, force_sharded, wd = R.log_b, R.staged, R.force_sharded, R.wd
    barrier, dev_stats, emit_line = R.barrier, R.dev_stats, R.emit_line
    lg = world.bit_length() - 1
    # weak scaling: per-GPU work equals the single-GPU workload; strong: the single-GPU domain over all GPUs
    log_n = args.log_n + (lg if args.scaling == "weak" else 0)
    from zkstark_amd import sharded
    trace = zk.trace_fibsq((1 << log_n) - 1)

    def all_ok(ok):                                       # do all ranks agree that a step worked? (gloo, host)
        t_ = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.MIN)
        return bool(t_.item())

    def shared_from_rank0(make):
        """One object made on rank 0 and handed to every rank; a failure on rank 0 reaches every rank as the SAME ZkError
        (round 3 let rank 0 raise before the broadcast, leaving the others blocked in it)."""
        box = [None]
        if rank == 0:
            try:
                box[0] = ("ok", make())
            except zk.ZkError as e:
                box[0] = ("err", e.code, str(e))
        dist.broadcast_object_list(box, src=0)
        if box[0][0] == "err":
            raise zk.ZkError(box[0][1], f"rank 0: {box[0][2]}")
        return box[0][1]

    def shard_ctx(kind, plain, log_n_, transport_):
        if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "id":      # rehearsal: rank 0 cannot load RCCL
            uid_ = shared_from_rank0(lambda: (_ for _ in ()).throw(zk.ZkError(-4, "simulated: RCCL is not available (ZK_BENCH_SIMULATE_NATIVE_FAILURE=id)")))
        elif kind == "native":
            uid_ = shared_from_rank0(zk.shard_unique_id)
        else:
            uid_ = shared_from_rank0(lambda: os.urandom(128))    # names the shared-memory pages only (root board; peer-copy transport)
        return zk.ShardContext(log_n_, log_b, rank, world, uid_, device=local_rank, transport=transport_, force_collectives=force_sharded,
                               plain_collectives=plain, timeout_s=SHARD_TIMEOUT_S, peer_copy=(kind == "peer"))

    def make_prover(kind, plain):
        """kind: 'native' = RCCL loaded by the library (ncclCommInitRank inside zk_shard_create); 'peer' = the library's peer-copy
        transport (no RCCL: IPC handles + device-to-device copies, host-synchronous); 'torch' = the same
        collectives through torch.distributed's own RCCL communicator; 'staged' = host-staged (rehearsal on one GPU).
        zk_shard_create ends with the known-pattern self-test of the transport (all-to-all on every stream in use, all-gather)."""
        tp_ = None
        if kind == "staged":
            tp_ = sharded.staged_transport()
        elif kind == "torch":
            tp_ = sharded.device_transport(dist.new_group(backend="nccl"))
        sp_ = shard_ctx(kind, plain, log_n, tp_)
        sp_.trace_upload(trace)
        if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "1":   # rehearsal of the fallback
            sp_.inject_failure()
            sp_.close()
            raise zk.ZkError(-2, "simulated failure of the native transport (ZK_BENCH_SIMULATE_NATIVE_FAILURE)")
        if kind == "native" and os.environ.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "hang":  # rehearsal of the watchdog
            time.sleep(3600)
        return sp_, tp_, sp_.prove()                       # the first proof is part of "does this transport work"

    # The first proof must come into being AND be a valid proof on every rank (strict verifier: transcript replay +
    # every opening).  If not, the line says what failed and the run goes down a fixed ladder -- never silently:
    #   native RCCL, chunked exchange + root board  ->  native RCCL, plain collectives (one all-to-all per layer on the
    #   main stream, subtree roots by all-gather)  ->  plain peer copies through IPC handles, no RCCL (csrc/peer.hpp)  ->  the
    #   plain collectives through torch.distributed's own RCCL communicator (sharded.device_transport).
    # An ERROR moves to the next rung inside this process; a HANG is ended by the watchdog and the supervisor starts a
    # fresh worker on the next rung (every rung has its own deadline, RUNG_BUDGET_S).
    # One multi-GPU run is all this code gets (the driver's); everything a one-GPU box can rehearse of it is rehearsed.
    def proof_valid(p_):
        try:
            p_.verify(strict=True)
            return True
        except zk.ZkError as e:
            print(f"[bench] rank {rank}: the first proof does not verify: {e}", file=sys.stderr, flush=True)
            return False

    status_path = os.environ.get("ZK_BENCH_STATUS")

    def note_rung(i):
        if status_path:
            with open(status_path + ".tmp", "w") as f:
                f.write(str(i))
            os.replace(status_path + ".tmp", status_path)

    if staged:
        ladder = [("staged", bool(args.plain_collectives))]
        first_rung = 0
    else:
        ladder = list(LADDER)
        if args.plain_collectives:
            ladder = [(k, True) for k, _ in ladder]
        # every rank starts where the furthest rank starts (a supervisor that saw its worker die later than the others)
        t_ = torch.tensor([int(os.environ.get("ZK_BENCH_RUNG", "0"))], dtype=torch.int32)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)
        first_rung = int(t_.item())
    attempt_no = int(os.environ.get("ZK_BENCH_ATTEMPT", "0"))
    notes = [f"fresh worker #{attempt_no}: an earlier worker hung or died on a rung before {first_rung}"] if attempt_no else []
    sp = transport = proof = None
    kind, plain = ladder[min(first_rung, len(ladder) - 1)]
    rung_used = None
    for ri in range(first_rung, len(ladder)):
        kind, plain = ladder[ri]
        note_rung(ri)
        wd.arm(RUNG_BUDGET_S[min(ri, len(RUNG_BUDGET_S) - 1)], f"rung {ri}: {kind}{' + plain collectives' if plain else ''} "
               "(communicators, self-test, first verified proof)")
        err = None
        try:
            sp, transport, proof = make_prover(kind, plain)
            if not proof_valid(proof):
                err = "the first proof does not verify"
        except zk.ZkError as e:
            err = str(e)
            print(f"[bench] rank {rank}: {kind} transport{' (plain collectives)' if plain else ''} failed: {err}", file=sys.stderr, flush=True)
        if all_ok(err is None):
            rung_used = ri
            break
        notes.append(f"{kind}{' + plain collectives' if plain else ''} failed ({err or 'on another rank'})")
        if sp is not None:
            sp.inject_failure()                           # abort, do not destroy, a communicator that may be half-formed
            sp.close()
        sp = transport = proof = None
    if sp is None:
        sys.exit(5)
    args.plain_collectives = plain
    wd.arm(RUN_BUDGET_S, "the timed proofs of the headline")
    transport_note = ("FALLBACK: " + "; ".join(notes) + f"; running on {kind}{' + plain collectives' if plain else ''}") if notes else None

    def measure(sp_, log_n_, steps_, warm_):
        """`steps_` timed proofs on prover `sp_` (max over the ranks), then two untimed ones: every kernel class bracketed
        with HIP events, and every exchange (zk_shard_set_profiling) -- how long the collectives take on their streams, how
        much of that the hashing streams wait for, and the replicated tail, per rank, so that a bad scaling figure from the
        one multi-GPU run can be read: links, overlap or tail."""
        p_ = None
        for _ in range(warm_):
            p_ = sp_.prove()
        _lib.check(lib.zk_dev_set_profiling(1 << _lib.KERNEL_CLASSES.index("merkle_leaf")))   # dominant kernel only
        dev_stats()
        barrier()
        t0_ = time.perf_counter()
        for _ in range(steps_):
            p_ = sp_.prove()
        dt_local_ = time.perf_counter() - t0_             # this rank's own time (before the closing barrier)
        barrier()
        t_ = torch.tensor([time.perf_counter() - t0_], dtype=torch.float64)
        dist.all_reduce(t_, op=dist.ReduceOp.MAX)         # the slowest rank's time
        dom_ = dev_stats()["merkle_leaf"]
        st_ = sp_.stats()
        _lib.check(lib.zk_dev_set_profiling((1 << len(_lib.KERNEL_CLASSES)) - 1))
        sp_.prove()
        per_kernel_ = dev_stats()
        _lib.check(lib.zk_dev_set_profiling(0))
        sp_.set_profiling(True)
        sp_.prove()
        stx = sp_.stats()
        sp_.set_profiling(False)
        mine = {"rank": rank, "ms_per_step_local": dt_local_ / steps_ * 1e3, "sent_bytes": stx["sent_bytes"], "all_to_all_bytes": stx["all_to_all_bytes"],
                "exchange_ms": stx["exchange_ms"], "exposed_exchange_ms": stx["exposed_exchange_ms"], "tail_ms": stx["tail_ms"],
                "decommit_ms": stx["decommit_ms"],
                "exchanges": stx["exchanges"], "chunked_layers": stx["chunked_layers"], "rccl_nranks": stx["rccl_nranks"],
                "communicators": stx["communicators"], "selftest_ok": stx["selftest_ok"], "selftest_ms": stx["selftest_ms"]}
        per_rank_ = [None] * world
        dist.all_gather_object(per_rank_, mine)
        plan_ = zk.shard_plan(world, log_n_, log_b, force_collectives=force_sharded, plain_collectives=plain)
        gathered_ = [None] * world
        dist.all_gather_object(gathered_, p_.data[:64] + p_.state)
        return {"proof": p_, "dt": float(t_.item()), "steps": steps_, "dom": dom_, "per_kernel": per_kernel_, "st": st_, "per_rank": per_rank_,
                "plan": {k: plan_[k] for k in ("sharded_layers", "tail_rounds", "chunked_layers", "chunked_mask", "log_chunks", "overlap_min_log",
                                               "min_layer_log", "min_chunk_log", "piece_log", "all_to_all_bytes", "cp_from_f")},
                "chunked_mask": plan_["chunked_mask"], "ranks_agree": all(g == gathered_[0] for g in gathered_)}

    def shard_record(m_, n_elems):
        st_, per_rank_ = m_["st"], m_["per_rank"]
        return {**st_, "sent_bytes_per_proof_per_rank": st_["sent_bytes"], "ranks_agree": m_["ranks_agree"],
                "exchanged_bytes_per_element": st_["all_to_all_bytes"] * world / n_elems if world > 1 else 0.0,
                # from the profiled proof (max over ranks; per_rank has every rank's own figures)
                "exchange_ms": max(r["exchange_ms"] for r in per_rank_), "exposed_exchange_ms": max(r["exposed_exchange_ms"] for r in per_rank_),
                "tail_ms": max(r["tail_ms"] for r in per_rank_), "decommit_ms": max(r["decommit_ms"] for r in per_rank_),
                "selftest_ok": all(r["selftest_ok"] for r in per_rank_),
                "per_rank": per_rank_, "timeout_s": SHARD_TIMEOUT_S, "plan": m_["plan"]}

    def single_gpu_parity(log_n_, trace_, proof_, time_it=0):
        """Rank 0: the same trace on the single-GPU prover (itself pinned on the CPU oracle by the tests and by the N = 1
        line): every byte and the final channel state must be equal.  time_it > 0: also that prover's ms per proof."""
        if rank != 0:
            return None
        try:
            with zk.Context(log_n_, log_b, device=local_rank) as c1:
                one = c1.prove(trace_)
                rec = {"against": f"single-GPU prover at domain 2^{log_n_ + log_b} (oracle-pinned)", "equal": one.data == proof_.data and one.state == proof_.state}
                if time_it:
                    t0_ = time.perf_counter()
                    for _ in range(time_it):
                        c1.prove()
                    rec["single_gpu_ms"] = (time.perf_counter() - t0_) / time_it * 1e3
            return rec
        except zk.ZkError as e:
            return {"against": "single-GPU prover", "equal": None, "skipped": str(e)}

    m = measure(sp, log_n, args.steps, max(args.warmup - 1, 0))
    proof = m["proof"]
    N = 1 << (log_n + log_b)
    st = m["st"]
    result = {"dt": m["dt"], "dom": m["dom"], "per_kernel": m["per_kernel"], "setup_ms": st["setup_ms"], "device_bytes": int(st["device_bytes"]),
              "proof_bytes": len(proof.data), "scaling": args.scaling, "units": N * args.steps,
              "parallelism": {"native": f"one proof sharded over {world} GPUs (cyclic domain; native RCCL all-to-all per commitment)",
                              "peer": f"one proof sharded over {world} GPUs (cyclic domain; all-to-all per commitment as peer copies through IPC handles, no RCCL)",
                              "torch": f"one proof sharded over {world} GPUs (cyclic domain; RCCL all-to-all per commitment through torch.distributed)",
                              "staged": f"REHEARSAL: {world} ranks on one GPU, host-staged collectives"}[kind],
              "transport": kind, "transport_note": transport_note,
              "shard": shard_record(m, N),
              "ladder": {"rung": rung_used, "transport": kind, "plain_collectives": plain, "worker": attempt_no, "notes": notes,
                         "seconds_since_supervisor_start": (time.time() - float(os.environ["ZK_BENCH_T0"])) if os.environ.get("ZK_BENCH_T0") else None,
                         "rung_budget_s": list(RUNG_BUDGET_S), "run_budget_s": RUN_BUDGET_S, "leg_budget_s": LEG_BUDGET_S},
              "parity": None, "legs_skipped": [], "proof": proof, "log_n": log_n}
    # From here on the headline exists.  Every further leg runs under a SOFT deadline: if it hangs (a collective that
    # never returns), rank 0 prints the line with what has been measured and every rank exits 0.
    def soft(what):
        def late(_what):
            result["legs_skipped"].append(f"{what}: did not finish within {LEG_BUDGET_S:.0f} s (watchdog); this leg and the later ones are missing")
            if rank == 0:
                emit_line(result)
            return result.get("exit_code") or 0
        wd.arm(LEG_BUDGET_S, what, on_late=late)

    # parity: every rank's bytes must equal the single-GPU prover's
    soft("parity of the headline proof against the single-GPU prover")
    if rank == 0:
        proof.verify(strict=True)
    result["parity"] = single_gpu_parity(log_n, trace, proof)
    barrier()
    def time_lde_commit(ctx_, reps=10):
        root0 = ctx_.lde_commit()
        barrier()
        t0_ = time.perf_counter()
        for _ in range(reps):
            root1 = ctx_.lde_commit()
        barrier()
        dtl_ = torch.tensor([(time.perf_counter() - t0_) / reps], dtype=torch.float64)
        dist.all_reduce(dtl_, op=dist.ReduceOp.MAX)
        return float(dtl_.item()), root0 == root1, ctx_.stats()["all_to_all_bytes"], root1
    if not args.no_secondary:                            # BASELINE.json configs[3] shape at the prover's own domain
        soft("lde_commit_sharded")
        try:
            dtl, stable, a2a, root_c = time_lde_commit(sp)
        except zk.ZkError as e:                           # recorded; the prover is spent, the later legs make their own
            dtl = None
            result["lde_commit_sharded"] = {"error": str(e)}
    if not args.no_secondary and dtl is not None:
        lde_commit = {"workload": f"configs[3] shape: sharded LDE + all-to-all transpose + Merkle commit, domain 2^{log_n + log_b} over {world} GPUs",
                      "ms": dtl * 1e3, "value": N / dtl, "unit": "field-elements/s", "root_stable": stable,
                      "all_to_all_bytes_per_rank": a2a, "chunked": bool(m["chunked_mask"] & 1)}
        if m["chunked_mask"] & 1:
            # the same commitment with PLAIN collectives (one all-to-all on the main stream, no overlap with the hashing): the
            # A/B of the chunked exchange on the very links of this run, not inferred from one-GPU rehearsals
            try:
                with shard_ctx(kind, True, log_n, transport) as spp:
                    spp.trace_upload(trace)
                    dtp, stable_p, _, root_p = time_lde_commit(spp)
                lde_commit["plain_ab"] = {"ms": dtp * 1e3, "root_equal": root_p == root_c, "chunked_over_plain": dtl / dtp}
            except zk.ZkError as e:
                lde_commit["plain_ab"] = {"error": str(e)}
        result["lde_commit_sharded"] = lde_commit
    sp.close()
    # STRONG scaling at the metric's own domain (BASELINE: "at domain 2^20 / 2^24; 1/2/4/8-GPU scaling"): the single-GPU
    # workload -- one 2^(log_n + blow-up) proof -- sharded over the N ranks of this run, beside the weak-scaling headline
    strong_leg = (world > 1 or os.environ.get("ZK_BENCH_STRONG_LEG") == "1") and args.scaling == "weak" and not args.no_secondary
    if strong_leg:
        sl = args.log_n
        key = f"strong_2e{sl + log_b}"
        soft(key)
        if os.environ.get("ZK_BENCH_SIMULATE_LEG_HANG") == "strong":     # rehearsal: a collective of this leg never returns
            time.sleep(3600)
        try:
            tr_s = trace if sl == log_n else zk.trace_fibsq((1 << sl) - 1)
            with shard_ctx(kind, plain, sl, transport) as sps:
                sps.trace_upload(tr_s)
                first_s = sps.prove()
                ok_first = True
                try:
                    first_s.verify(strict=True)
                except zk.ZkError:
                    ok_first = False
                ms_ = measure(sps, sl, min(args.steps, 20), 2)
            par_s = single_gpu_parity(sl, tr_s, ms_["proof"], time_it=5)
            Ns = 1 << (sl + log_b)
            rec = {"workload": f"full prover, domain 2^{sl + log_b} (the single-GPU workload) sharded over {world} GPUs: strong scaling",
                   "ms": ms_["dt"] / ms_["steps"] * 1e3, "steps": ms_["steps"], "value": Ns * ms_["steps"] / ms_["dt"], "unit": "field-elements/s",
                   "first_proof_verifies": ok_first, "parity": par_s, "ranks_agree": ms_["ranks_agree"],
                   "shard": shard_record(ms_, Ns)}
            if par_s and par_s.get("single_gpu_ms"):
                rec["single_gpu_ms"] = par_s["single_gpu_ms"]
                rec["speedup_over_single_gpu"] = par_s["single_gpu_ms"] / rec["ms"]
            result[key] = rec
            if par_s and par_s.get("equal") is False:
                result["parity"] = {**(result["parity"] or {}), "equal": False, "strong_leg_differs": True}
        except zk.ZkError as e:
            result[key] = {"error": str(e)}
        barrier()
    if not args.no_secondary and world in (2, 4, 8) and log_b == 3:
        # BASELINE.json configs[3] at EXACTLY its size: domain 2^26 (trace group 2^23) over the N GPUs of this run
        soft("config4_2e26")
        try:                                              # an ERROR in a secondary leg is recorded, never fatal to the line
            with shard_ctx(kind, plain, 23, transport) as sp4:
                sp4.trace_upload(zk.trace_fibsq((1 << 23) - 1))
                dtl, stable, a2a, root4 = time_lde_commit(sp4)
                st4 = sp4.stats()
            golden = None
            try:                                          # tests/golden/config4_2e26.json: the CPU oracle's root (orc.lde + orc.merkle_build)
                with open(os.path.join(ROOT, "tests", "golden", "config4_2e26.json")) as f:
                    golden = json.load(f)["pinned"]["f_eval_root"]
            except (OSError, KeyError, ValueError):
                pass
            result["config4_2e26"] = {"workload": f"configs[3]: domain 2^26 NTT (LDE) sharded over {world} GPUs, all-to-all transpose, Merkle commit",
                                      "ms": dtl * 1e3, "value": (1 << 26) / dtl, "unit": "field-elements/s", "root_stable": stable,
                                      "all_to_all_bytes_per_rank": a2a, "rccl_nranks": st4["rccl_nranks"], "chunked_layers": st4["chunked_layers"],
                                      "root": root4.hex(), "root_matches_golden": (root4.hex() == golden) if golden else None}
        except zk.ZkError as e:
            result["config4_2e26"] = {"error": str(e)}
    soft("closing barrier")
    barrier()
    wd.disarm()                                           # the last step that waits for a peer: from here on the line WILL be printed once
    parity = result["parity"]
    if parity and parity.get("equal") is False:
        print("[bench] sharded proof differs from the single-GPU prover", file=sys.stderr, flush=True)
        sys.exit(4)
    if not result["shard"]["ranks_agree"]:
        print("[bench] the ranks disagree on the proof", file=sys.stderr, flush=True)
        sys.exit(4)
    if rank == 0:
        emit_line(result)
        if result.get("exit_code"):
            sys.exit(result["exit_code"])
    dist.destroy_process_group()
