"""bench.py's launch contract: `--gpus N` without a launcher starts the N ranks itself (before touching the GPU)
and fails clearly when the node has fewer GPUs; on a GPU box the N > 1 leg is rehearsed with the ranks sharing
the one GPU (ZK_BENCH_STAGED=1: host-staged collectives, never a measurement)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_n_spawns_ranks_and_needs_n_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present: this would run the real benchmark")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode != 0
    assert "needs 2 GPUs" in out.stderr and "rank 1" in out.stderr        # the ranks WERE launched, then failed clearly
    assert out.stdout.strip() == ""                                       # no JSON line for an unmeasured run


def _load_bench():
    """The N > 1 side of the benchmark (launcher, supervisors, generations, ladder): bench_multi.py since round 6."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("zk_bench_multi_module", os.path.join(ROOT, "bench_multi.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_sigterm_ends_supervisors_and_workers():
    """An outer `timeout` (SIGTERM to the launcher) must not orphan the workers that hold the GPU: the launcher forwards the
    signal, every supervisor ends its worker, and the rendezvous files of the run are removed."""
    import glob
    import signal
    import time
    import psutil
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ZK_BENCH_TEST_WORKER_SLEEP"] = "60"
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    try:
        deadline = time.time() + 30
        kids = []
        while time.time() < deadline:                      # launcher -> 2 supervisors -> 2 workers
            kids = psutil.Process(p.pid).children(recursive=True)
            if len(kids) >= 4:
                break
            time.sleep(0.1)
        assert len(kids) >= 4, [k.cmdline() for k in kids]
        tag_files = lambda: [f for f in glob.glob("/tmp/zkbench_*") if f"_{p.pid}_" in f]
        assert tag_files()                                 # generation / status files of this run exist while it runs
        p.send_signal(signal.SIGTERM)
        out, err = p.communicate(timeout=40)
        assert p.returncode == 128 + signal.SIGTERM, (p.returncode, err[-2000:])
        gone, alive = psutil.wait_procs(kids, timeout=15)
        assert not alive, [a.cmdline() for a in alive]
        assert out.strip() == ""
        assert tag_files() == []
    finally:
        if p.poll() is None:
            p.kill()


def test_bench_worker_generations_agree():
    """The ranks' supervisors agree on the worker generation through one shared file: a rank whose worker died once more
    than its peers' does not wait alone in a rendezvous nobody else opens."""
    import threading
    b = _load_bench()
    tag = f"unittest_{os.getpid()}"
    try:
        # rank 0 is at its 3rd worker, rank 1 at its 1st: both land in generation 2
        g0 = b._bump_generation(tag, 2, 2)
        g1 = b._bump_generation(tag, 0, 2)
        assert g0 == g1 == 2 and b._current_generation(tag) == 2
        res = {}
        th = [threading.Thread(target=lambda r=r: res.__setitem__(r, b.join_generation(tag, 2, r, 2, 10.0))) for r in range(2)]
        [t.start() for t in th]; [t.join() for t in th]
        assert res == {0: "ok", 1: "ok"}
        # a worker waiting in generation 2 while the run moves on to 3 leaves at once; one whose peer never comes times out
        stale = {}
        t = threading.Thread(target=lambda: stale.__setitem__("r", b.join_generation(tag, 3, 0, 2, 10.0)))
        assert b._bump_generation(tag, 3, 2) == 3
        t.start()
        assert b._bump_generation(tag, 4, 2) == 4
        t.join()
        assert stale["r"] == "stale"
        assert b.join_generation(tag, 4, 0, 2, 0.3) == "timeout"
        # opening a generation removes what a killed run left under the same names
        store, joins = b._gen_paths(tag, 7, 2)
        for f in [store] + joins:
            open(f, "w").close()
        assert b._bump_generation(tag, 7, 2) == 7 and not any(os.path.exists(f) for f in [store] + joins)
    finally:
        import glob
        for f in glob.glob(f"/tmp/zkbench_*{tag}*"):
            os.unlink(f)


@pytest.mark.gpu
def test_bench_n2_rehearsal_on_one_gpu():
    """The N > 1 leg end to end (rank spawn, unique-id broadcast, native sharded prover, max over ranks, one JSON
    line from rank 0, parity against the single-GPU prover) with two ranks sharing the GPU."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ZK_BENCH_STAGED"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--log-n", "17"], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["scaling"] == "weak" and rec["config"]["log_n"] == 18
    assert rec["parity_checked"] is True and rec["shard"]["ranks_agree"] is True
    assert rec["shard"]["sharded_layers"] >= 1 and rec["lde_commit_sharded"]["root_stable"] is True
    assert rec["config"]["domain"] == 1 << 21 and rec["value"] == pytest.approx((1 << 21) / (rec["ms_per_step"] * 1e-3), rel=1e-6)
    # the strong-scaling leg at the single-GPU domain (here 2^20) rides in the same line, compared with the single-GPU prover
    st = rec["strong_2e20"]
    assert "2^20" in st["workload"] and st["parity"]["equal"] is True and st["ranks_agree"] is True and st["first_proof_verifies"] is True
    assert st["value"] == pytest.approx((1 << 20) / (st["ms"] * 1e-3), rel=1e-6) and st["single_gpu_ms"] > 0
    assert st["shard"]["plan"]["sharded_layers"] >= 1 and len(st["shard"]["per_rank"]) == 2
    assert all(k in st["shard"]["per_rank"][1] for k in ("exchange_ms", "exposed_exchange_ms", "tail_ms", "decommit_ms"))
    assert "legs_skipped" not in rec


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra,want_transport", [
    ({"ZK_BENCH_TRANSPORT": "torch"}, "torch"),                      # torch.distributed's RCCL communicator on device pointers
    ({"ZK_BENCH_TRANSPORT": "peer"}, "peer"),                        # the library's peer-copy transport (no RCCL; round 6)
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "1"}, "peer"),             # the native transport "fails" twice: the line says so and falls back
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "id"}, "peer"),            # rank 0 cannot even draw the unique id: every rank learns it together
    # the native transport HANGS (as ncclCommInitRank can): the worker's watchdog ends it, the supervisor starts a fresh
    # worker on the next rung, twice; the line comes from the third worker, on the rung that needs no RCCL
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "hang", "ZK_BENCH_RUNG_BUDGET_S": "10,10,60,60"}, "peer"),
    ({}, "native"),
])
def test_bench_sharded_transports_one_rank(env_extra, want_transport):
    """The N > 1 code path with the one rank a one-GPU box allows (collectives forced): the built-in RCCL transport, the
    torch.distributed one (sharded.device_transport: the library's device pointers wrapped as tensors, RCCL through
    torch's communicator on the library's stream), and the recorded fallback from the first to the second."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ZK_BENCH_FORCE_SHARDED="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "16", "--no-secondary"],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert rec["transport"] == want_transport and rec["parity_checked"] is True
    assert (rec["transport_note"] is not None) == ("ZK_BENCH_SIMULATE_NATIVE_FAILURE" in env_extra)
    assert len([l for l in out.stdout.splitlines() if l.strip()]) == 1          # exactly one line, whatever the ladder did
    lad = rec["ladder"]
    assert lad["transport"] == want_transport and rec["shard"]["selftest_ok"] is True
    if env_extra.get("ZK_BENCH_SIMULATE_NATIVE_FAILURE") == "hang":
        assert lad["worker"] == 2 and lad["rung"] == 2 and "WATCHDOG" in out.stderr and lad["seconds_since_supervisor_start"] < 120
    else:
        assert lad["worker"] == 0
    for k in ("exchange_ms", "exposed_exchange_ms", "tail_ms", "decommit_ms", "per_rank", "plan"):
        assert k in rec["shard"], k
    assert rec["shard"]["per_rank"][0]["rank"] == 0 and rec["shard"]["per_rank"][0]["exchanges"] > 0
    assert rec["shard"]["native_rccl"] == (1 if want_transport == "native" else 0)
    assert rec["shard"]["peer_copy"] == (1 if want_transport == "peer" else 0)
    if want_transport == "native":
        assert rec["shard"]["rccl_nranks"] == 1


@pytest.mark.gpu
def test_bench_a_secondary_leg_that_hangs_costs_only_that_leg():
    """Legs after the headline run under a soft deadline: when one hangs (here: the strong-scaling leg, simulated), rank 0
    prints the line with everything measured so far, names the missing leg, and every rank exits 0."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ZK_BENCH_FORCE_SHARDED="1", ZK_BENCH_STRONG_LEG="1", ZK_BENCH_SIMULATE_LEG_HANG="strong", ZK_BENCH_LEG_BUDGET_S="6",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "16"],
                         capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["parity_checked"] is True and rec["value"] > 0 and "lde_commit_sharded" in rec      # the legs before it are there
    assert "strong_2e19" not in rec and any("strong_2e19" in x for x in rec["legs_skipped"])
    assert "WATCHDOG: secondary leg" in out.stderr


@pytest.mark.gpu
def test_bench_strong_scaling_and_exact_config4_rehearsal():
    """--scaling strong keeps the single-GPU domain; with N in {2, 4, 8} the line also carries BASELINE configs[3] at exactly
    domain 2^26 (here: two ranks sharing the GPU, host-staged), whose root is the committed golden value."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["ZK_BENCH_STAGED"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--log-n", "17",
                          "--scaling", "strong"], capture_output=True, text=True, timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.strip()][-1])
    assert rec["scaling"] == "strong" and rec["config"]["log_n"] == 17 and rec["config"]["domain"] == 1 << 20
    c4 = rec["config4_2e26"]
    assert "2^26" in c4["workload"] and c4["root_stable"] is True and c4["root_matches_golden"] is True
    assert c4["all_to_all_bytes_per_rank"] == 4.0 * (1 << 26) / 2 / 2


def test_bench_plan_only_needs_no_gpu():
    """`bench.py --plan-only`: zk_shard_plan for N = 2, 4, 8 at the weak and the strong shape, with the estimated per-rank critical
    path, as one JSON document -- the prediction the one multi-GPU run is read against.  No GPU, no torch."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plan-only"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    doc = json.loads(out.stdout)
    runs = {(r["world"], r["shape"], "plain" in r["transport"]): r for r in doc["runs"]}
    assert set(w for w, _, _ in runs) == {2, 4, 8} and len(runs) == 12
    r = runs[(8, "weak", False)]
    assert r["domain_log2"] == 27 and r["plan"]["sharded_layers"] == 8 and r["plan"]["cp_from_f"] == 1
    N = 1 << 27
    words = N + sum(N >> rho for rho in range(1, 8))                  # f and FRI layers 1 .. 7; cp comes from the block of f
    assert r["plan"]["all_to_all_bytes"] == 4.0 * words / 8 * 7 / 8 and abs(r["bytes_per_element_on_the_links"] - 4.0 * words * 7 / 8 / N) < 1e-9
    assert runs[(8, "weak", True)]["plan"]["chunked_layers"] == 0 and runs[(8, "strong", False)]["domain_log2"] == 24
    for r in doc["runs"]:
        assert "ESTIMATE" in " ".join(r["estimate"].keys())             # every timing figure says what it is
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--plan-only"], capture_output=True, text=True, timeout=120, env=env)
    assert {r["world"] for r in json.loads(one.stdout)["runs"]} == {4}


def test_bench_is_three_files():
    """the entry point under 600 lines; the N > 1 machinery and the secondary legs in their own modules."""
    n = lambda f: sum(1 for _ in open(os.path.join(ROOT, f)))
    assert n("bench.py") < 600 and n("bench_multi.py") > 100 and n("bench_legs.py") > 100
    main = open(os.path.join(ROOT, "bench.py")).read()
    assert "def supervise" not in main and "def spawn_ranks" not in main and "import bench_multi" in main and "import bench_legs" in main


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra,want_note", [
    ({"ZK_BENCH_TRANSPORT": "peer"}, False),
    # the whole ladder with two ranks that SHARE the one GPU: RCCL cannot form a communicator of two ranks on one device (the
    # stand-in for "RCCL does not come up on the driver's node"), the run falls through to the rung that needs no RCCL
    ({"ZK_BENCH_RUNG_BUDGET_S": "40,30,60,60"}, True),
])
def test_bench_two_ranks_sharing_the_gpu_on_the_peer_copy_rung(env_extra, want_note):
    """Row e's fall-back below RCCL, end to end: launcher, supervisors, gloo control plane, zk_shard_* over the library's
    peer-copy transport (IPC handles between the two processes, device-to-device pulls), headline + strong leg + parity."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ZK_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "17"],
                         capture_output=True, text=True, timeout=560, env=env)
    assert out.returncode == 0, out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["transport"] == "peer" and rec["parity_checked"] is True and rec["shard"]["ranks_agree"] is True
    assert rec["shard"]["peer_copy"] == 1 and rec["shard"]["native_rccl"] == 0 and rec["shard"]["chunked_layers"] == 0
    assert (rec["transport_note"] is not None) == want_note
    st = rec["strong_2e20"]
    assert st["parity"]["equal"] is True and st["ranks_agree"] is True and st["shard"]["peer_copy"] == 1
    assert rec["lde_commit_sharded"]["root_stable"] is True


@pytest.mark.parametrize("script,want_rc,want_rungs,want_generations", [
    ("0", 0, ["0"], 1),                                   # the first worker prints the line
    ("3", 3, ["0"], 1),                                   # not enough GPUs: nothing another transport would change
    ("4", 4, ["0"], 1),                                   # a proof that differs: never retried
    ("8,8,0", 0, ["0", "0", "0"], 3),                     # the peers had moved on (twice): rejoin them, the ladder does not advance
    ("7,0", 0, ["0", "0"], 2),                            # died before it reached a rung: the same rung again
    ("7@0,7@1,7@2,0", 0, ["0", "1", "2", "3"], 4),        # a hang on every rung in turn: RCCL chunked -> RCCL plain -> peer copy -> torch
    ("7@0,7@1,7@2,7@3", 7, ["0", "1", "2", "3"], 4),      # the last rung fails too: the run ends with that code, no fifth worker
])
def test_bench_supervisor_state_machine(script, want_rc, want_rungs, want_generations):
    """bench_multi.supervise() with scripted workers (no GPU, no torch): which rung each fresh worker starts on, that stale
    generations do not advance the ladder, that codes 3 / 4 end the run, and that the run's files under /tmp are removed whatever
    happened (every branch of the supervisor is exercised here or by a GPU rehearsal)."""
    import glob
    import re
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(ZK_BENCH_FORCE_SHARDED="1", ZK_BENCH_TEST_WORKER_SCRIPT=script)
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=env)
    out, err = p.communicate(timeout=120)
    assert p.returncode == want_rc, (p.returncode, err[-2000:])
    rungs = re.findall(r"\[bench-test\] worker: rung (\d+)", err)
    gens = re.findall(r"generation (\d+):", err)
    assert rungs == want_rungs, err[-2000:]
    assert len(set(gens)) == want_generations and gens == sorted(gens, key=int), gens
    assert out.strip() == ""                                   # scripted workers print no line
    # the run's files are named after the supervisor's PARENT (the launcher, here this test process) and the rendezvous port
    assert [f for f in glob.glob("/tmp/zkbench_*") if f"_{os.getpid()}_0" in f] == []
