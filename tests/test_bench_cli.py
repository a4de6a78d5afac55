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


@pytest.mark.gpu
@pytest.mark.parametrize("env_extra,want_transport", [
    ({"ZK_BENCH_TRANSPORT": "torch"}, "torch"),                      # torch.distributed's RCCL communicator on device pointers
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "1"}, "torch"),            # the native transport "fails": the line says so and falls back
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "id"}, "torch"),           # rank 0 cannot even draw the unique id: every rank learns it together
    # the native transport HANGS (as ncclCommInitRank can): the worker's watchdog ends it, the supervisor starts a fresh
    # worker on the next rung, twice; the line comes from the third worker
    ({"ZK_BENCH_SIMULATE_NATIVE_FAILURE": "hang", "ZK_BENCH_RUNG_BUDGET_S": "10,10,60"}, "torch"),
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
    for k in ("exchange_ms", "exposed_exchange_ms", "tail_ms", "per_rank", "plan"):
        assert k in rec["shard"], k
    assert rec["shard"]["per_rank"][0]["rank"] == 0 and rec["shard"]["per_rank"][0]["exchanges"] > 0
    assert rec["shard"]["native_rccl"] == (1 if want_transport == "native" else 0)
    if want_transport == "native":
        assert rec["shard"]["rccl_nranks"] == 1


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
