"""Per-level cost of the latency phase: whole trees over 2^9 .. 2^18 random leaves built to the root on the device
(zk.Merkle.new -> zk_merkle_build_host: no host hand-over), under rocprofv3 --kernel-trace; tools/merkle_levels_report.py
turns the trace into per-launch durations by (kernel, grid)."""
import sys
sys.path.insert(0, '.')
import numpy as np
import zkstark_amd as zk
rng = np.random.default_rng(1)
for log_m in range(9, 19):
    vals = rng.integers(0, zk.P, size=1 << log_m, dtype=np.uint64).astype(np.uint32)
    for _ in range(4):
        m = zk.Merkle.new(1 << log_m, vals)
print("done")
