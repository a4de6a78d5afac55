import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import torch
import zkstark_amd as zk
import os, sys; sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); import sharded_mirror as sharded   # the torch.distributed mirror (test infrastructure)
log_n = 21
a = zk.trace_fibsq((1 << log_n) - 1)
be = sharded.HipBackend(0)
sp = sharded.ShardedProver(log_n, 3, sharded.LocalComm(), be)
sp.trace_upload(a)
for _ in range(2): sp.prove()
pr = cProfile.Profile(); pr.enable()
for _ in range(5): sp.prove()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
