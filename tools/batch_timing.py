import sys
sys.path.insert(0, '.')
import zkstark_amd as zk
lb = 10
with zk.BatchContext(10, 3, lb) as bc:
    bc.gen_fibsq([1] * (1 << lb), [3141592 + p for p in range(1 << lb)])
    for _ in range(3): bc.prove_raw()
