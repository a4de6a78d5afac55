import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import zkstark_amd as zk
import oracle as orc
lb = 10
a1s = [3141592 + 7 * p for p in range(1 << lb)]
with zk.BatchContext(10, 3, lb) as bc:
    bc.gen_fibsq([1] * len(a1s), a1s)
    proofs = bc.prove()
for p in (0, 1, 511, 512, 1000, 1023):
    want = orc.prove(10, 3, 1, a1s[p], want_vectors=False)
    assert want.rc == 0 and proofs[p].data == want.proof and proofs[p].state == want.state, p
for pr in proofs:
    pr.verify(strict=True)
print("1024 proofs: 6 compared with the oracle, all verified (strict)")
