"""CPU check of the merge rule of the slabbed step kernels (flash_viterbi_amd/csrc/fv_kernels.hip.inc, trellis_step with
row_lo / srows / merge): source rows are swept one slab per launch, in ascending order; every launch leaves the
best (value, lowest row) of ITS rows, and a later slab replaces what the earlier ones left only if it is STRICTLY greater.
That must be the reference's single ascending scan with strict '>' from (-FLT_MAX, -1)
(src/FLASH_Viterbi_multithread.c:167-173) for every slab size — ties across slab borders included."""
import numpy as np
import pytest

FLT_MAX = np.finfo(np.float32).max


def reference_scan(vals):
    score, arg = -FLT_MAX, -1
    for k, v in enumerate(vals):
        if v > score:
            score, arg = v, k
    return score, arg


def slabbed(vals, slab):
    out_v, out_k = None, None
    for lo in range(0, len(vals), slab):
        part = vals[lo:lo + slab]
        r, kb = -np.inf, -1                                    # a launch: best value, lowest row among equals
        for k, v in enumerate(part):
            if v > r:
                r, kb = v, lo + k
        if lo > 0 and out_k >= 0 and not (r > out_v):          # merge: the slabs below stand unless beaten strictly
            r, kb = out_v, out_k
        anyv = r > -FLT_MAX
        out_v, out_k = (r if anyv else -FLT_MAX), (kb if anyv else -1)
    return out_v, out_k


@pytest.mark.parametrize("seed", range(20))
def test_slab_merge_equals_the_ascending_scan(seed):
    rs = np.random.RandomState(seed)
    for _ in range(200):
        n = rs.randint(1, 200)
        vals = rs.randint(-3, 4, n).astype(np.float32)          # few distinct values: ties everywhere
        dead = rs.rand(n) < rs.choice([0.0, 0.3, 1.0])
        vals[dead] = -np.inf if rs.rand() < 0.5 else -FLT_MAX    # unreachable predecessors (log 0 / -FLT_MAX rows)
        want = reference_scan(vals)
        for slab in (1, 2, 3, 7, 16, 64, n, n + 5):
            assert slabbed(vals, slab) == want, (slab, vals.tolist())
