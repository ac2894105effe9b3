"""CPU check of the lemma behind fvb::replay_safe (flash_viterbi_amd/csrc/fv_beam_kernels.hip.inc).

The GPU resolve code decides a run of undecided FLASH-BS steps only back to a step whose exact heap replay provably does
not depend on its "doubtful" columns (scores that are upper bounds of their exact values).  Claim: if every doubtful
column c lies outside the initial build (c >= B) and at least B NON-doubtful scores of states < c are >= its bound, then
generate_state_heap (reference src/FLASH_BS_Viterbi_multithread.c:167-211, restated here after oracle/flashvit_oracle.c's
heap_offer / heap_build / heap_replace_min) leaves the SAME heap array for the bounds and for any smaller exact values.
Checked on random score rows with many duplicates; the converse side — a column that fails the test CAN change the
heap — is shown on directed examples so that the criterion is known to be doing something."""
import random

import pytest


def build(h, total):                                   # create_min_heap, FLASH_BS:96-123
    for node in range(total // 2, 0, -1):
        parent, child, temp = node, 2 * node, h[node]
        while child <= total:
            if child + 1 <= total and h[child][0] > h[child + 1][0]:
                child += 1
            if temp[0] <= h[child][0]:
                break
            h[parent] = h[child]
            parent = child
            child *= 2
        h[parent] = temp


def replace_min(h, total, v, state):                   # replace_min_heap_element, FLASH_BS:131-165
    h[1] = (v, state)
    parent, child = 1, 2
    while child <= total:
        if child + 1 <= total and h[child][0] > h[child + 1][0]:
            child += 1
        if h[parent][0] <= h[child][0]:
            break
        h[parent], h[child] = h[child], h[parent]
        parent = child
        child *= 2


def state_heap(scores, beam):                          # generate_state_heap, FLASH_BS:167-211
    h = [None] * (beam + 1)
    for i, v in enumerate(scores):
        if i < beam:
            h[i + 1] = (v, i)
            if i == beam - 1:
                build(h, beam)
        elif v > h[1][0]:
            replace_min(h, beam, v, i)
    return h[1:]


def safe(scores, beam, doubtful):
    """The test of fvb::replay_safe."""
    dset = set(doubtful)
    for c in doubtful:
        if c < beam:
            return False
        if sum(1 for i in range(c) if i not in dset and scores[i] >= scores[c]) < beam:
            return False
    return True


@pytest.mark.parametrize("seed", range(40))
def test_heap_ignores_doubtful_columns_that_pass_the_test(seed):
    rng = random.Random(seed)
    for _ in range(60):
        K = rng.randint(40, 400)
        beam = rng.randint(2, min(40, K // 2))
        levels = rng.randint(3, 60)                    # few distinct values: duplicates at the cut are the rule
        scores = [float(rng.randint(0, levels)) for _ in range(K)]
        # candidates for "doubtful": any column that passes the test on its own, then the joint test decides
        pool = [c for c in range(beam, K) if safe(scores, beam, [c])]
        if not pool:
            continue
        doubtful = rng.sample(pool, min(len(pool), rng.randint(1, 8)))
        if not safe(scores, beam, doubtful):
            continue
        want = state_heap(scores, beam)
        for _ in range(4):
            lowered = list(scores)
            for c in doubtful:
                lowered[c] = scores[c] - rng.choice([0.0, 0.5, 1.0, 3.0, 1e9])
            assert state_heap(lowered, beam) == want


def test_a_column_that_fails_the_test_can_change_the_heap():
    # accepted with its bound, rejected with its exact value: fewer than B scores in front of it are >= the bound
    scores = [1.0, 2.0, 3.0, 6.0, 2.5, 9.0]
    assert not safe(scores, 3, [3])
    low = list(scores); low[3] = 0.5
    assert state_heap(scores, 3) != state_heap(low, 3)
    # inside the initial build every value is placed, so the layout follows it
    scores = [5.0, 4.0, 3.0, 9.0, 8.0, 7.0]
    assert not safe(scores, 3, [1])
    low = list(scores); low[1] = 8.5
    assert state_heap(scores, 3) != state_heap(low, 3)
    # witnesses must be non-doubtful: a column whose only witnesses are doubtful themselves fails the joint test
    scores = [7.0, 1.0, 1.0, 7.0, 6.0]
    assert safe(scores, 2, [4]) and not safe(scores, 2, [3, 4])
