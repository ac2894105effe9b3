"""CPU: the oracle (oracle/flashvit_oracle.c) against golden vectors produced by
binaries compiled from the reference's own sources (tests/golden/make_golden.py).
Bit-exact: decoded path (int), final score (float32 equality), memory figure."""
import numpy as np
import pytest

import oracle
from conftest import golden_model, golden_runs

PAIRS, IDS = golden_runs(include_big=True)      # cfg2 (K=3965) included: ~10 s of CPU with OpenMP


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_oracle_matches_reference_binary(g, r):
    A, B, Pi, ob = golden_model(g)
    m = oracle.OracleModel(A, B, Pi)
    T = len(ob)
    if r["algo"] == "vanilla":
        path, score, rc = m.vanilla_decode(ob)
        mem = m.K * T * 8        # sizeof(T1)+sizeof(T2), vanilla Viterbi.c:172
    elif r["algo"] == "checkpoint":
        path, score, rc = m.checkpoint_decode(ob, r["step"])
        mem = oracle.checkpoint_memory_bytes(m.K, T, r["step"])     # checkpoint Viterbi.c:250
    elif r["algo"] == "flash":
        path, score, cells, rc = m.full_decode(ob, r["N"])
        mem = oracle.full_memory_bytes(m.K, T, r["N"])
    else:
        path, score, cells, rc = m.beam_decode(ob, r["N"], r["B"])
        mem = oracle.beam_memory_bytes(m.K, T, r["N"], r["B"])
    assert rc >= 0
    assert path.tolist() == r["path"]
    assert score == np.float32(r["score"])
    assert mem == r["memory"]
    if -1 in r["path"]:
        assert rc == 1  # beam miss reported as a warning, path still the reference's


def test_oracle_rejects_reference_ub_sizes():
    g, _ = PAIRS[0]
    A, B, Pi, ob = golden_model(g)
    m = oracle.OracleModel(A, B, Pi)
    # T == 2N with N > 2: single-element segment, reference prints a wrong path
    assert m.full_decode(ob[:8], 4, check=False)[3] == -1
    assert m.beam_decode(ob[:8], 4, 8, check=False)[3] == -1
    # beam wider than K reads uninitialised heap slots in the reference
    assert m.beam_decode(ob, 1, m.K + 1, check=False)[3] == -1


def test_forward_table_consistent_with_decode():
    """fvo_full_forward + a plain backtrack reproduces N=1 decoding's end state and score."""
    g, _ = PAIRS[0]
    A, B, Pi, ob = golden_model(g)
    m = oracle.OracleModel(A, B, Pi)
    row, args = m.full_forward(ob, 0, len(ob) - 1)
    path, score, _, _ = m.full_decode(ob, 1)
    assert int(np.argmax(row)) == path[-1] and row.max() == score
