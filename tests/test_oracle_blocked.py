"""CPU: the oracle's row-streaming beam step (beam_block, used by every beam decode) against the
cell-at-a-time restatement of FLASH_BS_Viterbi_multithread.c:437-446 (beam_cell), cell by cell: scores
float32-equal and winning SLOT equal, including exact ties between slots, -inf transitions, unreachable
destinations (-FLT_MAX, -1) and block tails (K not a multiple of the block)."""
import numpy as np
import pytest

import modelgen
import oracle


@pytest.mark.parametrize("kind,K,M,beam,seed", [("data_script", 300, 7, 40, 3), ("data_script", 1000, 5, 1000, 4),
                                                ("ties_semi", 777, 4, 97, 5), ("ties_all", 513, 4, 256, 6),
                                                ("data_script", 257, 3, 2, 7)])
def test_blocked_beam_step_equals_cell_form(kind, K, M, beam, seed):
    spec = dict(kind=kind, K=K, M=M, T=4, prob=0.05 if kind == "data_script" else 0.5, seed=seed)
    A, B, Pi, _ = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    rs = np.random.RandomState(seed)
    for trial in range(3):
        states = rs.choice(K, size=beam, replace=False).astype(np.int32)
        # few distinct values => many exact ties between slots; one unreachable entry
        vals = (-rs.randint(1, 6, size=beam) * 0.25).astype(np.float32)
        if trial == 1:
            vals[rs.randint(beam)] = np.float32(-np.finfo(np.float32).max)
        for o in range(M):
            s0, a0 = om.beam_step_probe(vals, states, o, blocked=False)
            s1, a1 = om.beam_step_probe(vals, states, o, blocked=True)
            assert (s0 == s1).all() and (a0 == a1).all()
    om.close()
