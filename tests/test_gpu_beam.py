"""GPU parity of the FLASH-BS (dynamic beam) path through the C-ABI.

Bar: decoded path bit-exact — including the -1 entries the reference prints after a beam miss —
and final score float32-equal, against golden vectors from the reference-built binaries and
against the oracle on fresh seeded inputs."""
import numpy as np
import pytest

import oracle
from conftest import golden_model, golden_runs
from flash_viterbi_amd import decoder

pytestmark = pytest.mark.gpu

PAIRS, IDS = golden_runs(include_big=True, algo="flashbs")
EAGER = 1 << 20      # FV_OPT_DEBUG bit 20: no speculative member lists


@pytest.fixture(scope="module")
def ctxs():
    cache = {}

    def get(g):
        if g["name"] not in cache:
            A, B, Pi, ob = golden_model(g)
            fv = decoder.FlashViterbi(0)
            fv.set_model(A, B, Pi)
            cache[g["name"]] = (fv, ob)
        return cache[g["name"]]
    yield get
    for fv, _ in cache.values():
        fv.close()


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_beam_reference_mode_matches_golden(ctxs, g, r):
    """Default: duplicate scores at the cut are carried speculatively and the heap is replayed only where its outcome is
    observable (lazy replays); FV_OPT_DEBUG bit 20: every duplicate step replays at once (the round-2 path); bit 19: all
    layouts rebuilt, which decides every step."""
    fv, ob = ctxs(g)
    for dbg in (0, EAGER, 524288, EAGER | 524288, NOCUT, NOCUT | 512, OWNPRED):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_beam(ob, r["N"], r["B"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert path.tolist() == r["path"], dbg
        assert score == np.float32(r["score"])
        assert rc == (decoder.WARN_BEAM_MISS if -1 in r["path"] else 0)
        assert dbg != EAGER or fv.stats()["beam_spec_steps"] == 0
    assert decoder.reference_memory_bytes(fv.K, len(ob), r["N"], r["B"]) == r["memory"]


OWNPRED = 1 << 7     # FV_OPT_DEBUG bit 7: the next cut is predicted from the pass's own two last cuts only (not from the earlier generation's)
NOCUT = 1 << 23      # FV_OPT_DEBUG bit 23: runs of undecided steps are always decided in full (no replay_safe shortcut)
MANY = 1 << 22       # FV_OPT_DEBUG bit 22: every selection in the memory-resident form, compaction blocks of two rounds


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_beam_memory_resident_select_matches_golden(ctxs, g, r):
    """K > 65536 is beyond the 64 rounds of keys the register selects hold: every selection of such a model runs in the
    lean kernel — on the candidate list where there is one, else over the K scores in memory, compacted in blocks of
    rounds.  Bit 22 takes that route at any K (blocks of two rounds, so several blocks already at K > 2048); with bit 10
    (no candidate lists) every step selects over all K scores in memory."""
    fv, ob = ctxs(g)
    for dbg in (MANY, MANY | 1024, MANY | EAGER, MANY | 1024 | 524288):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_beam(ob, r["N"], r["B"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert path.tolist() == r["path"], dbg
        assert score == np.float32(r["score"])
        assert rc == (decoder.WARN_BEAM_MISS if -1 in r["path"] else 0)


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_beam_q16_filter_kernel_matches_golden(ctxs, g, r):
    """FV_OPT_DEBUG bit 9 forces beam_step_q16 (filter on the 16-bit table + float64 refine), which the
    library otherwise uses for large launches only (bit 26: in its 8-wave workgroup form, otherwise taken by launches with
    more workgroups than the chip holds at once; bit 25: never); bit 8 forces the float64 kernel."""
    fv, ob = ctxs(g)
    for dbg in (512, 512 | (1 << 26), 512 | (1 << 25), 256, 256 | (1 << 25)):     # (256: beams up to 64 run four-wave workgroups, with bit 25 sixteen)
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_beam(ob, r["N"], r["B"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert path.tolist() == r["path"] and score == np.float32(r["score"])
        assert rc == (decoder.WARN_BEAM_MISS if -1 in r["path"] else 0)


@pytest.mark.parametrize("K,M,T,N,B,seed,prob", [(300, 11, 70, 4, 20, 201, 0.15), (1000, 50, 40, 3, 128, 202, 0.1),
                                                 (65, 5, 129, 8, 65, 203, 0.3), (2049, 20, 24, 1, 500, 204, 0.05),
                                                 (130, 4, 90, 16, 2, 205, 0.5), (700, 9, 33, 5, 699, 206, 0.2),
                                                 # K > 4096 and K > 16384: the 16- and 64-round instantiations of topb_select
                                                 (5000, 20, 20, 3, 100, 207, 0.05), (16500, 10, 10, 1, 300, 208, 0.02)])
def test_beam_matches_oracle_fresh_inputs(K, M, T, N, B, seed, prob):
    import modelgen
    spec = dict(kind="data_script", K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, N, B)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    for dbg in (0, 512, EAGER, EAGER | 512):     # library's choice of beam step kernel, then the 16-bit filter kernel forced; lazy / eager replays
        fv.set_option(decoder.OPT_DEBUG, dbg)
        path, score, rc = fv.decode_beam(ob, N, B)
        assert path.tolist() == opath.tolist() and score == oscore and rc == orc, dbg
    fv.close()


@pytest.mark.parametrize("kind,K,M,T,N,B,seed,prob", [("data_script", 7000, 8, 60, 4, 64, 241, 0.05), ("ties_semi", 7000, 4, 24, 3, 64, 242, 0.5),
                                                      ("data_script", 16500, 10, 24, 8, 300, 243, 0.02)])
def test_beam_four_wave_select_equals_workgroup_select(kind, K, M, T, N, B, seed, prob):
    """Candidate lists of up to 2048 entries are selected by four waves (the other twelve leave the kernel at once);
    FV_OPT_DEBUG bit 15 keeps the whole-workgroup form.  Both must give the oracle's bits, and the same statistics
    (selects on a list, exact replays)."""
    import modelgen
    spec = dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, N, B)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    seen = []
    for dbg in (0, 32768, 512, 512 | 32768):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        path, score, rc = fv.decode_beam(ob, N, B)
        assert path.tolist() == opath.tolist() and score == oscore and rc == orc, dbg
        st = fv.stats()
        seen.append((st["beam_cand_selects"], st["beam_exact_sets"]))
    fv.close()
    assert seen[0] == seen[1] and seen[2] == seen[3]
    assert kind != "data_script" or seen[0][0] > 0


@pytest.mark.parametrize("kind,K,M,T,N,B,seed,prob", [("data_script", 3000, 8, 96, 8, 64, 251, 0.1), ("ties_semi", 2500, 4, 64, 16, 40, 252, 0.5),
                                                      ("data_script", 7000, 8, 50, 5, 100, 253, 0.05)])
def test_beam_pass_groups_on_several_streams_equal_one_stream(kind, K, M, T, N, B, seed, prob):
    """Big steps deal the passes of a generation to four streams (the selects / replays of one group run under the step
    kernels of the others).  FV_OPT_DEBUG bit 17 forces that for any size, bit 16 forbids it: same bits as the oracle
    either way, N = 5 / 8 / 16 give generations of 1..64 passes (groups of unequal sizes and lengths)."""
    import modelgen
    spec = dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, N, B)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    for dbg in (131072, 65536, 131072 | 512, 131072 | 1024, 0, 524288, 131072 | 524288, 131072 | EAGER):     # bit 19: every layout rebuilt, always
        fv.set_option(decoder.OPT_DEBUG, dbg)
        for rep in range(2):
            path, score, rc = fv.decode_beam(ob, N, B)
            assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (dbg, rep)
    fv.close()


@pytest.mark.parametrize("kind,K,M,T,N,B,seed", [("ties_semi", 1200, 4, 40, 4, 50, 221), ("ties_all", 600, 4, 30, 3, 33, 222),
                                                 ("ties_semi", 5000, 4, 16, 1, 200, 223)])
def test_beam_tie_heavy_models_match_oracle(kind, K, M, T, N, B, seed):
    """Few distinct probabilities => many equal scores: nearly every selection has duplicates at the cut (exact heap
    replay inside topb_select), many cells have tied maxima (tie_fixup), beam misses occur."""
    import modelgen
    spec = dict(kind=kind, K=K, M=M, T=T, prob=0.5, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, N, B)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    for dbg in (0, 512, 524288, EAGER, EAGER | 512, MANY, MANY | 1024, NOCUT, NOCUT | 512, 512 | (1 << 26)):    # bit 19: the layouts of every step and the tie fix-up run unconditionally
        fv.set_option(decoder.OPT_DEBUG, dbg)
        path, score, rc = fv.decode_beam(ob, N, B)
        assert path.tolist() == opath.tolist() and score == oscore and rc == orc
        if dbg == 0 and kind == "ties_all":
            # default mode: the first back-track met a tagged cell, so the gated heap_build_all / tie_fixup / second
            # walk did run (the branch must not ship untested)
            assert fv.stats()["beam_ties"] > 0
    assert fv.stats()["beam_exact_sets"] > 0
    fv.close()


def test_beam_tie_gate_stays_shut_on_a_tie_free_model():
    """The other side of the gate: a generate_data model whose back-tracked paths meet no tied cell never rebuilds
    the layouts (beam_ties counts the cells tie_fixup re-decided: none)."""
    import modelgen
    spec = dict(kind="data_script", K=700, M=9, T=40, prob=0.2, seed=261)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, 4, 60)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    path, score, rc = fv.decode_beam(ob, 4, 60)
    assert path.tolist() == opath.tolist() and score == oscore and rc == orc
    assert fv.stats()["beam_ties"] == 0
    fv.close()


@pytest.mark.parametrize("kind,K,M,T,N,B,seed,prob", [("ties_semi", 7000, 4, 24, 3, 64, 231, 0.5), ("ties_all", 6500, 4, 20, 1, 40, 232, 0.5),
                                                      ("data_script", 9000, 6, 40, 4, 100, 233, 0.01)])
def test_beam_candidate_lists_under_ties_and_bad_predictions(kind, K, M, T, N, B, seed, prob):
    """K >= 3 * 2048 switches the selects' candidate lists on (beam_step's epilogue collects the scores above a predicted
    lower bound of the next cut).  Tie-heavy models give a zero beam spread and jumpy cut values (the predictor is then
    wrong: lists too short or overflowing are ignored), a very sparse model leaves most scores at -FLT_MAX.  Every margin
    from "no list is ever long enough" to "every list overflows" must give the oracle's bits; FV_OPT_DEBUG bit 10 turns
    the lists off."""
    import modelgen
    spec = dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, N, B)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    used = 0
    for margin, dbg in ((500, 0), (0, 0), (100000, 0), (500, 512), (500, 1024)):
        fv.set_option(decoder.OPT_SEL_MARGIN, margin)
        fv.set_option(decoder.OPT_DEBUG, dbg)
        path, score, rc = fv.decode_beam(ob, N, B)
        assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (margin, dbg)
        used += fv.stats()["beam_cand_selects"] if dbg != 1024 else 0
        assert dbg != 1024 or fv.stats()["beam_cand_selects"] == 0
    fv.close()
    assert kind != "data_script" or used > 0          # the generate_data model does use its lists


def test_beam_long_sequence_with_chains_of_undecided_steps_equals_oracle():
    """T = 1500 at K = 20000: scores near -17000 have float spacings of 2e-3, most cuts fall on duplicated values, runs of
    undecided steps form, and ~70 selections find a doubtful column inside their beam.  The resolve code then decides the run
    only back to a step whose replay provably ignores its doubtful columns (replay_safe; FV_OPT_DEBUG bit 23: the whole run):
    fewer exact replays, the same bits."""
    import modelgen
    spec = dict(kind="sparse_fast", K=20000, M=8, T=1500, prob=0.05, seed=77)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, 4, 400)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    seen = {}
    for dbg in (0, NOCUT, 512):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        path, score, rc = fv.decode_beam(ob, 4, 400)
        assert path.tolist() == opath.tolist() and score == oscore and rc == orc, dbg
        st = fv.stats()
        seen[dbg] = (st["beam_chain_cuts"], st["beam_exact_sets"], st["beam_reach_events"])
    fv.close()
    assert seen[0][0] > 0 and seen[NOCUT][0] == 0 and seen[0][1] < seen[NOCUT][1] and seen[0][2] > 0, seen


def test_beam_equal_to_K_is_full_decode():
    """B = K keeps every state: same path and score as the full-state decoder."""
    import modelgen
    spec = dict(kind="data_script", K=128, M=10, T=64, prob=0.3, seed=77)
    A, Bm, Pi, ob = modelgen.model32(spec)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    fpath, fscore, _ = fv.decode_full(ob, 4)
    om = oracle.OracleModel(A, Bm, Pi)
    bo, so, _, _ = om.beam_decode(ob, 4, 128)
    bpath, bscore, _ = fv.decode_beam(ob, 4, 128)
    assert bpath.tolist() == bo.tolist() and bscore == so
    assert bscore == fscore
    fv.close()


def test_beam_argument_errors():
    fv = decoder.FlashViterbi(0)
    A = np.full((8, 8), 0.125, np.float32)
    fv.set_model(A, A[:, :2] * 4, A[0])
    ob = np.zeros(10, np.int32)
    for bad in (1, 9):
        with pytest.raises(decoder.FlashVitError):
            fv.decode_beam(ob, 1, bad)
    fv.close()
