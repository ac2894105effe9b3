"""GPU parity of the full-state FLASH path, through the C-ABI (libflashvit.so).

Bar: decoded path bit-exact (int32), final score float32-equal, against (1) golden
vectors from the reference-built binaries and (2) the oracle on fresh seeded inputs."""
import numpy as np
import pytest

import oracle
from conftest import golden_model, golden_runs
from flash_viterbi_amd import decoder

pytestmark = pytest.mark.gpu

PAIRS, IDS = golden_runs(include_big=True, algo="flash")
KERNELS = [decoder.KERNEL_F64_STREAM, decoder.KERNEL_F32_REFINE, decoder.KERNEL_F16_REFINE, decoder.KERNEL_Q16_REFINE,
           decoder.KERNEL_SPARSE_Q16, decoder.KERNEL_U16_REFINE]
KIDS = ["f64stream", "f32refine", "f16refine", "q16refine", "sparseq16", "u16refine"]


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


@pytest.mark.parametrize("kernel", KERNELS, ids=KIDS)
@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_reference_mode_matches_golden(ctxs, g, r, kernel):
    fv, ob = ctxs(g)
    fv.set_option(decoder.OPT_KERNEL, kernel)
    path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
    assert rc == 0
    assert path.tolist() == r["path"]
    assert score == np.float32(r["score"])
    st = fv.stats()
    assert st["kernel"] == kernel
    assert decoder.reference_memory_bytes(fv.K, len(ob), r["N"]) == r["memory"]


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_packed_u16_filter_for_batched_launches_matches_golden(ctxs, g, r):
    """FV_KERNEL_U16_REFINE runs generations of more than four right-hand passes as batches of four on three streams
    with the packed 16-bit filter (co-resident workgroups of different launches), everything else batched with the
    f32 filter; FV_OPT_DEBUG bit 18 turns the streams off, bit 14 forces the packed filter for every batched launch,
    bit 13 selects its 16-wave workgroup form.  (The default form is what every KERNEL_U16_REFINE / AUTO test runs.)"""
    fv, ob = ctxs(g)
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_U16_REFINE)
    for dbg in (262144, 262144 | 16384, 16384, 16384 | 8192, 16384 | 4, 262144 | 16384 | 8192):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])
    # the sparse walk forks its right-hand generations the same way: bit 18 is its single-stream form
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_SPARSE_Q16)
    fv.set_option(decoder.OPT_DEBUG, 262144)
    try:
        path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
    finally:
        fv.set_option(decoder.OPT_DEBUG, 0)
    assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])


@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_single_pass_mode_matches_golden_empirically(ctxs, g, r):
    """Not guaranteed by construction (different rounding history for right-hand tasks) but it
    holds on every fixture; a failure here is information, not necessarily a bug."""
    fv, ob = ctxs(g)
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
    path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_SINGLE_PASS)
    assert rc == 0 and score == np.float32(r["score"])
    if g["spec"]["kind"] == "data_script":
        assert path.tolist() == r["path"]


@pytest.mark.parametrize("K,M,T,N,seed,prob", [(300, 11, 70, 4, 101, 0.15), (1000, 50, 40, 3, 102, 0.15),
                                               (65, 5, 129, 8, 103, 0.15), (2049, 20, 24, 1, 104, 0.15),
                                               (16, 3, 200, 16, 105, 0.8),
                                               # more 16-column tiles than CUs: the two-workgroups-per-CU variant
                                               (4500, 12, 14, 3, 106, 0.05)])
def test_reference_mode_matches_oracle_fresh_inputs(K, M, T, N, seed, prob):
    import modelgen
    spec = dict(kind="data_script", K=K, M=M, T=T, prob=prob, seed=seed)
    A, B, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    opath, oscore, ocells, orc = om.full_decode(ob, N)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    for kernel in KERNELS:
        for batch in (1, 8):
            fv.set_option(decoder.OPT_KERNEL, kernel)
            fv.set_option(decoder.OPT_MAX_BATCH, batch)
            path, score, rc = fv.decode_full(ob, N)
            assert rc == 0 and path.tolist() == opath.tolist() and score == oscore
    fv.close()


def test_model_above_one_falls_back_to_f64_stream():
    """Unnormalised weights (> 1) void the F32_REFINE bracket; AUTO must pick F64_STREAM and still match."""
    rs = np.random.RandomState(5)
    K, M, T = 96, 6, 50
    A = (rs.uniform(0, 3, (K, K)) * (rs.uniform(0, 1, (K, K)) < 0.3)).astype(np.float32)
    A[np.arange(K), rs.randint(0, K, K)] = 1.5
    B = rs.uniform(0.1, 2, (K, M)).astype(np.float32)
    Pi = rs.uniform(0.1, 1, K).astype(np.float32)
    ob = rs.randint(0, M, T).astype(np.int32)
    om = oracle.OracleModel(A, B, Pi)
    opath, oscore, _, _ = om.full_decode(ob, 4)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    path, score, rc = fv.decode_full(ob, 4)
    assert fv.stats()["kernel"] == decoder.KERNEL_F64_STREAM
    assert path.tolist() == opath.tolist() and score == oscore
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_F32_REFINE)
    with pytest.raises(decoder.FlashVitError):
        fv.decode_full(ob, 4)
    fv.close()


def test_argument_errors():
    fv = decoder.FlashViterbi(0)
    with pytest.raises(decoder.FlashVitError):
        fv.decode_full(np.zeros(4, np.int32), 1)            # no model yet
    A = np.full((4, 4), 0.25, np.float32)
    fv.set_model(A, A[:, :2] * 2, A[0])
    with pytest.raises(decoder.FlashVitError):
        fv.decode_full(np.array([0, 5, 1], np.int32), 1)    # symbol out of range
    with pytest.raises(decoder.FlashVitError):
        fv.decode_full(np.zeros(8, np.int32), 4)            # T == 2N
    bad = A.copy(); bad[1, 1] = -0.1
    with pytest.raises(decoder.FlashVitError):
        fv.set_model(bad, A[:, :2] * 2, A[0])
    fv.close()


@pytest.mark.parametrize("kernel", KERNELS, ids=KIDS)
@pytest.mark.parametrize("bits", [2, 4, 8, 2 | 4 | 8], ids=["noreverse", "altloads", "fulllast", "all"])
def test_tuning_switches_do_not_change_results(ctxs, kernel, bits):
    """Sweep direction, load schedule and the single-column last step are speed knobs only."""
    pairs = [(g, r) for g, r in PAIRS if g["name"] in ("cfg2_K3965_T256", "ties_semi_K96_T80", "ds_K200_T100")]
    for g, r in pairs:
        fv, ob = ctxs(g)
        fv.set_option(decoder.OPT_KERNEL, kernel)
        fv.set_option(decoder.OPT_DEBUG, bits)
        try:
            path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])


SLABS = 1 << 21      # FV_OPT_DEBUG bit 21: the float64 kernel sweeps the source rows in three slabs


@pytest.mark.parametrize("kernel", [decoder.KERNEL_F64_STREAM, decoder.KERNEL_Q16_REFINE, decoder.KERNEL_F32_REFINE, decoder.KERNEL_F16_REFINE],
                         ids=["f64stream", "q16refine", "f32refine", "f16refine"])
@pytest.mark.parametrize("g,r", PAIRS, ids=IDS)
def test_step_kernel_in_slabs_of_source_rows_matches_golden(ctxs, g, r, kernel):
    """Models whose score row does not fit LDS and that the packed 16-bit kernel cannot take (K > 65536, or entries above 1
    beyond K ~ 40100) run trellis_step one slab of source rows per launch — the float64 kernel, or the f32 filter on the 16-bit
    table with its refine inside the slab — each launch merging into the exact result of the slabs below it (strict '>'
    across slabs = the reference's ascending scan).  Forced here at every golden's size, batched (right-hand generations)
    and not."""
    fv, ob = ctxs(g)
    fv.set_option(decoder.OPT_KERNEL, kernel)
    for dbg, batch in ((SLABS, 8), (SLABS, 1), (SLABS | 8, 3)):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        fv.set_option(decoder.OPT_MAX_BATCH, batch)
        try:
            path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
            fv.set_option(decoder.OPT_MAX_BATCH, 8)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"]), (dbg, batch)


VPAIRS, VIDS = golden_runs(include_big=True, algo="vanilla")


@pytest.mark.parametrize("g,r", VPAIRS, ids=VIDS)
def test_vanilla_baseline_matches_reference_vanilla_binary(ctxs, g, r):
    """fv_decode_vanilla against goldens from Base_line/C implementations/vanilla Viterbi.c."""
    fv, ob = ctxs(g)
    for dbg in (0, SLABS):                              # (the baseline's expression in slabs of source rows as well)
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_vanilla(ob)
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])


CPAIRS, CIDS = golden_runs(include_big=True, algo="checkpoint")


@pytest.mark.parametrize("g,r", CPAIRS, ids=CIDS)
def test_checkpoint_baseline_matches_reference_checkpoint_binary(ctxs, g, r):
    """fv_decode_checkpoint against goldens from Base_line/C implementations/checkpoint Viterbi.c
    (step 0 = floor(sqrt(T)), what its main passes; other steps through the function's own argument)."""
    fv, ob = ctxs(g)
    for dbg in (0, SLABS):
        fv.set_option(decoder.OPT_DEBUG, dbg)
        try:
            path, score, rc = fv.decode_checkpoint(ob, r["step"])
        finally:
            fv.set_option(decoder.OPT_DEBUG, 0)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])
    assert decoder.checkpoint_memory_bytes(fv.K, len(ob), r["step"]) == r["memory"]


@pytest.mark.parametrize("step", [0, 1, 2, 9, 10, 89, 90, 500])
def test_checkpoint_equals_vanilla_and_oracle_for_every_step(step):
    """Segment lengths 1, ragged last segment, one segment only (step >= T): same bits as vanilla."""
    import modelgen
    spec = dict(kind="data_script", K=300, M=9, T=90, prob=0.15, seed=311)
    A, B, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    cp, cs, _ = om.checkpoint_decode(ob, step)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    vpath, vscore, _ = fv.decode_vanilla(ob)
    path, score, rc = fv.decode_checkpoint(ob, step)
    assert rc == 0 and path.tolist() == cp.tolist() and score == cs
    assert path.tolist() == vpath.tolist() and score == vscore
    fv.close()


def test_vanilla_matches_oracle_and_flash_path_on_fresh_input():
    import modelgen
    spec = dict(kind="data_script", K=700, M=13, T=90, prob=0.1, seed=301)
    A, B, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    vp, vs, _ = om.vanilla_decode(ob)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    path, score, rc = fv.decode_vanilla(ob)
    assert path.tolist() == vp.tolist() and score == vs
    fpath, fscore, _ = fv.decode_full(ob, 8)
    assert fpath.tolist() == path.tolist()          # same optimum, possibly last-ulp different score
    assert abs(float(fscore) - float(score)) <= 1e-5 * abs(float(score))
    fv.close()


@pytest.mark.parametrize("nranks", [2, 3, 8])
def test_partitioned_decode_merges_to_single_rank_result(ctxs, nranks):
    """fv_set_partition: each 'rank' (here: contexts on one GPU, one after the other) decodes the
    whole-sequence pass plus its own segments; fv_merge_paths of the per-rank arrays equals the
    single-rank decode.  Full and beam variants."""
    g = next(g for g, r in PAIRS if g["name"] == "cfg1_K128_T256")
    A, B, Pi, ob = golden_model(g)
    want_full = next(r for r in g["runs"] if r["algo"] == "flash" and r["N"] == 8)["path"]
    want_beam = next(r for r in g["runs"] if r["algo"] == "flashbs" and r["N"] == 8 and r["B"] == 32)["path"]
    parts_full, parts_beam = [], []
    for rank in range(nranks):
        fv = decoder.FlashViterbi(0)
        fv.set_model(A, B, Pi)
        fv.set_partition(rank, nranks)
        parts_full.append(fv.decode_full(ob, 8)[0])
        parts_beam.append(fv.decode_beam(ob, 8, 32)[0])
        st = fv.stats()
        assert st["ranks"] == nranks and st["passes"] < 127      # 127 passes in all at T=256, N=8
        fv.close()
    T = len(ob)
    assert decoder.merge_paths(T, 8, nranks, np.stack(parts_full)).tolist() == want_full
    assert decoder.merge_paths(T, 8, nranks, np.stack(parts_beam)).tolist() == want_beam


@pytest.mark.parametrize("K,M,T,N", [(1, 2, 5, 1), (2, 2, 9, 1), (3, 3, 17, 4), (17, 2, 300, 16), (33, 50, 2, 1)])
def test_tiny_and_odd_sizes_match_oracle(K, M, T, N):
    import modelgen
    spec = dict(kind="data_script", K=K, M=M, T=T, prob=1.0 if K < 4 else 0.6, seed=7 + K)
    A, B, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    opath, oscore, _, _ = om.full_decode(ob, N)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    for kernel in KERNELS:
        fv.set_option(decoder.OPT_KERNEL, kernel)
        path, score, rc = fv.decode_full(ob, N)
        assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, kernel
    vp, vs, _ = om.vanilla_decode(ob)
    gp, gs, _ = fv.decode_vanilla(ob)
    assert gp.tolist() == vp.tolist() and gs == vs
    if K >= 2:
        for beam in sorted({2, K}):
            bo, bs, _, brc = om.beam_decode(ob, N, beam)
            bp, bsc, rc = fv.decode_beam(ob, N, beam)
            assert bp.tolist() == bo.tolist() and bsc == bs and rc == brc
    fv.close()
