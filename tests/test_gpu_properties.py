"""GPU: size-independent properties at BASELINE's full sizes (too large for the oracle in a test run).

cfg3 size (K=3965, T=4096, N=8): the decoded path is a valid state sequence (every transition and
emission has non-zero probability), its log-score re-computed on the host in float64 equals the
returned score to 1e-5 relative, and every kernel / schedule agrees with every other bit for bit."""
import numpy as np
import pytest

import modelgen
from flash_viterbi_amd import decoder

pytestmark = pytest.mark.gpu


def path_logscore64(A, B, Pi, ob, path):
    p = np.asarray(path, dtype=np.int64)
    with np.errstate(divide="ignore"):
        s = np.log(np.float64(Pi[p[0]])) + np.log(np.float64(B[p[0], ob[0]]))
        s += np.log(A[p[:-1], p[1:]].astype(np.float64)).sum()
        s += np.log(B[p[1:], ob[1:]].astype(np.float64)).sum()
    return float(s)


@pytest.fixture(scope="module")
def cfg3():
    spec = dict(kind="data_script", K=3965, M=50, T=4096, prob=0.112, seed=12)
    A, B, Pi, ob = modelgen.model32(spec)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    yield A, B, Pi, ob, fv
    fv.close()


def test_cfg3_path_is_valid_and_score_consistent(cfg3):
    A, B, Pi, ob, fv = cfg3
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
    path, score, rc = fv.decode_full(ob, 8, decoder.MODE_REFERENCE)
    assert rc == 0 and path.min() >= 0 and path.max() < 3965
    s64 = path_logscore64(A, B, Pi, ob, path)
    assert np.isfinite(s64), "decoded path uses a zero-probability transition or emission"
    assert abs(s64 - float(score)) <= 1e-5 * abs(s64)          # north_star tolerance for log-scores
    st = fv.stats()
    assert st["passes"] == 2047 and st["kernel"] == decoder.KERNEL_SPARSE_Q16


def test_cfg3_all_kernels_and_schedules_agree(cfg3):
    A, B, Pi, ob, fv = cfg3
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
    ref_path, ref_score, _ = fv.decode_full(ob, 8, decoder.MODE_REFERENCE)
    for kernel in (decoder.KERNEL_Q16_REFINE, decoder.KERNEL_U16_REFINE, decoder.KERNEL_F32_REFINE, decoder.KERNEL_F64_STREAM):
        fv.set_option(decoder.OPT_KERNEL, kernel)
        path, score, rc = fv.decode_full(ob, 8, decoder.MODE_REFERENCE)
        assert rc == 0 and (path == ref_path).all() and score == ref_score, kernel
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_U16_REFINE)
    fv.set_option(decoder.OPT_DEBUG, 16384)          # packed 16-bit filter for the batched launches too
    path, score, rc = fv.decode_full(ob, 8, decoder.MODE_REFERENCE)
    fv.set_option(decoder.OPT_DEBUG, 0)
    assert rc == 0 and (path == ref_path).all() and score == ref_score
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
    # other segmentations replay different passes but must land on the same optimum here
    for n in (1, 16):
        path, score, rc = fv.decode_full(ob, n, decoder.MODE_REFERENCE)
        assert rc == 0 and score == ref_score
        assert abs(path_logscore64(A, B, Pi, ob, path) - float(score)) <= 1e-5 * abs(float(score))
    sp_path, sp_score, _ = fv.decode_full(ob, 8, decoder.MODE_SINGLE_PASS)
    assert sp_score == ref_score and abs(path_logscore64(A, B, Pi, ob, sp_path) - float(sp_score)) <= 1e-5 * abs(float(sp_score))
    v_path, v_score, _ = fv.decode_vanilla(ob)
    assert abs(float(v_score) - float(ref_score)) <= 1e-5 * abs(float(ref_score))


def test_cfg3_beam_path_valid_and_not_better_than_full(cfg3):
    A, B, Pi, ob, fv = cfg3
    fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
    _, full_score, _ = fv.decode_full(ob[:512], 8, decoder.MODE_REFERENCE)
    for beam in (64, 512):
        path, score, rc = fv.decode_beam(ob[:512], 8, beam, decoder.MODE_REFERENCE)
        assert rc in (0, decoder.WARN_BEAM_MISS)
        assert (rc == decoder.WARN_BEAM_MISS) == bool((path < 0).any())     # the reference prints -1 after a miss
        assert path.max() < 3965
        if rc == 0:
            s64 = path_logscore64(A, B, Pi, ob[:512], path)
            assert np.isfinite(s64)
            assert s64 <= float(full_score) + 1e-5 * abs(float(full_score))   # a pruned search cannot beat the optimum


@pytest.mark.parametrize("prob,expect", [(0.9, decoder.KERNEL_U16_REFINE), (0.5, decoder.KERNEL_U16_REFINE),
                                         (0.3, decoder.KERNEL_SPARSE_Q16)])
def test_auto_kernel_choice_by_density_and_parity(prob, expect):
    """Dense models take the dense 16-bit table, sparse ones the walk; both equal the oracle."""
    import oracle
    spec = dict(kind="data_script", K=600, M=12, T=80, prob=prob, seed=41)
    A, B, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, B, Pi)
    opath, oscore, _, _ = om.full_decode(ob, 4)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    path, score, rc = fv.decode_full(ob, 4)
    st = fv.stats()
    assert st["kernel"] == expect and abs(st["density"] - prob) < 0.03
    assert rc == 0 and path.tolist() == opath.tolist() and score == oscore
    for kernel in (decoder.KERNEL_SPARSE_Q16, decoder.KERNEL_Q16_REFINE, decoder.KERNEL_F16_REFINE, decoder.KERNEL_U16_REFINE):
        fv.set_option(decoder.OPT_KERNEL, kernel)
        path, score, rc = fv.decode_full(ob, 4)
        assert path.tolist() == opath.tolist() and score == oscore
    fv.set_option(decoder.OPT_DEBUG, 16384)          # packed 16-bit filter for the batched launches too
    path, score, rc = fv.decode_full(ob, 4)
    fv.set_option(decoder.OPT_DEBUG, 0)
    assert path.tolist() == opath.tolist() and score == oscore
    bo, bs, _, brc = om.beam_decode(ob, 4, 40)
    bp, bsc, rc = fv.decode_beam(ob, 4, 40)
    assert bp.tolist() == bo.tolist() and bsc == bs and rc == brc
    fv.close()


def test_no_accepted_option_value_changes_a_result():
    """nothing reachable through include/flashvit.h may return a wrong path with rc 0.  Every bit of
    FV_OPT_DEBUG is either refused (the timing switches that leave a part of a kernel out live in the separate timing
    build) or speed-only: each one alone, and all accepted ones together, decode the goldens' paths; the same for every
    kernel choice, batch limit and select margin."""
    from conftest import golden_model, load_goldens
    g = next(x for x in load_goldens() if x["name"] == "cfg1_K128_T256")
    A, B, Pi, ob = golden_model(g)
    full = next(r for r in g["runs"] if r["algo"] == "flash" and r["N"] == 8)
    beam = next(r for r in g["runs"] if r["algo"] == "flashbs" and r["N"] == 8)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    try:
        accepted = 0
        for bit in range(27):
            v = 1 << bit
            if v & decoder.DEBUG_TIMING_ONLY:
                with pytest.raises(decoder.FlashVitError):
                    fv.set_option(decoder.OPT_DEBUG, v)
                continue
            accepted |= v
        for v in [1 << b for b in range(27) if (1 << b) & accepted] + [accepted, accepted & ~(256 | 65536), accepted & ~(1 << 25), 0]:
            fv.set_option(decoder.OPT_DEBUG, v)
            for kernel in (decoder.KERNEL_AUTO, decoder.KERNEL_U16_REFINE, decoder.KERNEL_Q16_REFINE, decoder.KERNEL_SPARSE_Q16):
                fv.set_option(decoder.OPT_KERNEL, kernel)
                path, score, rc = fv.decode_full(ob, 8)
                assert rc == 0 and path.tolist() == full["path"] and score == np.float32(full["score"]), (v, kernel)
            path, score, rc = fv.decode_beam(ob, 8, beam["B"])
            assert path.tolist() == beam["path"] and score == np.float32(beam["score"]), v
        fv.set_option(decoder.OPT_DEBUG, 0)
        for key, vals in ((decoder.OPT_MAX_BATCH, (1, 2, 3, 8)), (decoder.OPT_SEL_MARGIN, (0, 1, 100000)), (decoder.OPT_PROFILE, (1, 0))):
            for v in vals:
                fv.set_option(key, v)
                path, score, rc = fv.decode_full(ob, 8)
                assert rc == 0 and path.tolist() == full["path"]
                path, score, rc = fv.decode_beam(ob, 8, beam["B"])
                assert path.tolist() == beam["path"]
        for key, bad in ((decoder.OPT_DEBUG, -1), (decoder.OPT_DEBUG, 1 << 27), (decoder.OPT_KERNEL, 7), (decoder.OPT_MAX_BATCH, 9), (99, 0)):
            with pytest.raises(decoder.FlashVitError):
                fv.set_option(key, bad)
    finally:
        fv.close()


def test_one_context_many_shapes_in_any_order():
    """Workspace buffers (score rows, back-pointers, the pinned result block, beam member lists, doubtful-column lists, the
    multi-device gather) grow on demand and are reused: one context — plain and multi-device — decodes sequences of
    different length, splits and beam widths in shuffled order, then a second model of another size, and every result
    equals the oracle's."""
    import oracle
    rs = np.random.RandomState(77)
    models = []
    for K, M, seed in ((230, 7, 311), (90, 4, 312)):
        A, Bm, Pi, _ = modelgen.model32(dict(kind="data_script", K=K, M=M, T=8, prob=0.2, seed=seed))
        models.append((A, Bm, Pi, oracle.OracleModel(A, Bm, Pi), M))
    for devices in (0, [0, 0]):
        fv = decoder.FlashViterbi(devices)
        try:
            for A, Bm, Pi, om, M in models:
                fv.set_model(A, Bm, Pi)
                shapes = [(T, N, B) for T in (9, 64, 33, 301, 17) for N, B in ((1, 0), (4, 0), (3, 11), (1, 40), (8, 25))]
                rs.shuffle(shapes)
                for T, N, B in shapes:
                    if T < 2 * N or (N > 2 and T == 2 * N):
                        continue
                    ob = rs.randint(0, M, T).astype(np.int32)
                    if B:
                        opath, oscore, _, orc = om.beam_decode(ob, N, B)
                        path, score, rc = fv.decode_beam(ob, N, B)
                    else:
                        opath, oscore, _, orc = om.full_decode(ob, N)
                        path, score, rc = fv.decode_full(ob, N)
                    assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (devices, T, N, B)
        finally:
            fv.close()
    for m in models:
        m[3].close()
