"""GPU: adversarial inputs for the filter + refine kernels.

The 2- and 4-byte table kernels return the reference's bits because of an error bracket: every cell
that can attain the exact maximum lies inside a window of the filter's best, and everything inside the window is
re-evaluated with the reference expression (src/FLASH_Viterbi_multithread.c:167-173).  Every other fixture draws its
transition weights from U(0.01, 1); here the inputs are built to sit where that argument is thinnest:

  (a) wide dynamic range  entries of {1e-16, 1e-8, 1e-3, U(0.1, 1)} (legal '%.16f' text): score rows spread wider
                          than the 16-bit code range (65534 steps = max|log A|), so whole columns have no live
                          predecessor inside it — the packed kernel's saturated branch (`refine_saturated` counts it);
  (b) binade crossings    long sequences whose scores cross -1024, -2048, -4096 ...: candidates on the two sides of
                          a power of two have different float spacings (tie-heavy models and generate_data models);
  (c) directed pairs      per column two predecessors whose exact values tie or differ by one float step, placed in
                          one register pair / one lane / different waves, the true winner being the one the 16-bit
                          codes rank LOWER; checked column by column through one-step decodes.

Expected values come from the oracle (pinned to the reference binaries by tests/test_oracle_golden.py)."""
import math

import numpy as np
import pytest

import modelgen
import oracle
from flash_viterbi_amd import decoder, hostio

pytestmark = pytest.mark.gpu

FULL_KERNELS = [decoder.KERNEL_F64_STREAM, decoder.KERNEL_F32_REFINE, decoder.KERNEL_F16_REFINE, decoder.KERNEL_Q16_REFINE,
                decoder.KERNEL_SPARSE_Q16, decoder.KERNEL_U16_REFINE]
# forced forms of the packed 16-bit kernel (FV_OPT_DEBUG): one stream, packed filter for every batched launch,
# its 16-wave form, the alternate load schedule
U16_FORMS = (0, 262144, 16384, 16384 | 8192, 16384 | 4, 262144 | 16384 | 8192)


def _classes(rs, shape, probs=(0.25, 0.25, 0.25, 0.25)):
    """entries of {1e-16, 1e-8, 1e-3, U(0.1, 1)}"""
    cls = rs.choice(4, size=shape, p=probs)
    u = rs.uniform(0.1, 1.0, size=shape)
    return np.where(cls == 0, 1e-16, np.where(cls == 1, 1e-8, np.where(cls == 2, 1e-3, u)))


def wide_model(kind, K, M, T, seed):
    rs = np.random.RandomState(seed)
    if kind == "wideA":
        # every transition present, weights over 16 decades: the table's step is large (36.8 / 65534), the filter's
        # window wide, the score rows stay inside the code range
        A = _classes(rs, (K, K))
        Bm = _classes(rs, (K, M), (0.1, 0.2, 0.3, 0.4))
        Pi = _classes(rs, (K,), (0.1, 0.2, 0.3, 0.4))
    elif kind == "wideB":
        # sparse transitions of ordinary weights (code range = |log 0.1| = 2.3), emissions over 16 decades: most score
        # rows lie further below the row maximum than the code range reaches; columns all of whose in-edges come from
        # such rows have only saturated sums
        p = min(1.0, 6.0 / K)
        A = rs.uniform(0.1, 1.0, (K, K)) * (rs.uniform(0, 1, (K, K)) < p)
        A[np.arange(K), rs.randint(0, K, K)] = rs.uniform(0.1, 1.0, K)        # every state has a successor
        Bm = _classes(rs, (K, M))
        Pi = _classes(rs, (K,), (0.1, 0.2, 0.3, 0.4))
    elif kind == "wideAB":
        # both: sparse transitions over 16 decades (a third of the graph), emissions likewise
        A = _classes(rs, (K, K)) * (rs.uniform(0, 1, (K, K)) < 0.3)
        A[np.arange(K), rs.randint(0, K, K)] = 0.5
        Bm = _classes(rs, (K, M))
        Pi = _classes(rs, (K,))
    else:
        raise ValueError(kind)
    ob = rs.randint(0, M, T).astype(np.int32)
    return hostio.quantize_text16(A), hostio.quantize_text16(Bm), hostio.quantize_text16(Pi), ob


WIDE = [("wideA", 600, 6, 60, 401), ("wideB", 600, 6, 60, 402), ("wideAB", 600, 6, 60, 403),
        ("wideA", 4500, 5, 24, 404), ("wideB", 4500, 5, 24, 405)]


@pytest.mark.parametrize("kind,K,M,T,seed", WIDE)
def test_wide_dynamic_range_full_state(kind, K, M, T, seed):
    A, Bm, Pi, ob = wide_model(kind, K, M, T, seed)
    om = oracle.OracleModel(A, Bm, Pi)
    want = {N: om.full_decode(ob, N, check=False) for N in (1, 4)}
    om.close()
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    saturated = rescans = 0
    try:
        for N in (1, 4):
            opath, oscore, _, orc = want[N]
            for kernel in FULL_KERNELS:
                fv.set_option(decoder.OPT_KERNEL, kernel)
                for dbg in (U16_FORMS if kernel == decoder.KERNEL_U16_REFINE else (0,)):
                    for batch in (8, 1):
                        fv.set_option(decoder.OPT_DEBUG, dbg)
                        fv.set_option(decoder.OPT_MAX_BATCH, batch)
                        if orc < 0:
                            with pytest.raises(decoder.FlashVitError):
                                fv.decode_full(ob, N)
                            continue
                        path, score, rc = fv.decode_full(ob, N)
                        assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, (N, kernel, dbg, batch)
                        st = fv.stats()
                        if kernel == decoder.KERNEL_U16_REFINE:
                            saturated += st["refine_saturated"]
                            rescans += st["refine_rescan"]
    finally:
        fv.close()
    assert rescans > 0
    if kind == "wideB":      # (wideA / wideAB: 1e-16 transitions make the code range 36.8 wide: nothing lies beyond it)
        # the branch the suite had never executed before (fv_kernels.hip.inc, "saturated sums could matter")
        assert saturated > 0, "no column reached the end of the 16-bit code range: the saturated branch did not run"


@pytest.mark.parametrize("kind,K,M,T,seed,B", [("wideA", 600, 6, 60, 411, 64), ("wideB", 600, 6, 60, 412, 100),
                                               ("wideAB", 600, 6, 60, 413, 37), ("wideB", 4500, 5, 24, 414, 300)])
def test_wide_dynamic_range_beam(kind, K, M, T, seed, B):
    A, Bm, Pi, ob = wide_model(kind, K, M, T, seed)
    om = oracle.OracleModel(A, Bm, Pi)
    want = {N: om.beam_decode(ob, N, B, check=False) for N in (1, 4)}
    om.close()
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    try:
        for N in (1, 4):
            opath, oscore, _, orc = want[N]
            assert orc >= 0
            for dbg in (256, 512, 512 | (1 << 26), 0, 524288):          # float64 rows, 16-bit filter + refine (16- and 8-wave workgroups), library's choice, all layouts
                fv.set_option(decoder.OPT_DEBUG, dbg)
                path, score, rc = fv.decode_beam(ob, N, B)
                assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (N, dbg)
    finally:
        fv.close()


# ---------------------------------------------------------------- (b) binade crossings

LONG = [("ties_semi", 100, 4, 2500, 421, 0.5), ("ties_all", 96, 4, 2100, 422, 0.5), ("data_script", 300, 20, 3000, 423, 0.2),
        ("data_script", 130, 7, 4200, 424, 0.6)]


@pytest.mark.parametrize("kind,K,M,T,seed,prob", LONG)
def test_long_sequences_cross_binades_full_state(kind, K, M, T, seed, prob):
    """Scores fall by ~2-3 per step: T >= 2048 crosses -1024 ... -4096 (ties_*: every candidate of a column ties or
    nearly ties; data_script: ordinary near-ties).  All kernels, N = 1 (one pass over everything) and N = 8."""
    spec = dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    want = {N: om.full_decode(ob, N) for N in (1, 8)}
    om.close()
    assert min(w[1] for w in want.values()) < -2048.0
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    try:
        for N in (1, 8):
            opath, oscore, _, _ = want[N]
            for kernel in FULL_KERNELS:
                fv.set_option(decoder.OPT_KERNEL, kernel)
                for dbg in ((0, 16384, 16384 | 8192) if kernel == decoder.KERNEL_U16_REFINE else (0,)):
                    fv.set_option(decoder.OPT_DEBUG, dbg)
                    path, score, rc = fv.decode_full(ob, N)
                    assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, (N, kernel, dbg)
    finally:
        fv.close()


@pytest.mark.parametrize("kind,K,M,T,seed,prob,B", [("ties_semi", 100, 4, 2500, 431, 0.5, 30), ("ties_all", 96, 4, 2100, 432, 0.5, 17),
                                                    ("data_script", 300, 20, 3000, 433, 0.2, 50)])
def test_long_sequences_cross_binades_beam(kind, K, M, T, seed, prob, B):
    spec = dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    want = {N: om.beam_decode(ob, N, B) for N in (1, 8)}
    om.close()
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    try:
        for N in (1, 8):
            opath, oscore, _, orc = want[N]
            for dbg in (256, 512, 0):
                fv.set_option(decoder.OPT_DEBUG, dbg)
                path, score, rc = fv.decode_beam(ob, N, B)
                assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (N, dbg)
    finally:
        fv.close()


# ---------------------------------------------------------------- (c) directed pairs

def _f32(x):
    return np.float32(x)


def _cell(tmp, t1, a):
    """the reference's cell (FLASH:167-170) for one float32 transition weight a, evaluated as the C programs do"""
    s = _f32(tmp) + _f32(t1)                                   # float add
    return _f32(float(s) + math.log(float(_f32(a))))          # double add, one rounding


def directed_model(K, C, seed):
    """T = 2 probes.  Symbol 0 has B = 1 for every state, so the score row entering the step is T1[k] = (float)log Pi[k];
    symbol 1 + c makes column c the whole-sequence end state (every other emission is 1e-7 of it), so the decode of
    ob = [0, 1 + c] returns column c's argmax.  Column c holds two designated predecessors (a, b) whose exact values
    tie or differ by one float step, plus weaker and missing ones.  A[0][0] = 1e-5 pins max|log A| (= the table's
    step); every other weight is larger."""
    rs = np.random.RandomState(seed)
    Pi = hostio.quantize_text16(rs.uniform(0.05, 1.0, K))
    T1 = np.array([_f32(math.log(float(p))) for p in Pi], dtype=np.float32)
    A = np.zeros((K, K), dtype=np.float32)
    Bm = np.zeros((K, C + 1), dtype=np.float64)
    Bm[:, 0] = 1.0
    info = []
    for c in range(C):
        col = 1 + c                                          # column 0 keeps the pinning entry
        bc = rs.uniform(0.2, 1.0)
        Bm[:, 1 + c] = rs.uniform(0.5, 1.0, K) * 1e-7
        Bm[col, 1 + c] = bc
        tmp = _f32(math.log(float(hostio.quantize_text16(np.array([bc]))[0])))
        for _try in range(200):
            shape = c % 5
            a = int(rs.randint(0, K - 300))
            if shape == 0:   b = a ^ 1                       # the two halves of one packed register
            elif shape == 1: b = (a & ~7) | ((a + 2 + 2 * int(rs.randint(0, 3))) & 7)      # same 8-row lane load
            elif shape == 2: b = a + 32 * 8 * int(rs.randint(1, 2))                        # same lane, another row block
            elif shape == 3: b = a + 8 * int(rs.randint(1, 4))                             # another row group / wave
            else:            b = int(rs.randint(0, K))
            if b == a or b >= K:
                continue
            if rs.randint(0, 2):
                a, b = b, a
            alpha = _f32(rs.uniform(0.02, 0.9))
            ka = _cell(tmp, T1[a], alpha)
            rel = int(rs.randint(0, 3))                      # 0 tie, 1 b one float step above a, 2 one below
            target = ka if rel == 0 else np.nextafter(ka, _f32(np.inf) if rel == 1 else _f32(-np.inf), dtype=np.float32)
            sb = _f32(tmp) + T1[b]
            beta0 = math.exp(float(target) - float(sb))
            if not (2e-5 < beta0 < 0.999):
                continue
            cand = _f32(beta0)
            hits = []
            x = cand
            for _ in range(40):
                x = np.nextafter(x, _f32(0), dtype=np.float32)
            for _ in range(80):
                if _cell(tmp, T1[b], x) == target:
                    hits.append(x)
                x = np.nextafter(x, _f32(2), dtype=np.float32)
            if not hits:
                continue
            beta = hits[int(rs.randint(0, len(hits)))]
            # the rest of the column: two thirds missing, the others at least 0.3 below the pair
            others = rs.uniform(0, 1, K) < 0.33
            w = np.exp(np.minimum(float(ka) - 0.3 - rs.uniform(0, 4, K) - (float(tmp) + T1.astype(np.float64)), -1e-3))
            w = np.clip(w, 2e-5, 0.999)
            colv = np.where(others, w, 0.0)
            colv[a], colv[b] = alpha, beta
            A[:, col] = hostio.quantize_text16(colv)
            A[a, col], A[b, col] = alpha, beta
            info.append((col, a, b, rel))
            break
        else:
            raise AssertionError("no directed pair found")
    A[0, 0] = 1e-5
    # alpha / beta must survive the text round trip unchanged: they are float32 already, and '%.16f' of a float32 >= 2e-5
    # parses back to the same float32
    assert np.array_equal(hostio.quantize_text16(A.astype(np.float64)), A)
    return A, hostio.quantize_text16(Bm), Pi, T1, info


def _emulated_codes(A, T1, info):
    """What the packed 16-bit filter sees for each designated pair: z = c_T + c_A (fv_kernels.hip.inc, score_code and
    fv_set_model's table codes).  Returns z_a - z_b per column."""
    lmax = max(-math.log(float(x)) for x in A[A > 0])
    step = np.float32(lmax / 65534.0)
    inv = np.float32(1.0) / step
    mx = T1.max()
    out = []
    for col, a, b, rel in info:
        z = []
        for k in (a, b):
            ct = int(np.rint(np.float32(np.float32(mx - T1[k]) * inv)))
            ca = int(np.rint(-math.log(float(A[k, col])) / float(step)))
            z.append(ct + ca)
        out.append(z[0] - z[1])
    return np.array(out)


def test_directed_ties_and_one_ulp_pairs():
    K, C = 640, 240
    A, Bm, Pi, T1, info = directed_model(K, C, 441)
    om = oracle.OracleModel(A, Bm, Pi)
    # the emulation of the cell in this file equals the oracle's (else the pairs are not what they claim to be)
    row, args = om.full_forward(np.array([0, 1 + 0], np.int32), 0, 1)
    col, a, b, rel = info[0]
    assert args[0][col] in (a, b)
    expected, exact_winner = [], []
    for c, (col, a, b, rel) in enumerate(info):
        ob = np.array([0, 1 + c], np.int32)
        opath, oscore, _, orc = om.full_decode(ob, 1)
        assert orc == 0 and opath[1] == col and opath[0] in (a, b), (c, opath, a, b)
        # the pair is what it was built to be: tie -> lower index, else the larger value
        want = min(a, b) if rel == 0 else (b if rel == 1 else a)
        assert opath[0] == want, (c, rel)
        expected.append((ob, opath, oscore))
        exact_winner.append(want == a)
    dz = _emulated_codes(A, T1, info)
    exact_winner = np.array(exact_winner)
    # adversity actually present: in a good share of the columns the 16-bit codes rank the true winner strictly lower
    worse = np.where(exact_winner, dz > 0, dz < 0)
    assert worse.sum() >= C // 8, f"only {worse.sum()} of {C} pairs are ranked the wrong way round by the codes"
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, Bm, Pi)
    try:
        for kernel in FULL_KERNELS:
            fv.set_option(decoder.OPT_KERNEL, kernel)
            for dbg in ((0, 8192 | 16384, 4) if kernel == decoder.KERNEL_U16_REFINE else (0,)):
                fv.set_option(decoder.OPT_DEBUG, dbg)
                for c, (ob, opath, oscore) in enumerate(expected):
                    path, score, rc = fv.decode_full(ob, 1)
                    assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, (kernel, dbg, c, info[c], int(dz[c]))
        # FLASH-BS over the same columns: beam = every state whose T1 is among the B largest; columns whose pair is
        # inside the beam exercise beam_step_q16's window the same way
        Bw = K // 2
        for dbg in (256, 512):
            fv.set_option(decoder.OPT_DEBUG, dbg)
            for c, (ob, _, _) in enumerate(expected[:120]):
                opath, oscore, _, orc = om.beam_decode(ob, 1, Bw)
                path, score, rc = fv.decode_beam(ob, 1, Bw)
                assert path.tolist() == opath.tolist() and score == oscore and rc == orc, (dbg, c, info[c])
    finally:
        fv.close()
        om.close()
