"""Test-side model builders: every golden case is described by a small spec dict
from which the float64 model the generator would have saved is rebuilt, then
pushed through the text quantisation the reference loader applies."""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from flash_viterbi_amd import hostio  # noqa: E402
from flash_viterbi_amd.generate_data import data_script  # noqa: E402


def _ties_all(K, M, seed, degree=8):
    """Every row has `degree` out-edges of weight 1/degree, B and Pi uniform: every finite
    score at a step is the same float, so the decoded path is decided by tie-breaking only."""
    rs = np.random.RandomState(seed)
    A = np.zeros((K, K))
    for k in range(K):
        A[k, rs.choice(K, size=degree, replace=False)] = 1.0 / degree
    return A, np.full((K, M), 1.0 / M), np.full(K, 1.0 / K)


def _ties_semi(K, M, seed):
    """Dyadic weights (row sums exactly 1): many, but not all, candidates tie."""
    rs = np.random.RandomState(seed)
    A = np.zeros((K, K))
    w = np.array([0.25, 0.25, 0.125, 0.125, 0.125, 0.125])
    for k in range(K):
        A[k, rs.choice(K, size=w.size, replace=False)] = rs.permutation(w)
    base = np.array([0.5, 0.25, 0.125, 0.125] + [0.0] * (M - 4)) if M >= 4 else np.full(M, 1.0 / M)
    if M > 4:  # keep every symbol possible: split the last weight
        base = np.array([0.5, 0.25, 0.125] + [0.125 / (M - 3)] * (M - 3))
    B = np.stack([rs.permutation(base) for _ in range(K)])
    return A, B, np.full(K, 1.0 / K)


def _sparse_fast32(K, M, seed, prob, block=256, workers=None):
    """float32 (A, B, Pi) of the generate_data distributions for sizes where the generator's own RNG call
    sequence is too slow to replay in a test (K = 65536: one K-element permutation per row, minutes): every
    entry of A is an edge with probability `prob` (so a row has Binomial(K, prob) out-edges at uniformly
    random places, data_script.py:13-17), weights U(0.01, 1), rows normalised (:19-32); B U(0.1, 1)
    row-normalised (:45-47); Pi = 1/K (:94); every value through the '%.16f' text quantisation the loader
    applies.  Same distributions, NOT the same random stream: the md5s of SURVEY App. C do not apply.
    Row blocks are independent (one PCG64 stream per block), produced on a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    if workers is None:
        try:
            workers = min(16, len(os.sched_getaffinity(0)))
        except AttributeError:
            workers = min(16, os.cpu_count() or 1)
    A = np.empty((K, K), dtype=np.float32)

    def one(r0):
        n = min(block, K - r0)
        g = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, r0])))
        w = g.random((n, K), dtype=np.float32)
        mask = w < np.float32(prob)
        blk = np.where(mask, g.uniform(0.01, 1.0, (n, K)), 0.0)
        empty = ~mask.any(axis=1)
        if empty.any():                                   # a row without edges cannot be normalised
            blk[empty, g.integers(0, K, int(empty.sum()))] = 1.0
        blk /= blk.sum(axis=1)[:, None]
        A[r0:r0 + n] = hostio.quantize_text16(blk)

    with ThreadPoolExecutor(max_workers=workers) as ex:
        list(ex.map(one, range(0, K, block)))
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, K, M])))
    B = g.uniform(0.1, 1.0, (K, M))
    B /= B.sum(axis=1)[:, None]
    return A, hostio.quantize_text16(B), hostio.quantize_text16(np.full(K, 1 / K))


def model64(spec):
    kind = spec["kind"]
    K, M = spec["K"], spec["M"]
    if kind == "data_script":
        return data_script.make_model64(K, M, spec["seed"], spec["prob"])
    if kind == "ties_all":
        return _ties_all(K, M, spec["seed"])
    if kind == "ties_semi":
        return _ties_semi(K, M, spec["seed"])
    raise ValueError(kind)


def observations(spec):
    if "ob" in spec:
        return np.asarray(spec["ob"], dtype=np.int32)
    rng = random.Random(spec.get("ob_seed", spec["seed"]))
    return np.asarray([rng.randint(0, spec["M"] - 1) for _ in range(spec["T"])], dtype=np.int32)


def model32(spec):
    """(A, B, Pi, ob) in the float32 the reference loader reads from the generator's text."""
    if spec["kind"] == "sparse_fast":
        A, B, Pi = _sparse_fast32(spec["K"], spec["M"], spec["seed"], spec["prob"])
        return A, B, Pi, observations(spec)
    A, B, Pi = model64(spec)
    return (hostio.quantize_text16(A), hostio.quantize_text16(B), hostio.quantize_text16(Pi),
            observations(spec))


def write_text(spec, out_dir):
    """Writes the four text files the reference programs open (prob is part of the name)."""
    A, B, Pi = model64(spec)
    data_script.write_files(out_dir, spec["K"], spec["T"], spec["prob"], A, B, Pi,
                            observations(spec), text=True, binary=False)


def sha(a):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]
