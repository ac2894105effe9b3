#!/usr/bin/env python3
"""Synthetic HMM generator: the build's counterpart of the reference's
generate_data/data_script.py (same CLI flags, same file names, same text format,
same distributions and the same numpy RNG call sequence, so the same seed gives
byte-identical A_/B_/Pi_ files).

Reference behaviour followed (file:line under /root/reference/generate_data/):
  * transition matrix: data_script.py:5-35 — per source state, Binomial(K, p)
    out-edges drawn without replacement, weights U(0.01, 1), row-normalised,
    everything else exactly 0;
  * emission matrix: data_script.py:38-49 — U(0.1, 1), row-normalised, the global
    numpy RNG re-seeded with the same seed first;
  * Pi = 1/K (:94); files and formats :98-101.

Deliberate differences:
  * the observation sequence is reproducible: the reference draws it from an
    unseeded `random` (:86); here `random` is seeded with the -s seed (as the
    reference's own data_script_dag.py:46 does) unless --ob-seed says otherwise;
  * text is written by libfvhost (C) instead of np.savetxt — byte-identical
    output, ~20x faster at K=3965 (a 299 MB file);
  * --bin also writes raw float32 caches (`*.f32`, `ob_*.i32`) holding exactly
    the floats the reference's fscanf("%f") loader would obtain from the text;
  * -o chooses the output directory (the reference writes to the cwd);
  * --stream (automatic for K >= 16384) produces and writes A in row blocks, byte-identical to the dense
    route, so K = 65536 (34 GB as float64, ~82 GB as text) never has to fit in memory.
"""
import getopt
import os
import random
import sys

import numpy as np

USAGE = ("data_script.py -s <seed> -n <n_ob> -K <K> -T <T> -b <beam_width> -p <prob> "
         "[-o <out_dir>] [--ob-seed <seed>] [--bin] [--no-text] [--stream]")


def make_transition(K, seed, prob):
    """float64 K x K row-stochastic sparse-random transition matrix (data_script.py:5-35)."""
    np.random.seed(seed)
    A = np.zeros((K, K), dtype=np.float64)
    for src in range(K):
        n_edges = np.random.binomial(K, p=prob, size=None)
        dst = np.random.choice(K, size=n_edges, replace=False)
        w = np.random.uniform(0.01, 1, size=n_edges)
        A[src, dst] = w
    for src in range(K):
        A[src, :] = A[src, :] / np.sum(A[src, :])
    return A


def stream_transition(K, seed, prob, block_rows=512):
    """Yields (first_row, float64 block) of the same matrix make_transition returns, block_rows rows at a
    time.  The reference draws every row's edges first and normalises afterwards (data_script.py:11-32),
    but only the draws consume the RNG and the normalisation is row-local, so producing row by row gives
    the same numbers; the row sum is still taken over the dense K-entry row (np.sum's pairwise order)."""
    np.random.seed(seed)
    for r0 in range(0, K, block_rows):
        n = min(block_rows, K - r0)
        blk = np.zeros((n, K), dtype=np.float64)
        for q in range(n):
            n_edges = np.random.binomial(K, p=prob, size=None)
            dst = np.random.choice(K, size=n_edges, replace=False)
            blk[q, dst] = np.random.uniform(0.01, 1, size=n_edges)
        for q in range(n):
            blk[q, :] = blk[q, :] / np.sum(blk[q, :])
        yield r0, blk


def write_files_streaming(out_dir, K, n_ob, T, prob, seed, ob, text=True, binary=True, block_rows=512):
    """write_files for models too large to hold as float64 (K = 65536: 34 GB): A is produced and written
    in row blocks; B, Pi, ob are small."""
    from flash_viterbi_amd import hostio
    os.makedirs(out_dir, exist_ok=True)
    p = lambda kind, ext: os.path.join(out_dir, file_stem(kind, K, T, prob) + ext)
    fbin = None
    if binary:
        fbin = open(p("A", ".f32"), "wb")
        fbin.write(hostio.bin_header(1, K, K))
    for r0, blk in stream_transition(K, seed, prob, block_rows):
        if text:
            hostio.append_matrix_text16(p("A", ".txt"), blk, first=(r0 == 0))
        if fbin:
            fbin.write(hostio.quantize_text16(blk).tobytes())
    if fbin:
        fbin.close()
    B = make_emission(K, n_ob, seed)
    Pi = np.full(K, 1 / K)
    if text:
        hostio.write_matrix_text16(p("B", ".txt"), B)
        hostio.write_vector_text16(p("Pi", ".txt"), Pi)
        hostio.write_ints_text(p("ob", ".txt"), ob)
    if binary:
        hostio.write_bin_f32(p("B", ".f32"), hostio.quantize_text16(B))
        hostio.write_bin_f32(p("Pi", ".f32"), hostio.quantize_text16(Pi).reshape(1, -1))
        hostio.write_bin_i32(p("ob", ".i32"), np.asarray(ob, dtype=np.int32).reshape(1, -1))


def make_emission(K, n_ob, seed):
    """float64 K x n_ob row-stochastic emission matrix (data_script.py:38-49)."""
    np.random.seed(seed)
    B = np.random.uniform(0.1, 1, (K, n_ob))
    return B / B.sum(axis=1)[:, None]


def make_observations(T, n_ob, seed):
    rng = random.Random(seed)
    return [rng.randint(0, n_ob - 1) for _ in range(T)]


def make_model64(K, n_ob, seed, prob):
    """(A, B, Pi) in float64, exactly the arrays the reference generator saves."""
    return make_transition(K, seed, prob), make_emission(K, n_ob, seed), np.full(K, 1 / K)


def make_model32_fast(K, n_ob, seed, prob, block=256, workers=None):
    """float32 (A, B, Pi) of the generate_data distributions for sizes where the generator's own RNG call
    sequence is too slow to replay (K = 65536: one K-element permutation per row, minutes): every
    entry of A is an edge with probability `prob` (so a row has Binomial(K, prob) out-edges at uniformly
    random places, data_script.py:13-17), weights U(0.01, 1), rows normalised (:19-32); B U(0.1, 1)
    row-normalised (:45-47); Pi = 1/K (:94); every value through the '%.16f' text quantisation the loader
    applies.  Same distributions, NOT the same random stream: the md5s of SURVEY App. C do not apply.
    Row blocks are independent (one PCG64 stream per block), produced on a thread pool."""
    from concurrent.futures import ThreadPoolExecutor
    from flash_viterbi_amd import hostio
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
    g = np.random.Generator(np.random.PCG64(np.random.SeedSequence([seed, K, n_ob])))
    B = g.uniform(0.1, 1.0, (K, n_ob))
    B /= B.sum(axis=1)[:, None]
    return A, hostio.quantize_text16(B), hostio.quantize_text16(np.full(K, 1 / K))


def file_stem(kind, K, T, prob):
    return f"{kind}_K{K}_T{T}_prob{prob}"


def write_files(out_dir, K, T, prob, A, B, Pi, ob, text=True, binary=False):
    from flash_viterbi_amd import hostio
    os.makedirs(out_dir, exist_ok=True)
    p = lambda kind, ext: os.path.join(out_dir, file_stem(kind, K, T, prob) + ext)
    if text:
        hostio.write_matrix_text16(p("A", ".txt"), A)
        hostio.write_matrix_text16(p("B", ".txt"), B)
        hostio.write_vector_text16(p("Pi", ".txt"), Pi)
        hostio.write_ints_text(p("ob", ".txt"), ob)
    if binary:
        hostio.write_bin_f32(p("A", ".f32"), hostio.quantize_text16(A))
        hostio.write_bin_f32(p("B", ".f32"), hostio.quantize_text16(B))
        hostio.write_bin_f32(p("Pi", ".f32"), hostio.quantize_text16(Pi).reshape(1, -1))
        hostio.write_bin_i32(p("ob", ".i32"), np.asarray(ob, dtype=np.int32).reshape(1, -1))


def main(argv):
    try:
        opts, _ = getopt.getopt(argv, "hs:n:K:T:b:p:o:", ["ob-seed=", "bin", "no-text", "stream"])
    except getopt.GetoptError:
        print(USAGE)
        sys.exit(2)
    got = {}
    out_dir, ob_seed, binary, text, stream = ".", None, False, True, False
    for opt, arg in opts:
        if opt == "-h":
            print(USAGE)
            sys.exit()
        elif opt in ("-s", "-n", "-K", "-T", "-b"):
            got[opt] = int(arg)
        elif opt == "-p":
            got[opt] = float(arg)
        elif opt == "-o":
            out_dir = arg
        elif opt == "--ob-seed":
            ob_seed = int(arg)
        elif opt == "--bin":
            binary = True
        elif opt == "--no-text":
            text = False
        elif opt == "--stream":
            stream = True
    if len(got) != 6:
        print(USAGE)
        sys.exit(2)
    sd, n_ob, K, T, beam, prob = got["-s"], got["-n"], got["-K"], got["-T"], got["-b"], got["-p"]
    os.makedirs(out_dir, exist_ok=True)
    # the reference leaves a one-line stub behind (data_script.py:83-84); keep it so a
    # directory produced here lists the same files
    with open(os.path.join(out_dir, f"ANS_K{K}_T{T}_prob{prob}_beam_width{beam}.txt"), "w") as f:
        f.write(f"sd={sd}, n_ob={n_ob}, K={K}, T={T}, beam_width={beam}, prob={prob}\n")
    ob = make_observations(T, n_ob, sd if ob_seed is None else ob_seed)
    if stream or K >= 16384:
        write_files_streaming(out_dir, K, n_ob, T, prob, sd, ob, text=text, binary=binary)
        return
    A, B, Pi = make_model64(K, n_ob, sd, prob)
    write_files(out_dir, K, T, prob, A, B, Pi, ob, text=text, binary=binary)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    main(sys.argv[1:])
