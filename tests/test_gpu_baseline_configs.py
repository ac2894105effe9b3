"""GPU: BASELINE.json's large configurations at FULL size against the oracle, through the C ABI.

  configs[2]  K=3965  T=4096           N=8   full-state FLASH      (tests/test_gpu_properties.py has the
                                                                    size-independent properties of the same case)
  configs[3]  K=16384 T=256  B=256     N=8   FLASH-BS
  configs[4]  K=65536 T=1024 B=1024    N=8   FLASH-BS

Bar: decoded path bit-exact, final score float32-equal, return code equal (a beam miss is part of the
result).  The oracle (oracle/flashvit_oracle.c) is pinned to the reference's own binaries by
tests/test_oracle_golden.py; at these sizes the reference programs themselves would need hours
(single-threaded whole-sequence pass, one libm log() per cell), so the oracle is what runs here.
Both beam step kernels are forced in turn (FV_OPT_DEBUG 256: float64 rows, 512: 16-bit filter + refine),
then the library's own choice.

cfg3 / cfg4 use the generate_data random stream (seed 12, as bench.py); cfg5 uses modelgen's
"sparse_fast" builder — the same distributions from a vectorised random stream, because replaying
generate_data's per-row K-element permutations takes minutes at K = 65536 (data_script.make_model32_fast).
"""
import time

import numpy as np
import pytest

import modelgen
import oracle
from flash_viterbi_amd import decoder

pytestmark = pytest.mark.gpu


def _log(msg):
    print(f"[baseline-configs] {msg}", flush=True)


def _beam_case(spec, n_split, beam, partition_check=False, expect_cuts=False):
    t0 = time.time()
    A, Bm, Pi, ob = modelgen.model32(spec)
    _log(f"K={spec['K']} model built in {time.time() - t0:.1f}s")
    fv = decoder.FlashViterbi(0)
    try:
        t0 = time.time()
        fv.set_model(A, Bm, Pi)
        _log(f"fv_set_model {time.time() - t0:.1f}s")
        got = {}
        for dbg in (256, 512, 1 << 20, 1 << 23, 0):  # float64 step kernel, 16-bit filter, eager replays (round-2 path), undecided runs always decided in full, default
            fv.set_option(decoder.OPT_DEBUG, dbg)
            path, score, rc = fv.decode_beam(ob, n_split, beam, decoder.MODE_REFERENCE)
            st = fv.stats()
            got[dbg] = (path, score, rc)
            # the shortcut of the resolve code (a run of undecided steps decided only back to a step whose replay provably
            # does not depend on them) must have been taken where it is expected, and never when it is switched off
            assert st["beam_chain_cuts"] == 0 or not (dbg & ((1 << 23) | (1 << 20))), dbg
            assert not (expect_cuts and dbg == 0) or st["beam_chain_cuts"] > 0
            _log(f"FV_OPT_DEBUG={dbg}: gpu_ms {st['gpu_ms']:.2f} top_ms {st['top_pass_ms']:.2f} exact replays {st['beam_exact_sets']} speculative steps "
                 f"{st['beam_spec_steps']} reach events {st['beam_reach_events']} chain cuts {st['beam_chain_cuts']} dup cols {st['beam_dup_cols']} rc {rc}")
        fv.set_option(decoder.OPT_DEBUG, 0)
        if partition_check:
            # a 1-GPU run with n_split = 8 must equal the 8-rank run; every simulated rank decodes
            # its share on this GPU and the product's merge puts the slices together
            slices = []
            for r in range(8):
                fv.set_partition(r, 8)
                slices.append(fv.decode_beam(ob, n_split, beam)[0])
            fv.set_partition(0, 1)
            merged = decoder.merge_paths(ob.size, n_split, 8, np.stack(slices))
            assert merged.tolist() == got[0][0].tolist(), "8-rank partition differs from the 1-rank decode"
    finally:
        fv.close()
    t0 = time.time()
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.beam_decode(ob, n_split, beam)
    om.close()
    _log(f"oracle {time.time() - t0:.1f}s score {oscore} rc {orc}")
    for dbg, (path, score, rc) in got.items():
        assert path.tolist() == opath.tolist(), f"FV_OPT_DEBUG={dbg}: path differs from the oracle"
        assert score == oscore and rc == orc, f"FV_OPT_DEBUG={dbg}: score/rc {score}/{rc} vs {oscore}/{orc}"


def test_cfg3_full_decode_equals_oracle():
    """BASELINE configs[2] on one GPU: K=3965, T=4096, n_split=8, FV_MODE_REFERENCE, dense and sparse kernels."""
    spec = dict(kind="data_script", K=3965, M=50, T=4096, prob=0.112, seed=12)
    A, Bm, Pi, ob = modelgen.model32(spec)
    fv = decoder.FlashViterbi(0)
    try:
        fv.set_model(A, Bm, Pi)
        got = {}
        for kernel in (decoder.KERNEL_Q16_REFINE, decoder.KERNEL_U16_REFINE, decoder.KERNEL_AUTO):      # U16: what bench.py --workload cfg3 times
            fv.set_option(decoder.OPT_KERNEL, kernel)
            got[kernel] = fv.decode_full(ob, 8, decoder.MODE_REFERENCE)
            _log(f"cfg3 kernel {fv.stats()['kernel']}: gpu_ms {fv.stats()['gpu_ms']:.1f}")
    finally:
        fv.close()
    t0 = time.time()
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.full_decode(ob, 8)
    om.close()
    _log(f"cfg3 oracle {time.time() - t0:.1f}s")
    assert orc == 0
    for kernel, (path, score, rc) in got.items():
        assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, kernel


def test_full_state_decode_beyond_the_f32_lds_limit_equals_oracle():
    """one float32 score row of K = 44000 states no longer fits LDS (the f32 kernels stop at
    K ~ 40100); a row of 16-bit score codes does, so the packed 16-bit kernel takes over (its table is built on the
    device on first use).  The reference sizes everything from K_STATE and has no such ceiling
    (src/FLASH_Viterbi_multithread.c:25-34)."""
    spec = dict(kind="sparse_fast", K=44000, M=20, T=8, prob=0.02, seed=31)
    A, Bm, Pi, ob = modelgen.model32(spec)
    fv = decoder.FlashViterbi(0)
    try:
        fv.set_model(A, Bm, Pi)
        path, score, rc = fv.decode_full(ob, 1, decoder.MODE_REFERENCE)
        st = fv.stats()
        assert st["kernel"] == decoder.KERNEL_U16_REFINE
        _log(f"K=44000 full decode gpu_ms {st['gpu_ms']:.1f} passes {st['passes']}")
        path3, score3, rc3 = fv.decode_full(ob, 3, decoder.MODE_REFERENCE)
        # the float64 kernel takes such a model in slabs of source rows (two launches per step here), and so does the f32
        # filter on the 16-bit table
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_F64_STREAM)
        path64, score64, rc64 = fv.decode_full(ob, 3, decoder.MODE_REFERENCE)
        assert fv.stats()["kernel"] == decoder.KERNEL_F64_STREAM
        _log(f"K=44000 float64 kernel in slabs gpu_ms {fv.stats()['gpu_ms']:.1f}")
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_Q16_REFINE)
        pathq, scoreq, rcq = fv.decode_full(ob, 3, decoder.MODE_REFERENCE)
        assert fv.stats()["kernel"] == decoder.KERNEL_Q16_REFINE
        _log(f"K=44000 f32 filter on the 16-bit table in slabs gpu_ms {fv.stats()['gpu_ms']:.1f}")
        assert rcq == rc64 and pathq.tolist() == path64.tolist() and scoreq == score64
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_F32_REFINE)
        with pytest.raises(decoder.FlashVitError):
            fv.decode_full(ob, 1)                          # the kernels that need a float32 row in LDS still say UNSUPPORTED
    finally:
        fv.close()
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, orc = om.full_decode(ob, 1)
    opath3, oscore3, _, _ = om.full_decode(ob, 3)
    om.close()
    assert rc == 0 and orc == 0 and path.tolist() == opath.tolist() and score == oscore
    assert rc3 == 0 and path3.tolist() == opath3.tolist() and score3 == oscore3
    assert rc64 == 0 and path64.tolist() == opath3.tolist() and score64 == oscore3


def test_more_than_65536_states_equals_oracle():
    """VERDICT r2, missing item 5: the reference sizes everything from K_STATE (src/FLASH_Viterbi_multithread.c:25-34,
    src/FLASH_BS_Viterbi_multithread.c:27-36); K = 69632 is beyond a row of 16-bit score codes in LDS and beyond the 64
    rounds of the register selects.  Full-state: trellis_step in slabs of source rows (two per step, eight when four tasks share
    a launch) — the f32 filter on the 16-bit table (AUTO) and the float64 kernel.  FLASH-BS: both step kernels; selections on the candidate lists, and over the K scores in memory
    where there is none (the first two steps of every pass; every step with FV_OPT_DEBUG bit 10)."""
    spec = dict(kind="sparse_fast", K=69632, M=20, T=24, prob=0.03, seed=47)
    t0 = time.time()
    A, Bm, Pi, ob = modelgen.model32(spec)
    _log(f"K=69632 model built in {time.time() - t0:.1f}s")
    fv = decoder.FlashViterbi(0)
    try:
        t0 = time.time()
        fv.set_model(A, Bm, Pi)
        _log(f"fv_set_model {time.time() - t0:.1f}s")
        full = fv.decode_full(ob[:7], 2, decoder.MODE_REFERENCE)
        st = fv.stats()
        assert st["kernel"] == decoder.KERNEL_Q16_REFINE          # AUTO: the f32 filter on the 16-bit table, in slabs
        _log(f"K=69632 full decode T=7 N=2: gpu_ms {st['gpu_ms']:.1f} passes {st['passes']} step launches {st['step_launches']}")
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_F64_STREAM)
        full64 = fv.decode_full(ob[:7], 2, decoder.MODE_REFERENCE)
        _log(f"K=69632 float64 kernel: gpu_ms {fv.stats()['gpu_ms']:.1f}")
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
        assert full64[2] == full[2] and full64[0].tolist() == full[0].tolist() and full64[1] == full[1]
        beam = {}
        for dbg in (0, 256, 512, 1024, 1 << 20):
            fv.set_option(decoder.OPT_DEBUG, dbg)
            beam[dbg] = fv.decode_beam(ob, 3, 512, decoder.MODE_REFERENCE)
            st = fv.stats()
            _log(f"K=69632 beam B=512 FV_OPT_DEBUG={dbg}: gpu_ms {st['gpu_ms']:.2f} selects on a list {st['beam_cand_selects']} exact replays {st['beam_exact_sets']}")
        fv.set_option(decoder.OPT_DEBUG, 0)
    finally:
        fv.close()
    t0 = time.time()
    om = oracle.OracleModel(A, Bm, Pi)
    bpath, bscore, _, brc = om.beam_decode(ob, 3, 512)
    _log(f"oracle beam {time.time() - t0:.1f}s")
    t0 = time.time()
    opath, oscore, _, orc = om.full_decode(ob[:7], 2)
    om.close()
    _log(f"oracle full {time.time() - t0:.1f}s")
    assert orc == 0 and full[2] == 0 and full[0].tolist() == opath.tolist() and full[1] == oscore
    for dbg, (path, score, rc) in beam.items():
        assert path.tolist() == bpath.tolist() and score == bscore and rc == brc, dbg


def test_cfg4_beam_decode_equals_oracle():
    """BASELINE configs[3]: K=16384, T=256, B=256, n_split=8 (127 passes, batched launches, in-kernel replays)."""
    _beam_case(dict(kind="data_script", K=16384, M=50, T=256, prob=0.112, seed=12), 8, 256, partition_check=True)


def test_cfg5_beam_decode_equals_oracle():
    """BASELINE configs[4]: K=65536, T=1024, B=1024, n_split=8 (the select's 64-round instantiation at its K
    limit, ~1500 exact replays, 34 GB float64 row table), plus the 8-rank partition of SURVEY 8(e)."""
    _beam_case(dict(kind="sparse_fast", K=65536, M=50, T=1024, prob=0.112, seed=12), 8, 1024, partition_check=True, expect_cuts=True)
