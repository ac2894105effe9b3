"""GPU: the single-process multi-device context (fv_create_multi, VERDICT r2 item 3 / SURVEY 8(b)).

One host process, one member per listed device, whole-sequence pass on every member, segments round-robin, one gather.
On the builder's and the driver's 1-GPU boxes the device list repeats id 0: the members then share the GPU and the
gather is device-to-device copies, everything else — host threads, partition, rendezvous, merge — is the path an
8-GPU node takes with ncclCommInitAll.  With two or more visible GPUs the RCCL form is exercised as well."""
import numpy as np
import pytest

import oracle
from conftest import golden_model, golden_runs
from flash_viterbi_amd import decoder

pytestmark = pytest.mark.gpu


def _devices(n):
    vis = decoder.device_count()
    return [i % vis for i in range(n)]


FULL, FULL_IDS = golden_runs(include_big=False, algo="flash")
BEAM, BEAM_IDS = golden_runs(include_big=False, algo="flashbs")


@pytest.fixture(scope="module")
def multi3():
    cache = {}

    def get(g):
        if g["name"] not in cache:
            A, B, Pi, ob = golden_model(g)
            fv = decoder.FlashViterbi([0, 0, 0])
            fv.set_model(A, B, Pi)
            cache[g["name"]] = (fv, ob)
        return cache[g["name"]]
    yield get
    for fv, _ in cache.values():
        fv.close()


@pytest.mark.parametrize("g,r", FULL, ids=FULL_IDS)
def test_three_members_on_one_gpu_decode_the_golden_full(multi3, g, r):
    fv, ob = multi3(g)
    for kernel in (decoder.KERNEL_AUTO, decoder.KERNEL_U16_REFINE, decoder.KERNEL_F64_STREAM):
        fv.set_option(decoder.OPT_KERNEL, kernel)
        path, score, rc = fv.decode_full(ob, r["N"], decoder.MODE_REFERENCE)
        assert rc == 0 and path.tolist() == r["path"] and score == np.float32(r["score"])
    assert fv.stats()["ranks"] == 3


@pytest.mark.parametrize("g,r", BEAM, ids=BEAM_IDS)
def test_three_members_on_one_gpu_decode_the_golden_beam(multi3, g, r):
    fv, ob = multi3(g)
    path, score, rc = fv.decode_beam(ob, r["N"], r["B"], decoder.MODE_REFERENCE)
    assert path.tolist() == r["path"] and score == np.float32(r["score"])
    assert rc == (decoder.WARN_BEAM_MISS if -1 in r["path"] else 0)


@pytest.mark.parametrize("ndev", [2, 5, 8])
def test_members_equal_single_device_on_cfg2_sized_work(ndev):
    """K = 1500, T = 200, n_split = 8: 2 / 5 / 8 members (8 = one segment each, the 8-GPU shape), full-state and beam,
    against the oracle; repeated decodes on one context (buffers, events and the barrier are reused)."""
    import modelgen
    spec = dict(kind="data_script", K=1500, M=20, T=200, prob=0.1, seed=301)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, _ = om.full_decode(ob, 8)
    bpath, bscore, _, brc = om.beam_decode(ob, 8, 128)
    om.close()
    fv = decoder.FlashViterbi(_devices(ndev))
    try:
        fv.set_model(A, Bm, Pi)
        for rep in range(3):
            path, score, rc = fv.decode_full(ob, 8)
            assert rc == 0 and path.tolist() == opath.tolist() and score == oscore, rep
            path, score, rc = fv.decode_beam(ob, 8, 128)
            assert path.tolist() == bpath.tolist() and score == bscore and rc == brc, rep
        assert fv.stats()["ranks"] == ndev
        # baselines run on the first member alone
        vpath, vscore, _ = fv.decode_vanilla(ob)
        assert vpath.tolist() == opath.tolist()
        # argument errors come back as errors from every member, nothing hangs
        with pytest.raises(decoder.FlashVitError):
            fv.decode_full(np.full(200, 99, np.int32), 8)
        with pytest.raises(decoder.FlashVitError):
            fv.decode_beam(ob, 8, 5000)
        path, score, rc = fv.decode_full(ob, 8)
        assert rc == 0 and path.tolist() == opath.tolist()
        with pytest.raises(decoder.FlashVitError):
            fv.set_partition(0, 2)               # a multi-device context has its partition
    finally:
        fv.close()


def test_one_device_list_is_a_plain_context():
    import modelgen
    spec = dict(kind="data_script", K=200, M=9, T=50, prob=0.2, seed=302)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, _ = om.full_decode(ob, 4)
    fv = decoder.FlashViterbi([0])
    fv.set_model(A, Bm, Pi)
    path, score, rc = fv.decode_full(ob, 4)
    assert rc == 0 and path.tolist() == opath.tolist() and fv.stats()["ranks"] == 1
    fv.close()
    with pytest.raises(decoder.FlashVitError):
        decoder.FlashViterbi([0, 99])


@pytest.mark.skipif("decoder.device_count() < 2")
def test_distinct_devices_use_rccl():
    """Only where the box has several GPUs: the ncclCommInitAll / ncclAllGather form."""
    import modelgen
    spec = dict(kind="data_script", K=1500, M=20, T=200, prob=0.1, seed=301)
    A, Bm, Pi, ob = modelgen.model32(spec)
    om = oracle.OracleModel(A, Bm, Pi)
    opath, oscore, _, _ = om.full_decode(ob, 8)
    fv = decoder.FlashViterbi(list(range(min(8, decoder.device_count()))))
    fv.set_model(A, Bm, Pi)
    path, score, rc = fv.decode_full(ob, 8)
    assert rc == 0 and path.tolist() == opath.tolist() and score == oscore
    fv.close()
