"""CPU: the drop-in boundary — libraries load without a GPU and export every symbol their
headers declare, the host I/O reproduces the generate_data text format byte for byte, the
product never touches the oracle, and the host programs accept the reference's run.py patching."""
import ctypes
import io
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, golden_model, load_goldens
from flash_viterbi_amd import build as fvbuild
from flash_viterbi_amd import decoder, hostio

INCLUDE = os.path.join(ROOT, "include")


def declared_functions(header):
    text = open(os.path.join(INCLUDE, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fvh?_[a-z0-9_]+)\s*\(", text)))


def test_hip_library_exports_every_declared_symbol():
    lib = decoder.load_library()          # dlopen works without a GPU
    names = declared_functions("flashvit.h")
    assert "fv_decode_full" in names and "fv_decode_beam" in names and "fv_comm_init" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/flashvit.h but not exported"
    assert sorted(decoder.EXPORTS) == [n for n in names]


def test_host_library_exports_every_declared_symbol():
    lib = hostio.lib()
    for n in declared_functions("flashvit_host.h"):
        assert hasattr(lib, n), n


def test_no_gpu_means_loud_failure_not_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(decoder.FlashVitError):
        decoder.FlashViterbi(0)


def test_product_never_references_the_oracle():
    """Nothing under flash_viterbi_amd/ may import, load, run or even locate the checker: no `oracle` module or
    directory, no oracle/build_ref.py, no oracle/_ref binaries, no fvo_ symbols.  The one allowed mention is the
    sentence in decoder.py's docstring saying exactly that."""
    pkg = os.path.join(ROOT, "flash_viterbi_amd")
    allowed = {("decoder.py", "visible, construction fails loudly.  (The CPU restatement lives in oracle/ and is test")}
    bad = re.compile(r"oracle|build_ref|_ref\b|_ref/|libfvoracle|fvo_", re.I)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".cpp", ".h", ".hip", ".inc", ".sh", ".json", ".md")):
                for line in open(os.path.join(dirpath, f)).read().splitlines():
                    if bad.search(line) and (f, line.strip()) not in allowed:
                        pytest.fail(f"{f}: product code refers to the checker: {line.strip()}")
    # and neither shared library links it
    for lib in (fvbuild.HIP_LIB, fvbuild.HOST_LIB):
        out = subprocess.run(["ldd", lib], capture_output=True, text=True).stdout
        assert "fvoracle" not in out


def test_text_writer_matches_numpy_savetxt_bytes(tmp_path):
    rs = np.random.RandomState(0)
    a = rs.uniform(0, 1, (7, 5)) * (rs.uniform(0, 1, (7, 5)) < 0.5)
    a[0, 0] = 1e-9
    a[1, 1] = 1.0
    p = str(tmp_path / "m.txt")
    hostio.write_matrix_text16(p, a)
    ref = io.BytesIO()
    np.savetxt(ref, a, fmt="%.16f")                   # reference generate_data/data_script.py:98
    assert open(p, "rb").read() == ref.getvalue()
    v = rs.uniform(0, 1, 9)
    hostio.write_vector_text16(p, v)
    ref = io.BytesIO()
    np.savetxt(ref, v, fmt="%.16f", newline=" ")      # data_script.py:100
    assert open(p, "rb").read() == ref.getvalue()
    ob = rs.randint(0, 50, 11)
    hostio.write_ints_text(p, ob)
    ref = io.BytesIO()
    np.savetxt(ref, ob, fmt="%d", newline=" ")        # data_script.py:101
    assert open(p, "rb").read() == ref.getvalue()
    assert hostio.read_ints_text(p, 11).tolist() == ob.tolist()
    with pytest.raises(IOError):
        hostio.read_ints_text(p, 12)                  # short file is an error, not garbage


def test_quantize_equals_text_round_trip_and_bin_cache(tmp_path):
    rs = np.random.RandomState(1)
    a = rs.uniform(0, 1, (6, 4)) ** 8
    p = str(tmp_path / "a.txt")
    hostio.write_matrix_text16(p, a)
    back = hostio.read_floats_text(p, (6, 4))
    assert back.dtype == np.float32 and (back == hostio.quantize_text16(a)).all()
    b = str(tmp_path / "a.f32")
    hostio.write_bin_f32(b, back)
    assert (hostio.read_bin_f32(b, 6, 4) == back).all()
    with pytest.raises(IOError):
        hostio.read_bin_f32(b, 4, 6)                  # shape is part of the header


def test_bin_cache_is_bound_to_its_text_file(tmp_path):
    """regenerated inputs keep their file names, so a raw cache must not outlive the text it was
    parsed from.  A bound cache is taken only while the text's size and mtime are those recorded; an unbound
    one (generator --bin) only when there is no text at all."""
    rs = np.random.RandomState(2)
    a = rs.uniform(0, 1, (5, 3))
    txt, binf = str(tmp_path / "A_K5_T9_prob0.5.txt"), str(tmp_path / "A_K5_T9_prob0.5.f32")
    hostio.write_matrix_text16(txt, a)
    a32 = hostio.read_floats_text(txt, (5, 3))
    hostio.write_bin_f32_src(binf, a32, txt)
    assert (hostio.read_bin_f32_src(binf, 5, 3, txt) == a32).all()
    assert (hostio.read_bin_f32(binf, 5, 3) == a32).all()            # plain reader accepts both header versions
    # same name, other content (another seed): same size, later mtime
    hostio.write_matrix_text16(txt, rs.uniform(0, 1, (5, 3)))
    st = os.stat(txt)
    os.utime(txt, ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000_000))
    with pytest.raises(IOError, match="does not belong"):
        hostio.read_bin_f32_src(binf, 5, 3, txt)
    # unbound cache next to a text file: refused; without the text: accepted
    hostio.write_bin_f32(binf, a32)
    with pytest.raises(IOError, match="does not belong"):
        hostio.read_bin_f32_src(binf, 5, 3, txt)
    os.remove(txt)
    assert (hostio.read_bin_f32_src(binf, 5, 3, txt) == a32).all()


def test_generator_reproduces_fixture_hashes():
    g = load_goldens()[0]
    golden_model(g)        # asserts sha(A), sha(B), sha(Pi), ob against the fixture


def test_memory_formula_matches_reference_printout():
    for g in load_goldens(include_big=True):
        K, T = g["spec"]["K"], g["spec"]["T"]
        for r in g["runs"]:
            if r["algo"] == "checkpoint":
                assert decoder.checkpoint_memory_bytes(K, T, r["step"]) == r["memory"]
            elif r["algo"] != "vanilla":
                assert decoder.reference_memory_bytes(K, T, r["N"], r.get("B", 0)) == r["memory"]


@pytest.mark.parametrize("name", ["FLASH_Viterbi_hip", "FLASH_BS_Viterbi_hip"])
def test_host_program_takes_reference_runpy_patching(name, tmp_path):
    """Apply the substitutions of the reference's run.py:29-47 to our source and compile it."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("run_hip", os.path.join(ROOT, "flash_viterbi_amd", "src", "run_hip.py"))
    run_hip = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(run_hip)
    src = open(os.path.join(ROOT, "flash_viterbi_amd", "src", name + ".c")).read()
    p = {"K_STATE": 77, "T_STATE": 7, "obserRouteLEN": 33, "prob": 0.3, "MAX_THREADS": 3, "BeamSearchWidth": 9}
    out = run_hip.patch_config(src, name, p)
    assert "#define K_STATE 77" in out and "#define T_STATE 7\n" in out and "#define obserRouteLEN 33" in out
    assert "const float prob = 0.3;" in out and "#define MAX_THREADS 3" in out and "prob%.1f" in out
    if "BS" in name:
        assert "const int BeamSearchWidth = 9;" in out
    c = tmp_path / (name + "_modified.c")
    c.write_text(out)
    cmd = ["gcc", "-c", "-Wall", "-Werror", "-I", INCLUDE, str(c), "-o", str(tmp_path / "x.o")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr


def test_streaming_generator_is_byte_identical_to_dense(tmp_path):
    from flash_viterbi_amd.generate_data import data_script
    K, M, T, prob, seed = 150, 9, 20, 0.2, 4
    ob = data_script.make_observations(T, M, seed)
    A, B, Pi = data_script.make_model64(K, M, seed, prob)
    d1, d2 = str(tmp_path / "dense"), str(tmp_path / "stream")
    data_script.write_files(d1, K, T, prob, A, B, Pi, ob, text=True, binary=True)
    data_script.write_files_streaming(d2, K, M, T, prob, seed, ob, text=True, binary=True, block_rows=37)
    names = sorted(os.listdir(d1))
    assert names == sorted(os.listdir(d2)) and len(names) == 8
    for n in names:
        assert open(os.path.join(d1, n), "rb").read() == open(os.path.join(d2, n), "rb").read(), n
    back = hostio.read_bin_f32(os.path.join(d2, f"A_K{K}_T{T}_prob{prob}.f32"), K, K)
    assert (back == hostio.quantize_text16(A)).all()
