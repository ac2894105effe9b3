"""GPU: the C host programs end to end — generate_data text files in, the reference's three stdout
lines out (`time:`, `path: [...]`, `memory:`), patched and compiled exactly the way run_hip.py (and the
reference's run.py) does it.  Paths and memory figures must equal the reference binaries' goldens."""
import importlib.util
import os
import re
import subprocess

import pytest

import modelgen
from conftest import ROOT, load_goldens
from flash_viterbi_amd import build as fvbuild

pytestmark = pytest.mark.gpu


def _run_hip_module():
    spec = importlib.util.spec_from_file_location("run_hip", os.path.join(ROOT, "flash_viterbi_amd", "src", "run_hip.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("name,case,algo", [("FLASH_Viterbi_hip", "ds_K77_M7_T33", "flash"),
                                            ("FLASH_BS_Viterbi_hip", "ds_K77_M7_T33", "flashbs"),
                                            ("FLASH_Viterbi_hip", "cfg1_K128_T256", "flash"),
                                            ("FLASH_BS_Viterbi_hip", "ties_semi_K96_T80", "flashbs")])
def test_host_program_prints_reference_path(tmp_path, name, case, algo):
    run_hip = _run_hip_module()
    g = next(x for x in load_goldens() if x["name"] == case)
    spec = g["spec"]
    data_dir = str(tmp_path) + os.sep
    modelgen.write_text(spec, data_dir)
    src = open(os.path.join(ROOT, "flash_viterbi_amd", "src", name + ".c")).read()
    for r in [x for x in g["runs"] if x["algo"] == algo][:2]:
        p = {"K_STATE": spec["K"], "T_STATE": spec["M"], "obserRouteLEN": spec["T"], "prob": spec["prob"],
             "MAX_THREADS": r["N"], "BeamSearchWidth": r.get("B", 32)}
        run_hip.data_path = data_dir
        text = run_hip.patch_config(src, name, p)
        c = tmp_path / (name + "_modified.c")
        c.write_text(text)
        exe = str(tmp_path / (name + "_modified"))
        res = subprocess.run(fvbuild.program_cc(str(c), exe), capture_output=True, text=True)
        assert res.returncode == 0, res.stderr
        # text parse, cache write, cache read; then the same program as the one host process of two and of three devices
        # (FV_NGPUS: on a 1-GPU box the members share the GPU and merge by device-to-device copies)
        for cache, ngpus in (("0", 1), ("1", 1), ("1", 1), ("1", 2), ("1", 3)):
            out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, FV_BIN_CACHE=cache, FV_NGPUS=str(ngpus)))
            assert out.returncode == 0, out.stderr
            assert int(re.search(r"n_gpus: (\d+)", out.stderr).group(1)) == ngpus
            assert re.search(r"time: ([\d.]+)", out.stdout)                       # reference run.py:75
            assert int(re.search(r"memory: (\d+)", out.stdout).group(1)) == r["memory"]   # run.py:76
            path = [int(x) for x in re.search(r"path: \[([^\]]*)\]", out.stdout).group(1).split()]
            assert path == r["path"]
            assert float(re.search(r"score: (\S+)", out.stderr).group(1)) == pytest.approx(r["score"], rel=1e-7)


def test_run_hip_main_writes_reference_schema_plus_gpu_columns(tmp_path, monkeypatch):
    """run_hip.main() end to end on a small parameter set: the CSV keeps the reference's columns first
    (src/run.py:105) and adds n_gpus, cells/s, roofline fraction, the path hash and the path-equality flag
    (SURVEY 8 f-2), the flag filled from a --ref-md5 file holding the golden path of the reference binary."""
    import csv
    import json
    import sys
    run_hip = _run_hip_module()
    g = next(x for x in load_goldens() if x["name"] == "cfg1_K128_T256")
    spec = g["spec"]
    rf = next(x for x in g["runs"] if x["algo"] == "flash" and x["N"] == 8)
    rb = next(x for x in g["runs"] if x["algo"] == "flashbs" and x["N"] == 8 and x["B"] == 32)
    data_dir, res_dir, src_dir = str(tmp_path / "data") + os.sep, str(tmp_path / "result") + os.sep, str(tmp_path / "src") + os.sep
    os.makedirs(src_dir)
    for name in run_hip.file_names:
        with open(os.path.join(ROOT, "flash_viterbi_amd", "src", name + ".c")) as f:
            open(src_dir + name + ".c", "w").write(f.read())
    modelgen.write_text(spec, data_dir)
    p = {"K_STATE": spec["K"], "T_STATE": spec["M"], "obserRouteLEN": spec["T"], "prob": spec["prob"], "MAX_THREADS": 8, "BeamSearchWidth": 32}
    monkeypatch.setattr(run_hip, "data_path", data_dir)
    monkeypatch.setattr(run_hip, "result_path", res_dir)
    monkeypatch.setattr(run_hip, "base_path", src_dir)
    monkeypatch.setattr(run_hip, "parameters", [p])
    ref = {run_hip.ref_key("FLASH_Viterbi_hip", p): run_hip.path_md5(rf["path"]),
           run_hip.ref_key("FLASH_BS_Viterbi_hip", p): run_hip.path_md5(rb["path"])}
    ref_file = tmp_path / "ref.json"
    ref_file.write_text(json.dumps(ref))
    monkeypatch.setattr(sys, "argv", ["run_hip.py", "--ref-md5", str(ref_file)])
    run_hip.main()
    run_hip.main()                       # second run: appends, reads the bound .f32 caches
    for name, r in (("FLASH_Viterbi_hip", rf), ("FLASH_BS_Viterbi_hip", rb)):
        rows = list(csv.reader(open(res_dir + name + "_result.csv")))
        assert rows[0][:9] == ["timestamp", "K_STATE", "T_STATE", "obserRouteLEN", "prob", "MAX_THREADS", "BeamSearchWidth", "time", "memory"]
        assert len(rows) == 3
        for row in rows[1:]:
            d = dict(zip(rows[0], row))
            assert int(d["memory"]) == r["memory"] and d["n_gpus"] == "1" and d["ref_path_equal"] == "True"
            assert float(d["cells_per_s"]) > 0 and 0.0 < float(d["roofline_frac"]) < 1.5
            assert d["path_md5"] == run_hip.path_md5(r["path"])


def test_host_program_reports_missing_input(tmp_path):
    run_hip = _run_hip_module()
    src = open(os.path.join(ROOT, "flash_viterbi_amd", "src", "FLASH_Viterbi_hip.c")).read()
    run_hip.data_path = str(tmp_path) + os.sep
    p = {"K_STATE": 8, "T_STATE": 3, "obserRouteLEN": 5, "prob": 0.5, "MAX_THREADS": 1, "BeamSearchWidth": 4}
    c = tmp_path / "x.c"
    c.write_text(run_hip.patch_config(src, "FLASH_Viterbi_hip", p))
    exe = str(tmp_path / "x")
    assert subprocess.run(fvbuild.program_cc(str(c), exe), capture_output=True).returncode == 0
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "cannot open" in out.stderr
