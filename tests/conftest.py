import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_goldens(include_big=False):
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*.json"))):
        with open(f) as fh:
            g = json.load(fh)
        if g["spec"]["K"] > 1024 and not include_big:
            continue
        out.append(g)
    return out


def golden_runs(include_big=False, algo=None):
    """Flat list of (golden, run) pairs with readable ids."""
    pairs, ids = [], []
    for g in load_goldens(include_big):
        for r in g["runs"]:
            if algo and r["algo"] != algo:
                continue
            pairs.append((g, r))
            ids.append(f"{g['name']}-{r['algo']}-N{r['N']}" + (f"-B{r['B']}" if "B" in r else "") + (f"-S{r['step']}" if "step" in r else ""))
    return pairs, ids


_model_cache = {}


def golden_model(g):
    """float32 (A, B, Pi, ob) for a golden, rebuilt from its spec and checked against the stored hashes."""
    import modelgen
    key = g["name"]
    if key not in _model_cache:
        A, B, Pi, ob = modelgen.model32(g["spec"])
        assert modelgen.sha(A) == g["sha"]["A"], "generator drift: A differs from the fixture's hash"
        assert modelgen.sha(B) == g["sha"]["B"] and modelgen.sha(Pi) == g["sha"]["Pi"]
        assert [int(x) for x in ob] == g["ob"]
        _model_cache[key] = (A, B, Pi, ob)
    return _model_cache[key]
