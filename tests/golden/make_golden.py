#!/usr/bin/env python3
"""Regenerates tests/golden/*.json by running binaries compiled from the
reference's own sources (oracle/build_ref.py) on inputs written in the
generate_data text format.  Needs /root/reference; run it in the build
container, never on the GPU box.

A fixture is data only: the model spec (seed / kind / sizes), the observation
sequence, hashes of the float32 arrays the loader sees, and for each run the
reference program's printed path, `memory:` figure and (from a spliced-in
fprintf, see build_ref.py) the whole-sequence final score.

  python3 tests/golden/make_golden.py            # all small cases
  python3 tests/golden/make_golden.py --big      # also K=3965 (minutes of CPU)
  python3 tests/golden/make_golden.py --append   # keep the recorded runs, execute only new ones
"""
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import build_ref  # noqa: E402
import modelgen  # noqa: E402

# the observation sequences the survey's golden paths were taken on
OB_CFG1 = [int(x) for x in """10 13 46 4 32 49 40 7 42 14 30 44 30 35 17 18 15 24 35 6 37 4 18 40 1 40 17 11 33 35 27 35 40 30 27 10 31 40 43 16 16 9 33 15 12 11 15 21 26 46 30 42 0 45 29 22 9 9 41 25 24 46 16 34 0 24 36 25 7 12 40 39 35 45 9 6 34 30 37 3 12 31 14 37 48 36 15 8 16 13 32 41 45 24 2 28 21 6 13 5 4 3 40 6 35 47 13 2 27 47 26 39 11 19 32 33 39 17 23 26 36 41 35 35 33 43 1 39 28 8 37 17 14 18 5 28 2 33 27 1 5 0 25 45 38 27 32 12 37 39 11 2 31 9 10 20 20 25 19 32 43 29 1 19 1 44 13 19 25 5 33 27 11 1 47 17 44 26 12 26 9 27 23 16 15 44 46 14 16 39 24 33 42 45 20 43 40 27 32 19 2 7 25 4 44 29 46 5 8 34 40 15 5 48 27 48 17 38 41 34 23 26 25 30 1 19 12 45 46 40 9 32 18 9 26 15 45 12 35 17 10 18 2 45 37 37 17 37 7 13 1 33 21 43 44 13""".split()]
OB_CFG2 = [int(x) for x in """15 16 18 9 27 30 26 28 18 19 41 23 18 14 16 25 26 22 10 46 37 46 9 15 41 30 45 21 21 29 15 4 15 3 45 19 46 49 40 14 35 13 35 27 36 36 39 21 43 14 16 4 22 28 20 40 40 31 42 27 13 21 32 13 21 29 21 8 22 4 2 1 26 21 43 27 49 28 43 4 29 42 46 22 47 48 31 28 33 41 11 23 28 39 6 18 10 45 1 41 29 4 25 12 12 39 30 34 2 34 11 33 5 33 11 34 17 8 9 22 13 18 28 25 0 49 19 32 0 33 15 0 18 20 24 46 27 44 0 49 6 49 18 11 13 48 33 14 44 46 40 11 31 5 21 26 41 12 45 10 12 27 21 12 46 25 23 41 24 28 15 14 2 33 2 42 13 8 37 33 0 19 20 38 37 23 16 29 23 0 36 48 44 25 21 0 39 23 7 32 9 28 46 21 6 15 40 27 43 22 25 47 28 25 42 28 33 47 20 9 20 9 9 47 7 46 18 36 7 43 29 36 21 17 22 44 43 11 14 26 29 17 1 18 37 43 42 27 35 24 31 32 4 2 8 3""".split()]

F = lambda *ns: [dict(algo="flash", N=n) for n in ns]
BS = lambda *nb: [dict(algo="flashbs", N=n, B=b) for n, b in nb]
V = [dict(algo="vanilla", N=1)]     # Base_line/C implementations/vanilla Viterbi.c (cross-check baseline)
C = lambda *steps: [dict(algo="checkpoint", N=1, step=st) for st in steps]   # .../checkpoint Viterbi.c; step 0 = floor(sqrt(T)), its main's choice

CASES = [
    dict(name="cfg1_K128_T256", spec=dict(kind="data_script", K=128, M=50, T=256, prob=0.253, seed=12, ob=OB_CFG1),
         runs=F(1, 4, 8) + BS((1, 32), (2, 32), (3, 32), (4, 32), (8, 32), (16, 32), (4, 64), (4, 128)) + V + C(0, 100)),
    dict(name="ds_K200_T100", spec=dict(kind="data_script", K=200, M=50, T=100, prob=0.1, seed=3),
         runs=F(1, 3, 5, 7) + BS((1, 16), (3, 17), (5, 50), (4, 200)) + V + C(0, 7, 99)),
    dict(name="ds_K77_M7_T33", spec=dict(kind="data_script", K=77, M=7, T=33, prob=0.3, seed=5),
         runs=F(1, 2, 3, 4) + BS((1, 8), (3, 9), (4, 77)) + V + C(0, 1, 4, 33, 40)),
    dict(name="ds_K512_T64", spec=dict(kind="data_script", K=512, M=50, T=64, prob=0.05, seed=7),
         runs=F(1, 8) + BS((8, 64), (1, 100))),
    dict(name="ds_K5_T2", spec=dict(kind="data_script", K=5, M=3, T=2, prob=0.9, seed=1), runs=F(1) + BS((1, 2), (1, 5)) + V + C(0)),
    dict(name="ds_K5_T3", spec=dict(kind="data_script", K=5, M=3, T=3, prob=0.9, seed=2), runs=F(1, 2) + BS((1, 3))),
    dict(name="ds_K9_T7", spec=dict(kind="data_script", K=9, M=3, T=7, prob=0.8, seed=4), runs=F(1, 3) + BS((1, 4), (3, 4))),
    dict(name="ties_all_K64_T64", spec=dict(kind="ties_all", K=64, M=4, T=64, prob=0.5, seed=21),
         runs=F(1, 4) + BS((1, 8), (4, 16), (3, 64)) + V + C(0, 3)),
    dict(name="ties_semi_K96_T80", spec=dict(kind="ties_semi", K=96, M=4, T=80, prob=0.5, seed=22),
         runs=F(1, 3, 8) + BS((1, 12), (4, 32), (8, 33)) + V + C(0, 16)),
]
BIG_CASES = [
    dict(name="cfg2_K3965_T256", spec=dict(kind="data_script", K=3965, M=50, T=256, prob=0.112, seed=12, ob=OB_CFG2),
         runs=F(8) + BS((8, 32), (8, 256)) + V + C(0) + F(1) + BS((1, 32))),
    # the reference driver's own second parameter set (src/run.py:17-24: prob 0.169; MAX_THREADS 1, BeamSearchWidth 32)
    dict(name="cfg2b_K3965_T256_p0169", spec=dict(kind="data_script", K=3965, M=50, T=256, prob=0.169, seed=12, ob=OB_CFG2),
         runs=F(8, 1) + BS((8, 32), (1, 32))),
]


def run_key(r):
    return (r["algo"], r["N"], r.get("B"), r.get("step"))


def make_case(case, keep_dir=None, have=None):
    """have: runs already recorded for this case (--append): kept as they are, only the missing ones are executed."""
    spec = case["spec"]
    tmp = keep_dir or tempfile.mkdtemp(prefix="fvgold_")
    try:
        modelgen.write_text(spec, tmp)
        A, B, Pi, ob = modelgen.model32(spec)
        runs = []
        done = {run_key(r): r for r in (have or [])}
        for r in case["runs"]:
            if run_key(r) in done:
                runs.append(done[run_key(r)])
                continue
            kind = r["algo"]
            exe = build_ref.build(kind, spec["K"], spec["T"], spec["prob"], r["N"], r.get("B"), M=spec["M"], score=True,
                                  step=r.get("step", 0))
            out = build_ref.run(exe, tmp)
            rec = dict(r)
            rec.update(path=out["path"], memory=out["memory"], score=out["score"], ref_time_s=out["time"])
            runs.append(rec)
            print(f"  {case['name']} {r}: time {out['time']:.3f}s score {out['score']}")
        return dict(name=case["name"], spec=spec, ob=[int(x) for x in ob],
                    sha=dict(A=modelgen.sha(A), B=modelgen.sha(B), Pi=modelgen.sha(Pi)),
                    runs=runs,
                    provenance="reference src compiled by oracle/build_ref.py (gcc -g -pthread, run.py:54 flags)")
    finally:
        if keep_dir is None:
            shutil.rmtree(tmp, ignore_errors=True)


def main():
    cases = list(CASES)
    if "--big" in sys.argv:
        cases = BIG_CASES if "--only-big" in sys.argv else cases + BIG_CASES
    for case in cases:
        print(case["name"])
        path = os.path.join(HERE, case["name"] + ".json")
        have = None
        if "--append" in sys.argv and os.path.isfile(path):
            with open(path) as f:
                have = json.load(f)["runs"]
        rec = make_case(case, have=have)
        with open(path, "w") as f:
            json.dump(rec, f, separators=(",", ":"))
            f.write("\n")


if __name__ == "__main__":
    main()
