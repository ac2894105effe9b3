"""Debug helper: one beam case under several FV_OPT_DEBUG values against the oracle.
   python tools/dbg_beam_case.py kind K M T seed prob B N  dbg..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np, modelgen, oracle
from flash_viterbi_amd import decoder
kind, K, M, T, seed, prob, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), float(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
A, Bm, Pi, ob = modelgen.model32(dict(kind=kind, K=K, M=M, T=T, prob=prob, seed=seed))
om = oracle.OracleModel(A, Bm, Pi)
opath, oscore, _, orc = om.beam_decode(ob, N, B)
fv = decoder.FlashViterbi(0); fv.set_model(A, Bm, Pi)
for dbg in [int(x) for x in sys.argv[9:]]:
    fv.set_option(decoder.OPT_DEBUG, dbg)
    for rep in range(2):
        p, s, rc = fv.decode_beam(ob, N, B)
        st = fv.stats()
        diff = [i for i in range(T) if p[i] != opath[i]]
        print(f"dbg {dbg:8d} rep {rep}: equal {not diff} first diffs {diff[:5]} n {len(diff)} score {s == oscore} rc {rc}/{orc} exact {st['beam_exact_sets']} spec {st['beam_spec_steps']} "
              f"reach {st['beam_reach_events']} ties {st['beam_ties']} dupcols {st['beam_dup_cols']}", flush=True)
