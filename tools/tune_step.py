"""Timing experiments for the trellis-step kernel on the cfg2 workload (K=3965, T=256).
Not part of the test suite; run on the GPU box:  python tools/tune_step.py [debug bits ...]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder

g = json.load(open(os.path.join(ROOT, "tests/golden/cfg2_K3965_T256.json")))
A, B, Pi, ob = modelgen.model32(g["spec"])
ref = g["runs"][0]
fv = decoder.FlashViterbi(0)
fv.set_model(A, B, Pi)
variants = [int(x) for x in sys.argv[1:]] or [0, 2, 4, 6, 1, 3]
for kern in ((5,) if os.environ.get('FV_ONLY_SPARSE') else (5, 4)):
    for dbg in variants:
        fv.set_option(decoder.OPT_KERNEL, kern)
        fv.set_option(decoder.OPT_DEBUG, dbg)
        for mode in (1, 0):
            best = None
            for rep in range(5):
                p, s, rc = fv.decode_full(ob, 8, mode)
                st = fv.stats()
                if best is None or st["gpu_ms"] < best["gpu_ms"]:
                    best = st
            ok = p.tolist() == ref["path"] and s == np.float32(ref["score"])
            print(f"kern {kern} dbg {dbg} mode {mode} ok {ok} gpu_ms {best['gpu_ms']:.3f} top_ms {best['top_pass_ms']:.3f} "
                  f"us/step(top) {1e3*best['top_pass_ms']/255:.2f} launches {best['step_launches']} near {best['refine_near']} rescan {best['refine_rescan']}", flush=True)
