"""Timing of the FLASH-BS path.  python tools/tune_beam.py [K B T]  (default: cfg2 model, B=32 and 256)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder
if len(sys.argv) >= 4:
    K, B, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    t0 = time.time()
    spec = dict(kind="data_script", K=K, M=50, T=T, prob=0.112, seed=12)
    A, Bm, Pi, ob = modelgen.model32(spec)
    print(f"model K={K} generated in {time.time()-t0:.1f}s", flush=True)
    cases = [(B, None)]
else:
    g = json.load(open(os.path.join(ROOT, "tests/golden/cfg2_K3965_T256.json")))
    A, Bm, Pi, ob = modelgen.model32(g["spec"])
    K, T = g["spec"]["K"], g["spec"]["T"]
    cases = [(r["B"], r) for r in g["runs"] if r["algo"] == "flashbs"]
fv = decoder.FlashViterbi(0)
t0 = time.time(); fv.set_model(A, Bm, Pi); print(f"set_model {time.time()-t0:.2f}s", flush=True)
for B, ref in cases:
    best = None
    for rep in range(3):
        p, s, rc = fv.decode_beam(ob, 8, B, 0)
        st = fv.stats()
        if best is None or st["gpu_ms"] < best["gpu_ms"]: best = st
    ok = None if ref is None else (p.tolist() == ref["path"] and s == np.float32(ref["score"]))
    steps = best["task_steps"]
    print(f"K {K} B {B} T {T}: gpu_ms {best['gpu_ms']:.3f} top_ms {best['top_pass_ms']:.3f} us/step(top) {1e3*best['top_pass_ms']/(T-1):.1f} "
          f"launches {best['step_launches']} task_steps {steps} cells/s {K*B*T/(best['gpu_ms']*1e-3):.3e} "
          f"roofline(4*B*K*(T-1)/top) {4.0*B*K*(T-1)/(best['top_pass_ms']*1e-3)/8e12:.4f} exact_sets {best['beam_exact_sets']} dup_steps {best['beam_dup_steps']} dup_cols {best['beam_dup_cols']} ok {ok} rc {rc}", flush=True)
