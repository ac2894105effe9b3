"""Decode time of cfg2 (and optionally cfg3) by step kernel: python tools/time_kernels.py [T] [kernel ids...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder
T = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kernels = [int(x) for x in sys.argv[2:]] or [4, 6, 5]
A, B, Pi, ob = modelgen.model32(dict(kind="data_script", K=3965, M=50, T=T, prob=0.112, seed=12))
fv = decoder.FlashViterbi(0); fv.set_model(A, B, Pi)
if os.environ.get("FV_DEBUG"): fv.set_option(decoder.OPT_DEBUG, int(os.environ["FV_DEBUG"]))
ref = None
for k in kernels:
    fv.set_option(decoder.OPT_KERNEL, k)
    best = None
    for rep in range(8):
        try:
            p, s, rc = fv.decode_full(ob, 8, 0)
        except decoder.FlashVitError:          # timing-only debug switches void the result
            p, s = np.zeros(T, np.int32), np.float32(0)
        st = fv.stats()
        if best is None or st["gpu_ms"] < best["gpu_ms"]: best = st
    if ref is None: ref = (p.tolist(), s)
    same = p.tolist() == ref[0] and s == ref[1]
    print(f"kernel {k}: gpu_ms {best['gpu_ms']:.3f} top_pass_ms {best['top_pass_ms']:.3f} us/step(top) {1e3*best['top_steps_ms']/(T-1):.2f} "
          f"right-hand ms {best['gpu_ms']-best['top_pass_ms']:.3f} launches {best['step_launches']} near {best['refine_near']} rescan {best['refine_rescan']} same_as_first {same}", flush=True)
