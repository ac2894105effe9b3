"""Step time of the dense kernels against K (fixed cost + slope, cache effects).  GPU box only:
   python tools/scan_k.py [kernel id, default 4 = Q16] [K ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder

kern = int(sys.argv[1]) if len(sys.argv) > 1 else 4
Ks = [int(x) for x in sys.argv[2:]] or [512, 1024, 2048, 2816, 3965, 5632, 8192, 11264]
T = 128
for K in Ks:
    spec = dict(kind="data_script", K=K, M=50, T=T, prob=0.112, seed=12)
    A, B, Pi, ob = modelgen.model32(spec)
    fv = decoder.FlashViterbi(0)
    fv.set_model(A, B, Pi)
    fv.set_option(decoder.OPT_KERNEL, kern)
    fv.set_option(decoder.OPT_DEBUG, int(os.environ.get("FV_DEBUG", "0")))
    best = None
    for rep in range(6):
        fv.decode_full(ob, 1, decoder.MODE_SINGLE_PASS)
        st = fv.stats()
        if best is None or st["top_steps_ms"] < best["top_steps_ms"]:
            best = st
    us = 1e3 * best["top_steps_ms"] / (T - 1)
    tb = best["table_bytes_per_step"]
    print(f"K {K:6d} kernel {best['kernel']} us/step {us:7.2f} table MB/step {tb/1e6:8.2f} streamed TB/s {tb/us/1e6:6.2f} "
          f"per XCD MB {tb/8e6:6.2f}", flush=True)
    fv.close()
