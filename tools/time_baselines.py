"""GPU time of the baseline forms (fv_decode_vanilla, fv_decode_checkpoint) beside FLASH on the cfg2 workload."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder
g = json.load(open(os.path.join(ROOT, "tests/golden/cfg2_K3965_T256.json")))
A, B, Pi, ob = modelgen.model32(g["spec"])
fv = decoder.FlashViterbi(0)
fv.set_model(A, B, Pi)
fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_Q16_REFINE)
def best(fn):
    b = None
    for _ in range(6):
        out = fn(); st = fv.stats()
        if b is None or st["gpu_ms"] < b[1]["gpu_ms"]: b = (out, st)
    return b
ref = None
for name, fn in (("vanilla", lambda: fv.decode_vanilla(ob)), ("checkpoint step=16", lambda: fv.decode_checkpoint(ob, 0)),
                 ("checkpoint step=64", lambda: fv.decode_checkpoint(ob, 64)),
                 ("FLASH single pass (Q16)", lambda: fv.decode_full(ob, 8, decoder.MODE_SINGLE_PASS)),
                 ("FLASH reference schedule (Q16)", lambda: fv.decode_full(ob, 8, decoder.MODE_REFERENCE))):
    (path, score, rc), st = best(fn)
    ref = ref if ref is not None else path.tolist()
    print(f"{name:32s} gpu_ms {st['gpu_ms']:.3f} launches {st['step_launches']} task_steps {st['task_steps']} same path as vanilla {path.tolist() == ref} score {score}")
