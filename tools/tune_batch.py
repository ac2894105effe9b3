"""Timing of the batched generations (reference mode minus the whole-sequence pass) on cfg2."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, modelgen
from flash_viterbi_amd import decoder
g = json.load(open(os.path.join(ROOT, "tests/golden/cfg2_K3965_T256.json")))
A, B, Pi, ob = modelgen.model32(g["spec"])
ref = g["runs"][0]
fv = decoder.FlashViterbi(0)
fv.set_model(A, B, Pi)
for kern in ([int(x) for x in os.environ.get("FV_KERNS", "4,2,3").split(",")]):
    for dbg in ([int(x) for x in os.environ.get("FV_DBGS", "0").split(",")]):
        for mb in (4, 8):
            fv.set_option(decoder.OPT_KERNEL, kern); fv.set_option(decoder.OPT_DEBUG, dbg); fv.set_option(decoder.OPT_MAX_BATCH, mb)
            best = None
            for rep in range(4):
                p, s, rc = fv.decode_full(ob, 8, 0)
                st = fv.stats()
                if best is None or st["gpu_ms"] < best["gpu_ms"]: best = st
            rest = best["gpu_ms"] - best["top_pass_ms"]
            nl = best["step_launches"] - 255
            print(f"kern {kern} dbg {dbg} max_batch {mb}: total {best['gpu_ms']:.3f} rest {rest:.3f} ms launches {nl} task_steps {best['task_steps']-255} "
                  f"us/launch {1e3*rest/nl:.1f} us/task_step {1e3*rest/(best['task_steps']-255):.2f} ok {p.tolist()==ref['path']}", flush=True)
