"""A/B timing of FLASH-BS under several FV_OPT_DEBUG values on BASELINE cfg4 / cfg5 (same context, same model, paths compared).
   python tools/ab_beam.py cfg2b32|cfg2b256|cfg4|cfg5  dbg [dbg ...]        (first value = the reference run; 0 = library default)"""
import os, statistics, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import modelgen
from flash_viterbi_amd import decoder
CFG = {"cfg2b32": (dict(kind="data_script", K=3965, M=50, T=256, prob=0.112, seed=12), 8, 32, 9),      # the reference driver's own parameter set (src/run.py)
       "cfg2b256": (dict(kind="data_script", K=3965, M=50, T=256, prob=0.112, seed=12), 8, 256, 9),
       "cfg4": (dict(kind="data_script", K=16384, M=50, T=256, prob=0.112, seed=12), 8, 256, 7),
       "cfg5": (dict(kind="sparse_fast", K=65536, M=50, T=1024, prob=0.112, seed=12), 8, 1024, 3)}
spec, N, B, reps = CFG[sys.argv[1]]
t0 = time.time()
A, Bm, Pi, ob = modelgen.model32(spec)
print(f"{sys.argv[1]}: model in {time.time() - t0:.1f}s", flush=True)
fv = decoder.FlashViterbi(0)
fv.set_model(A, Bm, Pi)
ref = None
for dbg in [int(x, 0) for x in sys.argv[2:]]:
    fv.set_option(decoder.OPT_DEBUG, dbg)
    fv.decode_beam(ob, N, B)                       # warm (tables, workspace)
    ms, top = [], []
    for _ in range(reps):
        path, score, rc = fv.decode_beam(ob, N, B)
        st = fv.stats()
        ms.append(st["gpu_ms"]); top.append(st["top_pass_ms"])
    if ref is None: ref = (path.tolist(), score, rc)
    same = (path.tolist(), score, rc) == ref
    print(f"dbg {dbg:9d}: gpu_ms median {statistics.median(ms):8.3f} min {min(ms):8.3f}  whole-sequence pass {statistics.median(top):8.3f}  "
          f"right-hand {statistics.median(ms) - statistics.median(top):7.3f}  same result {same}  replays {st['beam_exact_sets']} reach {st['beam_reach_events']} cuts {st['beam_chain_cuts']}", flush=True)
fv.close()
