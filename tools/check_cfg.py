"""One-off full-size checks against the oracle (too long for the test suite).
  python tools/check_cfg.py full K T N          e.g. full 3965 4096 8   (BASELINE configs[2] on one GPU)
  python tools/check_cfg.py beam K T N B        e.g. beam 16384 256 8 256 (configs[3])"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, modelgen, oracle
from flash_viterbi_amd import build as _build
if os.environ.get("FV_LIB"): _build.HIP_LIB = os.path.abspath(os.environ["FV_LIB"])     # A/B runs against another build of the library
from flash_viterbi_amd import decoder
kind = sys.argv[1]; K, T, N = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B = int(sys.argv[5]) if kind == "beam" else 0
spec = dict(kind="sparse_fast" if K > 20000 else "data_script", K=K, M=50, T=T, prob=0.112, seed=12)
t0 = time.time(); A, Bm, Pi, ob = modelgen.model32(spec); print(f"model {time.time()-t0:.1f}s", flush=True)
fv = decoder.FlashViterbi(0); t0 = time.time(); fv.set_model(A, Bm, Pi); print(f"set_model {time.time()-t0:.2f}s", flush=True)
if os.environ.get("FV_KERNEL"): fv.set_option(decoder.OPT_KERNEL, int(os.environ["FV_KERNEL"]))   # 4 = dense Q16, 5 = sparse walk
if os.environ.get("FV_DEBUG"): fv.set_option(decoder.OPT_DEBUG, int(os.environ["FV_DEBUG"]))
if os.environ.get("FV_SEL_MARGIN"): fv.set_option(decoder.OPT_SEL_MARGIN, int(os.environ["FV_SEL_MARGIN"]))
best = None
for rep in range(3):
    p, s, rc = fv.decode_full(ob, N, 0) if kind == "full" else fv.decode_beam(ob, N, B, 0)
    st = fv.stats()
    if best is None or st["gpu_ms"] < best["gpu_ms"]: best = st
cells = K * (B or K) * T
print(f"{kind} K={K} T={T} N={N} B={B} kernel={best['kernel']}: gpu_ms {best['gpu_ms']:.3f} decode_ms {best['decode_ms']:.3f} top_ms {best['top_pass_ms']:.3f} "
      f"cells/s {cells/(best['gpu_ms']*1e-3):.4e} passes {best['passes']} launches {best['step_launches']} task_steps {best['task_steps']} "
      f"near {best['refine_near']} rescan {best['refine_rescan']} exact_sets {best['beam_exact_sets']} cand_selects {best['beam_cand_selects']} ties {best['beam_ties']} spec {best['beam_spec_steps']} reach {best['beam_reach_events']} list short/long {best['beam_list_short']}/{best['beam_list_long']} mean list {best['beam_list_entries']/max(1,best['beam_cand_selects']):.0f} rc {rc}", flush=True)
if "--no-oracle" not in sys.argv:
    oracle.set_threads(16)
    print("oracle: building log tables ...", flush=True)
    om = oracle.OracleModel(A, Bm, Pi)
    print("oracle: decoding ...", flush=True)
    t0 = time.time()
    op, osc, oc, orc = om.full_decode(ob, N) if kind == "full" else om.beam_decode(ob, N, B)
    dt = time.time() - t0
    print(f"oracle {dt:.1f}s ({cells/dt:.3e} cells/s, 16 OpenMP threads) path_equal {p.tolist()==op.tolist()} score_equal {s==osc} ({s} vs {osc}) rc {orc}", flush=True)
