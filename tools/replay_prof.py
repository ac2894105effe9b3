"""Where a critical-path heap replay of FLASH-BS spends its time (experiment build with -DFV_REPLAY_PROF).
  python tools/replay_prof.py --build          (here: compiles tools/micro/libflashvit_prof.so)
  python tools/replay_prof.py K T N B          (on the GPU box)"""
import os, sys, ctypes, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from flash_viterbi_amd import build as _build
PROF_LIB = os.path.abspath(os.environ.get("FV_PROF_LIB", os.path.join(ROOT, "tools", "micro", "libflashvit_prof.so")))
if sys.argv[1] == "--build":
    _build.HIP_TIMING_LIB = PROF_LIB
    _build.EXTRA_TIMING_FLAGS = ["-DFV_REPLAY_PROF"]
    print(_build.build_hip(force=True, timing=True)); sys.exit(0)
_build.HIP_LIB = PROF_LIB
import numpy as np, modelgen
from flash_viterbi_amd import decoder
K, T, N, B = (int(x) for x in sys.argv[1:5])
spec = dict(kind="sparse_fast" if K > 20000 else "data_script", K=K, M=50, T=T, prob=0.112, seed=12)
A, Bm, Pi, ob = modelgen.model32(spec)
fv = decoder.FlashViterbi(0); fv.set_model(A, Bm, Pi)
L = decoder.load_library()
out = (ctypes.c_ulonglong * 8)()
rp = (ctypes.c_ulonglong * 8)()
for rep in range(2):
    L.fv_debug_replay_prof(out)          # reset
    L.fv_debug_reach_prof(rp)
    p, s, rc = fv.decode_beam(ob, N, B, 0)
    st = fv.stats()
    L.fv_debug_replay_prof(out)
    n = max(1, out[0]); ops = max(1, out[5])
    print(f"K={K} T={T} N={N} B={B}: gpu_ms {st['gpu_ms']:.3f} top_ms {st['top_pass_ms']:.3f} exact_sets {st['beam_exact_sets']}; critical-path replays {out[0]}: "
          f"per replay: init+build {out[1]/n:.0f} ticks, loop {out[2]/n:.0f}, drain {out[3]/n:.0f}; batches {out[4]/n:.1f}, ops {out[5]/n:.1f}, "
          f"empty polls {out[6]/n:.1f}, popped {out[7]/n:.1f}; loop ticks per op {out[2]/ops:.1f} (s_memtime ticks = cycles at 2.4 GHz on this box: tools/micro/lone_wave.hip)", flush=True)
    L.fv_debug_reach_prof(rp)
    ne = max(1, rp[0])
    print(f"   reach events {rp[0]} (selects whose doubtful columns reached the beam, or immediate replays on a dirty step): per event, s_memtime ticks: "
          f"reach test {rp[1]/ne:.0f}, resolve (replays of the undecided steps + their doubtful columns) {rp[2]/ne:.0f}, repeated selection {rp[3]/ne:.0f}, "
          f"replay of the step itself {rp[4]/ne:.0f} ({rp[5]} of them)", flush=True)
