#!/usr/bin/env python3
"""Compile the reference's own decoders into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

The reference programs are configured at compile time (their `#define` block,
reference src/FLASH_Viterbi_multithread.c:10-16, src/FLASH_BS_Viterbi_multithread.c:10-17).
Its bench driver patches that block with regular expressions and runs gcc
(reference src/run.py:29-54).  This script does the same thing, with two
differences: the patched text is piped to gcc on stdin (no copy of the source
is ever written anywhere) and the only output is the executable under
oracle/_ref/ (git-ignored; it travels to the GPU box with the snapshot because
/root/reference does not exist there).

Every binary reads `{A,B,Pi,ob}_K{K}_T{T}_prob{p}.txt` from its current working
directory (data_path is patched to "./") and prints the reference's three
stdout lines.  With score=True one fprintf(stderr) of the whole-sequence
pass's final score is spliced in after the end-state argmax; nothing else changes.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/src"
REF_BASE = "/root/reference/Base_line/C implementations"
OUT = os.path.join(HERE, "_ref")

SOURCES = {"flash": "FLASH_Viterbi_multithread.c", "flashbs": "FLASH_BS_Viterbi_multithread.c",
           "vanilla": "vanilla Viterbi.c", "checkpoint": "checkpoint Viterbi.c"}
BASELINES = ("vanilla", "checkpoint")      # single-threaded programs of Base_line/: no MAX_THREADS in their config block


def source_path(kind):
    return os.path.join(REF_BASE if kind in BASELINES else REF_SRC, SOURCES[kind])
# gcc flags exactly as reference src/run.py:54
GCC = ["gcc", "-g", "-pthread", "-x", "c", "-", "-lm", "-Wl,-z,stack-size=268435456"]


def ref_name(kind, K, T, prob, N, beam=None, M=50, score=False, step=0):
    s = f"{kind}_K{K}_T{T}_p{prob}_N{N}"
    if kind == "flashbs":
        s += f"_B{beam}"
    if kind == "checkpoint" and step:
        s += f"_S{step}"
    if M != 50:
        s += f"_M{M}"
    if score:
        s += "_score"
    return s


def reference_available():
    return all(os.path.isfile(source_path(k)) for k in SOURCES)


def _patch(text, kind, K, T, prob, N, beam, M, score, step=0):
    def sub1(pat, repl, s):
        out, n = re.subn(pat, repl, s)
        if n < 1:
            raise RuntimeError(f"pattern not found in reference source: {pat}")
        return out
    text = sub1(r"#define K_STATE \d+", f"#define K_STATE {K}", text)
    text = sub1(r"#define T_STATE \d+", f"#define T_STATE {M}", text)
    text = sub1(r"#define obserRouteLEN \d+", f"#define obserRouteLEN {T}", text)
    text = sub1(r"const float prob = \d+\.\d+;", f"const float prob = {prob};", text)
    text = sub1(r'const char data_path\[\] = "[^"]*";', 'const char data_path[] = "./";', text)
    if kind not in BASELINES:
        text = sub1(r"#define MAX_THREADS \d+", f"#define MAX_THREADS {N}", text)
    if kind == "checkpoint" and step:
        # main() passes 0 (= floor(sqrt(T)) inside); a fixed step is the function's own second argument
        text = sub1(r"viterbi_checkpoint\(vit,0\);", f"viterbi_checkpoint(vit,{step});", text)
    if kind == "flashbs":
        text = sub1(r"const int BeamSearchWidth = \d+;", f"const int BeamSearchWidth = {beam};", text)
    digits = len(str(prob).split(".")[1]) if "." in str(prob) else 0
    text = sub1(r"prob%\.\d+f", f"prob%.{digits}f", text)
    if score:
        # after the whole-sequence end-state pick ("vit->Ans[R] = arg;" / "... .State;")
        if kind == "vanilla":
            text = sub1(r"(vit->Ans\[obserRouteLEN-1\] = arg;)", r'\1 fprintf(stderr, "score: %.9g\\n", (double)tmp);', text)
        elif kind == "checkpoint":
            text = sub1(r"(vit->Ans\[\(\*count\)--\] = arg;)", r'\1 fprintf(stderr, "score: %.9g\\n", (double)tmp);', text)
        elif kind == "flash":
            text = sub1(r"(vit->Ans\[R\] = arg;)", r'\1 fprintf(stderr, "score: %.9g\\n", (double)score);', text)
        else:
            text = sub1(r"(vit->Ans\[R\] = H\[cur\](?:\[1\])?\[arg\+1\]\.State;)",
                        r'\1 fprintf(stderr, "score: %.9g\\n", (double)score);', text)
    return text


def build(kind, K, T, prob, N, beam=None, M=50, score=False, force=False, step=0):
    """Returns the path of the binary, building it if the reference tree is present.
    Raises FileNotFoundError when neither the binary nor the reference exists."""
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, ref_name(kind, K, T, prob, N, beam, M, score, step))
    if os.path.isfile(exe) and not force:
        return exe
    if not reference_available():
        raise FileNotFoundError(f"{exe} not prebuilt and {REF_SRC} is absent")
    with open(source_path(kind), "r") as f:
        text = _patch(f.read(), kind, K, T, prob, N, beam, M, score, step)
    res = subprocess.run(GCC + ["-o", exe], input=text, text=True, capture_output=True)
    if res.returncode != 0:
        raise RuntimeError(f"gcc failed for {exe}:\n{res.stderr}")
    return exe


def run(exe, data_dir, timeout=3600):
    """Runs a reference binary in data_dir; returns dict(time, path, memory, score)."""
    res = subprocess.run([exe], cwd=data_dir, capture_output=True, text=True, timeout=timeout)
    if res.returncode != 0:
        raise RuntimeError(f"{exe} exited {res.returncode}: {res.stderr[-400:]}")
    out = res.stdout
    t = float(re.search(r"time: ([\d.]+)", out).group(1))          # reference run.py:75
    mem = int(re.search(r"memory: (-?\d+)", out).group(1))        # reference run.py:76
    path = [int(x) for x in re.search(r"path: \[([^\]]*)\]", out).group(1).split()]
    ms = re.search(r"score: (\S+)", res.stderr)
    return {"time": t, "memory": mem, "path": path, "score": float(ms.group(1)) if ms else None}


# Binaries that must exist on the GPU box (no reference tree there): the bench's
# cpu_baseline sample.  Golden-vector binaries are built on demand by
# tests/golden/make_golden.py.
DEFAULT_SET = [
    dict(kind="flash", K=3965, T=256, prob=0.112, N=8),    # cfg2, the bench configuration itself (1 warm + 3 timed)
    dict(kind="flash", K=3965, T=16, prob=0.112, N=1),     # MAX_THREADS=1 sample
    dict(kind="flash", K=3965, T=64, prob=0.112, N=8),     # bounded sample of cfg3 (same model, T=4096 sequence)
    dict(kind="flashbs", K=3965, T=256, prob=0.112, N=8, beam=256),    # bench.py's flash_bs extra of cfg2 (full sequence)
    dict(kind="flashbs", K=16384, T=64, prob=0.112, N=8, beam=256),    # bounded sample of cfg4
]

if __name__ == "__main__":
    if not reference_available():
        print("build_ref: /root/reference absent; keeping prebuilt oracle/_ref as is")
        sys.exit(0)
    for cfg in DEFAULT_SET:
        print(build(**cfg))
