"""CPU, world_size 2, gloo: the N>1 path.  Each rank executes exactly the passes the product's
plan assigns to it (fv_plan_passes with nranks=2) — with the oracle's forward pass standing in
for the GPU kernels — all-gathers the answer arrays and applies the product's own merge
(fv_merge_paths).  The merged path must equal the single-rank decode bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["FV_ROOT"]); sys.path.insert(0, os.path.join(os.environ["FV_ROOT"], "tests")); sys.path.insert(0, os.path.join(os.environ["FV_ROOT"], "oracle"))
import numpy as np, torch, torch.distributed as dist
import modelgen, oracle
from flash_viterbi_amd import decoder
from test_schedule import run_plan_on_cpu
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
oracle.set_threads(2)
for (K, M, T, N, seed) in [(120, 9, 96, 8, 31), (64, 5, 50, 5, 32), (90, 4, 41, 3, 33)]:
    spec = dict(kind="data_script", K=K, M=M, T=T, prob=0.25, seed=seed)
    A, B, Pi, ob = modelgen.model32(spec)
    m = oracle.OracleModel(A, B, Pi)
    mine, _ = run_plan_on_cpu(m, ob, N, 0, world, rank)
    t = torch.from_numpy(mine.astype(np.int32))
    bufs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)                                   # the one collective of the path
    merged = decoder.merge_paths(T, N, world, torch.stack(bufs).numpy())
    want, _, _, _ = m.full_decode(ob, N)
    assert merged.tolist() == want.tolist(), (K, T, N, rank)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_two_ranks_sharded_segments_equal_single_rank(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, FV_ROOT=ROOT, OMP_NUM_THREADS="2")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert res.stdout.count("ok") == 2
