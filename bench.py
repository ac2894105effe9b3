#!/usr/bin/env python3
"""bench.py — FLASH / FLASH-BS Viterbi decode throughput on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one complete decode of the workload, model tables already resident in HBM (fv_set_model is
outside the timed region; the observation upload and the path download, a few KB, are inside it).

  cfg2 (default)  BASELINE configs[1]: full-state FLASH, K=3965 T=256 M=50 prob=0.112 seed=12 n_split=8,
                  FV_MODE_REFERENCE (the reference's divide-and-conquer task tree replayed pass for pass)
  cfg3            configs[2]: the same model, T=4096 (the N>1 case north_star names)
  cfg4            configs[3]: FLASH-BS K=16384 T=256 beam=256 n_split=8
  cfg5            configs[4]: FLASH-BS K=65536 T=1024 beam=1024 n_split=8

metric = trellis cells/s with cells = K*K*T per decode (BASELINE.json's K·K·T ops); for the FLASH-BS
workloads the same JSON line carries K*beam*T cells (report K·B·(T-1) and say so) in
config.cells_per_decode.

N > 1: the n_split top-level segments are dealt round-robin to ranks (one process per GPU), every rank
runs the whole-sequence pass, one RCCL all-gather merges the path slices (include/flashvit.h,
fv_comm_init).  Same total work for every N => "scaling": "strong".  The whole-sequence pass is serial in
T, so the curve is Amdahl-bound (Multi-GPU).

cpu_baseline (rank 0, N = 1 only): SURVEY 8(d)'s protocol for the bench configuration — the reference
program itself (oracle/_ref, built from the reference's sources with run.py:54's flags), MAX_THREADS =
n_split, on the FULL cfg2 sequence, 1 warm + 3 timed runs of its own `time:` line, median; plus one
MAX_THREADS=1 run on a T=16 sample.  `cores` = physical cores of this host, `threads` = what the program
used.  cfg3 uses a T=64 sample of the reference program; cfg4 the reference FLASH-BS program on a T=64 sample (its
5 GB of text inputs written once into a scratch directory) with the oracle port beside it; cfg5 the port alone
(82 GB of text).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import shutil
import statistics
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from flash_viterbi_amd import decoder, hostio  # noqa: E402
from flash_viterbi_amd.generate_data import data_script  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
M_SYMBOLS, PROB, SEED, N_SPLIT = 50, 0.112, 12, 8

WORKLOADS = {
    "cfg2": dict(kind="full", K=3965, T=256, beam=0, gen="data_script", steps=50, warmup=5, label="BASELINE configs[1]"),
    "cfg3": dict(kind="full", K=3965, T=4096, beam=0, gen="data_script", steps=5, warmup=1, label="BASELINE configs[2]"),
    "cfg4": dict(kind="beam", K=16384, T=256, beam=256, gen="data_script", steps=10, warmup=2, label="BASELINE configs[3]"),
    "cfg5": dict(kind="beam", K=65536, T=1024, beam=1024, gen="fast", steps=3, warmup=1, label="BASELINE configs[4]"),
}
KERNEL_NAMES = {1: "fvk::trellis_step<double,1,2,true>", 2: "fvk::trellis_step<float,1,4,true>",
                3: "fvk::trellis_step<fvk::half_t,1,16,false>", 4: "fvk::trellis_step<fvk::q16_t,1,16,false>",
                5: "fvk::trellis_step_sparse<1>", 6: "fvk::trellis_step_u16<1,16,false,8>"}
DENSE_KERNEL = decoder.KERNEL_U16_REFINE      # the library's AUTO choice for a dense model: 16-bit table, every K*K cell swept


def build_workload(w):
    """(model64 or None, (A, B, Pi) float32 as the reference loader reads them, observations)."""
    K, T = w["K"], w["T"]
    ob = np.asarray(data_script.make_observations(T, M_SYMBOLS, SEED), dtype=np.int32)
    if w["gen"] == "fast":       # K = 65536: generate_data's own random stream takes minutes to replay (see its docstring)
        return None, data_script.make_model32_fast(K, M_SYMBOLS, SEED, PROB), ob
    A64, B64, Pi64 = data_script.make_model64(K, M_SYMBOLS, SEED, PROB)
    return (A64, B64, Pi64), (hostio.quantize_text16(A64), hostio.quantize_text16(B64), hostio.quantize_text16(Pi64)), ob


def host_cores():
    """(physical cores of the machine, CPUs this process may run on)."""
    phys = set()
    try:
        pkg = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("physical id"):
                    pkg = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if core is not None:
                        phys.add((pkg, core))
                    pkg = core = None
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return (len(phys) or (os.cpu_count() or 1)), usable


def run_reference(kind, K, T, N, beam, model64, ob, runs, warm):
    """The reference program (oracle/_ref) on the first T observations: list of its `time:` values and its path."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import build_ref
    exe = build_ref.build(kind, K, T, PROB, N, beam if kind == "flashbs" else None)     # FileNotFoundError if not prebuilt
    A64, B64, Pi64 = model64
    tmp = tempfile.mkdtemp(prefix="fvbench_")
    times, path = [], None
    try:
        data_script.write_files(tmp, K, T, PROB, A64, B64, Pi64, ob[:T], text=True)
        for r in range(warm + runs):
            out = build_ref.run(exe, tmp, timeout=1200)
            path = out["path"]
            if r >= warm:
                times.append(out["time"])
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return times, path


def cpu_baseline_full(w, model64, ob, fv):
    """SURVEY 8(d) CPU baseline for the full-state workloads (reference binary, see the module docstring)."""
    K, T = w["K"], w["T"]
    phys, usable = host_cores()
    full = w["T"] <= 256                      # cfg2: the bench configuration itself; cfg3: a T=64 sample
    Ts = T if full else 64
    hip_sample, _, _ = fv.decode_full(ob[:Ts], N_SPLIT, decoder.MODE_REFERENCE)
    try:
        times, path = run_reference("flash", K, Ts, N_SPLIT, 0, model64, ob, runs=3 if full else 1, warm=1 if full else 0)
    except FileNotFoundError:
        return cpu_baseline_port(w, model64, ob, fv, None)
    med = statistics.median(times)
    out = {"value": K * K * Ts / med, "unit": "cells/s", "cores": phys, "threads": N_SPLIT, "usable_cpus": usable,
           "kind": "reference", "seconds": med, "runs": times, "protocol": "1 warm + 3 timed, median" if full else "1 run",
           "sample": (f"K={K} T={Ts}" + (" (the full bench sequence)" if full else f" (first {Ts} observations of the bench workload)") +
                      f", MAX_THREADS={N_SPLIT}; reference src/FLASH_Viterbi_multithread.c built with run.py:54 flags, its own `time:` line"),
           "path_equal_to_hip": bool(path == hip_sample.tolist())}
    if full:
        T1 = 16
        try:
            t1, p1 = run_reference("flash", K, T1, 1, 0, model64, ob, runs=1, warm=0)
            h1, _, _ = fv.decode_full(ob[:T1], 1, decoder.MODE_REFERENCE)
            out["single_thread"] = {"value": K * K * T1 / t1[0], "unit": "cells/s", "threads": 1, "seconds": t1[0],
                                    "sample": f"K={K} T={T1} (first {T1} observations), MAX_THREADS=1",
                                    "path_equal_to_hip": bool(p1 == h1.tolist())}
        except FileNotFoundError:
            pass
    return out


def cpu_baseline_port(w, model64, ob, fv, f32):
    """The oracle port (OpenMP) on a bounded sample: used where the reference binary is not practical."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle
    K, T, beam = w["K"], w["T"], w["beam"]
    phys, usable = host_cores()
    threads = oracle.set_threads(min(usable, 16))
    if f32 is None:
        f32 = tuple(hostio.quantize_text16(x) for x in model64)
    Ts = min(T, 256 if K <= 16384 else 64)
    om = oracle.OracleModel(*f32)
    t0 = time.time()
    if w["kind"] == "beam":
        path, _, _, _ = om.beam_decode(ob[:Ts], N_SPLIT, beam)
        hip, _, _ = fv.decode_beam(ob[:Ts], N_SPLIT, beam, decoder.MODE_REFERENCE)
    else:
        path, _, _, _ = om.full_decode(ob[:Ts], N_SPLIT)
        hip, _, _ = fv.decode_full(ob[:Ts], N_SPLIT, decoder.MODE_REFERENCE)
    dt = time.time() - t0
    om.close()
    return {"value": K * K * Ts / dt, "unit": "cells/s", "cores": phys, "threads": threads, "usable_cpus": usable, "kind": "port",
            "seconds": dt, "protocol": "1 run",
            "sample": f"K={K} T={Ts} (first {Ts} observations) n_split={N_SPLIT}" + (f" beam={beam}" if beam else "") +
                      "; oracle/flashvit_oracle.c (OpenMP), log tables precomputed"
                      + ("; the reference program's text inputs would be 5 GB (K=16384) / 82 GB (K=65536) for its fscanf loader" if K > 4096 else ""),
            "beam_cells_per_sec": (K * beam * Ts / dt) if beam else None,
            "path_equal_to_hip": bool(path.tolist() == hip.tolist())}


def cpu_baseline_beam(w, model64, ob, fv, f32):
    """SURVEY 8(d) for the FLASH-BS workloads: the reference program itself (src/FLASH_BS_Viterbi_multithread.c:579-591,
    built by oracle/build_ref.py) on the first 64 observations — its text inputs are written once into a scratch
    directory (the transition matrix of K=16384 is ~5 GB of '%.16f' text, which is also why the sample is bounded) —
    with the OpenMP port's figure beside it.  K=65536 (82 GB of text) keeps the port alone."""
    port = cpu_baseline_port(w, model64, ob, fv, f32)
    if model64 is None:
        return port
    K, beam = w["K"], w["beam"]
    Ts = 64
    phys, usable = host_cores()
    try:
        t0 = time.time()
        times, path = run_reference("flashbs", K, Ts, N_SPLIT, beam, model64, ob, runs=1, warm=0)
        wall = time.time() - t0
    except (FileNotFoundError, OSError, RuntimeError) as e:
        port["reference_program"] = f"not run: {type(e).__name__}: {str(e)[:200]}"
        return port
    hip, _, _ = fv.decode_beam(ob[:Ts], N_SPLIT, beam, decoder.MODE_REFERENCE)
    return {"value": K * K * Ts / times[0], "unit": "cells/s", "cores": phys, "threads": N_SPLIT, "usable_cpus": usable,
            "kind": "reference", "seconds": times[0], "protocol": "1 run", "beam_cells_per_sec": K * beam * Ts / times[0],
            "sample": f"K={K} T={Ts} (first {Ts} observations of the bench workload) beam={beam}, MAX_THREADS={N_SPLIT}; reference "
                      "src/FLASH_BS_Viterbi_multithread.c built with run.py:54 flags, its own `time:` line (calc() only); "
                      f"writing and parsing its text inputs took the rest of {wall:.0f} s",
            "path_equal_to_hip": bool(path == hip.tolist()), "port": port}


def load_traffic(kernel_name):
    """PMC-measured HBM bytes per launch of `kernel_name` from the committed profile summary, with its provenance."""
    tr_file = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.isfile(tr_file):
        return None, None
    with open(tr_file) as f:
        trj = json.load(f)
    ent = trj.get("by_kernel", {}).get(kernel_name)
    if not ent:
        return None, None
    src = f"profiles/traffic.json ({trj.get('source', 'rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes')}; {trj.get('date', 'round 1')}); NOT measured by this run"
    return ent.get("hbm_bytes_per_launch"), src


def heaviest_kernel(traffic_json_kernel_of):
    """The step kernel with the largest summed time in the committed rocprofv3 summary of this workload
    (profiles/r03_bench_kernel_stats.csv), with its PMC bytes per launch and physical HBM fraction — for the dense decode
    that is the batched launch of the right-hand generations, not the kernel `roofline.achieved` describes."""
    import csv
    import re
    path = os.path.join(ROOT, "profiles", "r03_bench_kernel_stats.csv")
    if not os.path.isfile(path):
        return None
    best = None
    with open(path) as f:
        for r in csv.DictReader(f):
            if "trellis_step" not in r["Name"] or "sparse" in r["Name"]:
                continue
            if best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"]):
                best = r
    if best is None:
        return None
    name = re.sub(r"^void ", "", re.sub(r"\(.*$", "", best["Name"])).replace(", ", ",")
    traffic, src = load_traffic(name)
    avg_us = float(best["AverageNs"]) / 1e3
    return {"kernel": name, "calls_in_profile": int(best["Calls"]), "avg_us": avg_us, "share_of_profiled_gpu_time_pct": float(best["Percentage"]),
            "traffic": traffic, "physical_gbs": (traffic / (avg_us * 1e-6) / 1e9) if traffic else None,
            "physical_frac": (traffic / (avg_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            "source": "profiles/r03_bench_kernel_stats.csv (rocprofv3 --kernel-trace --stats of this command) + " + (src or "no PMC figure"),
            "note": ("a batched launch of the right-hand generations: its tasks share one sweep of the table, 4 B/cell is no bound for it, only the "
                     "physical fraction is quoted" if not name.startswith("fvk::trellis_step_u16<1,") else
                     "the single-task launch of the whole-sequence pass, i.e. the kernel `roofline.achieved` describes (rocprofv3's average; "
                     "the batched right-hand launches, trellis_step_u16<4,2,true,8>, come second)")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="cfg2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    w = WORKLOADS[args.workload]
    if args.steps is None:
        args.steps = w["steps"]
    if args.warmup is None:
        args.warmup = w["warmup"]
    K, T, beam = w["K"], w["T"], w["beam"]
    is_beam = w["kind"] == "beam"

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    model64, (A, B, Pi), ob = build_workload(w)
    gather_mode = "ncclAllGather inside libflashvit"
    dist = None
    # FV_BENCH_BACKEND=gloo: rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks (ranks share
    # devices, tensors of the rendezvous stay on the CPU, RCCL is not used: two ranks cannot share a GPU in one
    # communicator).  The driver's runs use the default, nccl (= RCCL), one rank per GPU.
    backend = os.environ.get("FV_BENCH_BACKEND", "nccl")
    tdev = "cpu"
    if "RANK" in os.environ:      # launched by torch.distributed.run (also exercised with 1 rank)
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
            tdev = "cuda"
        else:
            dist.init_process_group(backend=backend)
            local_rank = local_rank % max(1, torch.cuda.device_count())

    def new_decoder():
        d = decoder.FlashViterbi(local_rank)
        d.set_model(A, B, Pi)
        # full-state workloads: the timed region runs the DENSE streaming kernel (every one of the K*K cells of
        # every step is read and evaluated: that is what "K*K*T cells" counts).  The library's AUTO choice for
        # this model is the sparse walk (non-zero transitions only, same bits out); it is measured afterwards
        # and reported as `sparse_walk`, never as `value`.
        if not is_beam:
            d.set_option(decoder.OPT_KERNEL, DENSE_KERNEL)
        return d

    fv = new_decoder()
    single_rank_path = None
    if dist is not None and world > 1:
        # before the context is partitioned: the whole decode on this rank alone, to check the merged result against
        single_rank_path = (fv.decode_beam(ob, N_SPLIT, beam, decoder.MODE_REFERENCE) if is_beam
                            else fv.decode_full(ob, N_SPLIT, decoder.MODE_REFERENCE))[0].tolist()
    if dist is not None:
        # torch.distributed is the rendezvous only: the 128-byte RCCL id travels over it, the data-path
        # collective (one all-gather per decode) is issued by libflashvit on its own stream
        uid = [decoder.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        ok = torch.ones(1, device=tdev)
        try:
            if backend != "nccl":
                raise decoder.FlashVitError(-8, "rehearsal backend: RCCL communicator not created")
            # ncclCommInitRank blocks until every rank has entered it: a rank whose peer failed before getting there
            # would wait for ever and never reach the all_reduce below.  Watchdog: the whole job ends non-zero instead
            # (torch.distributed.run tears the other ranks down), it never hangs and never prints a half-made line.
            import threading
            box = {}

            def _init():
                try:
                    fv.comm_init(rank, world, uid[0])
                    box["ok"] = True
                except BaseException as e:          # noqa: BLE001 - re-raised on the main thread
                    box["err"] = e
            th = threading.Thread(target=_init, daemon=True)
            th.start()
            th.join(float(os.environ.get("FV_COMM_INIT_TIMEOUT", "180")))
            if th.is_alive():
                print(f"[rank {rank}] fv_comm_init did not return within its time limit: a peer never entered "
                      "ncclCommInitRank; aborting the job", file=sys.stderr, flush=True)
                os._exit(3)
            if "err" in box:
                raise box["err"]
        except decoder.FlashVitError as e:
            print(f"[rank {rank}] fv_comm_init failed ({e}); falling back to torch.distributed all_gather", file=sys.stderr)
            ok.zero_()
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)            # every rank takes the same route
        if ok.item() == 0:
            if fv._h:
                fv.close()
            fv = new_decoder()
            fv.set_partition(rank, world)
            gather_mode = "torch.distributed.all_gather + fv_merge_paths"

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if tdev == "cuda":
                torch.cuda.synchronize()

    def decode():
        # synchronous: returns after the library's stream has drained
        if is_beam:
            path, score, rc = fv.decode_beam(ob, N_SPLIT, beam, decoder.MODE_REFERENCE)
        else:
            path, score, rc = fv.decode_full(ob, N_SPLIT, decoder.MODE_REFERENCE)
        if dist is not None and gather_mode.startswith("torch"):
            import torch
            mine = torch.from_numpy(path).to(tdev)
            bufs = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(bufs, mine)
            path = decoder.merge_paths(T, N_SPLIT, world, torch.stack(bufs).cpu().numpy())
        return path, score, rc

    for _ in range(args.warmup):
        decode()
    barrier()
    t0 = time.perf_counter()
    top_ms = steps_ms = 0.0
    for _ in range(args.steps):
        path, score, rc = decode()
        s_ = fv.stats()
        top_ms += s_["top_pass_ms"]
        steps_ms += s_["top_steps_ms"]      # HIP events on the library's stream around the T-1 steps of the whole-sequence pass
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        tmax = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    st = fv.stats()

    # Roofline of the dominant kernel: the step of the whole-sequence pass (T-1 dependent steps per decode).
    #   full-state: one trellis_step launch per step, 4*K*K algorithmic bytes (4 B per trellis cell);
    #   FLASH-BS:   beam_step + top-B select per step, 4*B*K algorithmic bytes (4 B per (beam entry, destination) cell).
    # Step time = HIP-event time around those T-1 back-to-back steps in the TIMED region / (T-1), i.e. kernel time plus
    # the dispatch gap to the next launch (rocprofv3's per-kernel average, profiles/, excludes part of that gap).
    launch_us = 1e3 * (steps_ms / args.steps) / (T - 1)
    alg_bytes = 4.0 * K * (beam if is_beam else K)
    achieved = alg_bytes / (launch_us * 1e-6) / 1e9
    kname = "fvb::beam_step(+_q16) + fvb::topb_select" if is_beam else KERNEL_NAMES[st["kernel"]]
    if is_beam:
        # PMC passes over this workload (tools/prof_beam_pmc.sh): bytes of a single-pass launch of the step kernel the
        # whole-sequence pass uses, counters calibrated on the kernel's own gather pattern (tools/micro/gather_calib.hip)
        step_kernel = "fvb::beam_step_q16" if K * beam * 8.0 >= 80e6 else "fvb::beam_step"
        traffic, traffic_src = load_traffic(f"{step_kernel}@{args.workload}")
        if traffic_src:
            traffic_src += f"; median single-pass {step_kernel} launch (the whole-sequence pass), selects not included"
    else:
        traffic, traffic_src = load_traffic(kname)
    roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "achieved_is": "ALGORITHMIC GB/s (4 B per cell / step time): the SURVEY 8(d) fraction, not bytes moved",
                "traffic": traffic, "traffic_source": traffic_src,
                "physical_gbs": (traffic / (launch_us * 1e-6) / 1e9) if traffic else None,
                "physical_frac": (traffic / (launch_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "kernel": kname, "launch_us": launch_us, "alg_bytes_per_launch": alg_bytes, "launches_per_decode": T - 1,
                "table_bytes_streamed_per_launch": st["table_bytes_per_step"]}
    if not is_beam and args.workload == "cfg2":
        roofline["heaviest_by_total_time"] = heaviest_kernel(None)

    extra = {}
    if not is_beam:
        fv.decode_full(ob, N_SPLIT, decoder.MODE_SINGLE_PASS)
        extra["single_pass_mode_ms"] = fv.stats()["gpu_ms"]
        # the library's AUTO kernel for this model (sparse walk over the non-zero transitions)
        fv.set_option(decoder.OPT_KERNEL, decoder.KERNEL_AUTO)
        for _ in range(2):
            decode()
        a_ = fv.stats()
        if a_["kernel"] == decoder.KERNEL_SPARSE_Q16:
            nd = max(3, args.steps // 4)
            barrier()
            t1 = time.perf_counter()
            ssteps = 0.0
            for _ in range(nd):
                decode()
                ssteps += fv.stats()["top_steps_ms"]
            barrier()
            sw = time.perf_counter() - t1
            extra["sparse_walk"] = {
                "kernel": "fvk::trellis_step_sparse<1>", "transition_density": a_["density"],
                "decode_ms": 1e3 * sw / nd, "cells_per_sec_dense_equivalent": K * K * T * nd / sw,
                "launch_us": 1e3 * (ssteps / nd) / (T - 1), "stored_bytes_per_launch": a_["table_bytes_per_step"],
                "note": "only the non-zero transitions are stored and visited (log 0 = -inf can never win, "
                        "reference FLASH_Viterbi_multithread.c:171): bit-identical output, fewer cells evaluated; "
                        "not comparable with the HBM roofline of the K*K sweep"}
        fv.set_option(decoder.OPT_KERNEL, DENSE_KERNEL)

    if rank == 0:
        cells = K * K * T
        line = {
            "metric": "trellis_cells_per_sec", "value": cells * args.steps / dt, "unit": "cells/s",
            "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32 scores (+f64 add per cell, rounded to f32 as the reference does)",
            "data": "synthetic",
            "config": {"workload": (f"FLASH-BS dynamic-beam decode K={K} T={T} M={M_SYMBOLS} beam={beam} prob={PROB} seed={SEED} n_split={N_SPLIT} mode=reference ({w['label']})"
                                    if is_beam else
                                    f"FLASH Viterbi full-state decode K={K} T={T} M={M_SYMBOLS} prob={PROB} seed={SEED} n_split={N_SPLIT} mode=reference ({w['label']})"),
                       "cells_per_decode": cells,
                       "beam_cells_per_decode": K * beam * T if is_beam else None,
                       "beam_cells_per_sec": (K * beam * T * args.steps / dt) if is_beam else None,
                       "model_generator": "generate_data random stream (data_script.py -s 12)" if w["gen"] == "data_script"
                                          else "data_script.make_model32_fast (same distributions, vectorised random stream)",
                       "kernel": ("beam_step / beam_step_q16 by launch size" if is_beam else
                                  {1: "f64_stream", 2: "f32_refine", 3: "f16_refine", 4: "q16_refine", 5: "sparse_q16",
                                   6: "u16_refine (packed 16-bit filter: single-task launches of the whole-sequence pass; right-hand generations as batches of four on up to three streams; f32 filter for generations of <= 4 passes; same 16-bit table)"}[st["kernel"]]),
                       "kernel_note": (None if is_beam else "dense K*K sweep forced for value/roofline; the library's AUTO choice for this "
                                       "model is the sparse walk, reported separately as sparse_walk"),
                       "transition_density": st["density"] if not is_beam else None,
                       "passes": st["passes"], "step_launches": st["step_launches"], "task_steps": st["task_steps"],
                       "exact_heap_replays": st["beam_exact_sets"] if is_beam else None,
                       "speculative_duplicate_steps": st["beam_spec_steps"] if is_beam else None,
                       "reach_events": st["beam_reach_events"] if is_beam else None,
                       "chain_cuts": st["beam_chain_cuts"] if is_beam else None,
                       "rc": int(rc),
                       "merged_path_equal_to_single_rank": (None if single_rank_path is None else bool(np.asarray(path).tolist() == single_rank_path)),
                       "parallelism": f"segments over {args.gpus} rank(s)", "gather": gather_mode if dist is not None else "none",
                       "multi_gpu_status": "RCCL all-gather path unverified on hardware until a driver SCALE run exists" if args.gpus == 1 else None},
            "decode_ms": 1e3 * dt / args.steps,
            "forward_pass_ms": top_ms / args.steps,
            "forward_pass_cells_per_sec": K * (beam if is_beam else K) * (T - 1) / (1e-3 * top_ms / args.steps),
            "roofline": roofline,
        }
        line.update(extra)
        if args.gpus == 1 and args.workload == "cfg2":
            # extra: the FLASH-BS variant on the same model (never part of `value`)
            BEAM = 256
            fv.decode_beam(ob, N_SPLIT, BEAM)
            tb = time.perf_counter()
            nbs = 5
            for _ in range(nbs):
                _, _, brc = fv.decode_beam(ob, N_SPLIT, BEAM)
            bdt = (time.perf_counter() - tb) / nbs
            bst = fv.stats()
            line["flash_bs"] = {"workload": f"FLASH-BS K={K} T={T} n_split={N_SPLIT} beam={BEAM}, same model", "decode_ms": 1e3 * bdt,
                                "beam_cells_per_sec": K * BEAM * T / bdt, "exact_heap_replays": bst["beam_exact_sets"],
                                "speculative_steps": bst["beam_spec_steps"], "rc": int(brc)}
            if not args.no_cpu_baseline:
                try:      # the reference FLASH-BS program on the same full sequence, one run
                    bt, bpath = run_reference("flashbs", K, T, N_SPLIT, BEAM, model64, ob, runs=1, warm=0)
                    hp, _, _ = fv.decode_beam(ob, N_SPLIT, BEAM)
                    line["flash_bs"]["cpu_baseline"] = {"kind": "reference", "seconds": bt[0], "threads": N_SPLIT, "beam_cells_per_sec": K * BEAM * T / bt[0],
                                                        "sample": f"K={K} T={T} (the full sequence) beam={BEAM}, MAX_THREADS={N_SPLIT}; reference src/FLASH_BS_Viterbi_multithread.c, its own `time:` line",
                                                        "path_equal_to_hip": bool(bpath == hp.tolist())}
                except (FileNotFoundError, OSError, RuntimeError) as e:
                    line["flash_bs"]["cpu_baseline"] = f"not run: {type(e).__name__}"
        if args.gpus == 1 and not args.no_cpu_baseline:
            if is_beam:
                line["cpu_baseline"] = cpu_baseline_beam(w, model64, ob, fv, (A, B, Pi))
            elif model64 is None:
                line["cpu_baseline"] = cpu_baseline_port(w, model64, ob, fv, (A, B, Pi))
            else:
                line["cpu_baseline"] = cpu_baseline_full(w, model64, ob, fv)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    fv.close()


if __name__ == "__main__":
    main()
