#!/bin/bash
# bench lines of the other BASELINE workloads + rocprofv3 kernel stats of the two FLASH-BS ones (copy into profiles/).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_cfg4 gpurun_out/prof_cfg5
timeout -k 10 400 python3 bench.py --workload cfg3 > gpurun_out/bench_cfg3.json 2> gpurun_out/bench_cfg3.err &&
timeout -k 10 700 python3 bench.py --workload cfg4 > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err &&
timeout -k 10 500 python3 bench.py --workload cfg5 > gpurun_out/bench_cfg5.json 2> gpurun_out/bench_cfg5.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg4 -o beam -- python3 bench.py --workload cfg4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_cfg4.log 2>&1 &&
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg5 -o beam -- python3 bench.py --workload cfg5 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/prof_cfg5.log 2>&1
