#!/bin/bash
# PMC passes over one FLASH-BS bench workload (separate passes, no trace flags), for tools/pmc_beam_summary.py.
# usage (GPU box): tools/prof_beam_pmc.sh cfg4|cfg5
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
W=$1
rm -rf gpurun_out/bp_${W}_1 gpurun_out/bp_${W}_2 gpurun_out/bp_${W}_3
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/bp_${W}_1 -o pmc -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bp_${W}_1.log 2>&1 &&
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/bp_${W}_2 -o pmc -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bp_${W}_2.log 2>&1 &&
timeout -k 10 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/bp_${W}_3 -o pmc -- python3 bench.py --workload $W --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/bp_${W}_3.log 2>&1
