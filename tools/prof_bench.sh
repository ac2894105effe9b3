#!/bin/bash
# Regenerates the bench artefacts under gpurun_out/ (tools/pmc_summary.py then writes the summaries into profiles/):
#   bench line (default workload), rocprofv3 kernel stats (CSV), three PMC passes over the default workload (FETCH_SIZE;
#   WRITE_SIZE; TCC_HIT_sum + TCC_MISS_sum: separate passes, no trace flags).  FLASH-BS workloads: tools/prof_beam_pmc.sh.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_bench gpurun_out/pmc1 gpurun_out/pmc2 gpurun_out/pmc3
timeout -k 10 500 python3 bench.py > gpurun_out/bench_n1.json 2> gpurun_out/bench_n1.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bench -o bench -- python3 bench.py --steps 10 --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc1 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc2 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmc3 -o pmc -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/pmc3.log 2>&1
