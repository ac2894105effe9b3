#!/bin/bash
# SQ counters of the step kernels of one cfg2 decode (three decodes, dense Q16 kernel): where do the batched (NB = 8)
# launches spend their cycles?  Two --pmc passes (8 SQ slots each); summaries by tools/pmc_table.py.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/pmcA gpurun_out/pmcB
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmcA -o pmc -- python3 tools/prof_batch.py ${FVK:-4} > gpurun_out/pmcA.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d gpurun_out/pmcB -o pmc -- python3 tools/prof_batch.py ${FVK:-4} > gpurun_out/pmcB.log 2>&1
