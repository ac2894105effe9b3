#!/bin/bash
# rocprofv3 kernel trace of one FLASH-BS configuration; summary printed by tools/prof_beam_summary.py
# usage (on the GPU box): tools/prof_beam.sh K T N B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
rm -rf gpurun_out/prof_beam
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_beam -o beam -- python3 tools/check_cfg.py beam "$1" "$2" "$3" "$4" --no-oracle > gpurun_out/prof_beam.log 2>&1
