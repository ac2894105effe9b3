#!/bin/bash
# Counter calibration for the FLASH-BS gather patterns (tools/micro/gather_calib.hip): one rocprofv3 --pmc pass per counter
# set (no trace flags), summary by tools/gather_calib_summary.py.   usage (GPU box): tools/gather_calib.sh [K] [B]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
BIN=tools/micro/gather_calib.bin
[ -x $BIN ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $BIN tools/micro/gather_calib.hip || exit 1
rm -rf gpurun_out/gcal1 gpurun_out/gcal2 gpurun_out/gcal3
./$BIN "$@" > gpurun_out/gather_calib_times.txt &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/gcal1 -o pmc -- ./$BIN "$@" > gpurun_out/gcal1.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/gcal2 -o pmc -- ./$BIN "$@" > gpurun_out/gcal2.log 2>&1 &&
timeout -k 10 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/gcal3 -o pmc -- ./$BIN "$@" > gpurun_out/gcal3.log 2>&1 &&
python3 tools/gather_calib_summary.py "$@"
