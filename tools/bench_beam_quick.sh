#!/bin/bash
# beam parity tests + the two FLASH-BS bench lines (no CPU baseline): the quick loop for work on the beam kernels
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_beam.py -m gpu -x -q 2>&1 | tail -5 &&
timeout -k 10 200 python bench.py --workload cfg4 --no-cpu-baseline > gpurun_out/b4.json &&
timeout -k 10 200 python bench.py --workload cfg5 --no-cpu-baseline > gpurun_out/b5.json &&
python - <<'PY'
import json
for f in ("b4", "b5"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, "ms_per_step %.3f" % d["ms_per_step"], "whole-sequence pass %.3f ms" % d["forward_pass_ms"], "us/step %.2f" % d["roofline"]["launch_us"])
PY
