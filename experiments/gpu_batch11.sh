#!/bin/bash
mkdir -p gpurun_out/b11
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > gpurun_out/b11/gputests.txt 2>&1
tail -6 gpurun_out/b11/gputests.txt
python bench.py --steps 3 --warmup 1 --pcie --no-cpu-baseline > gpurun_out/b11/bench_pcie.json 2> gpurun_out/b11/bench_pcie.err; python - <<'PY'
import json
l=json.loads(open("gpurun_out/b11/bench_pcie.json").read().strip().splitlines()[-1])
print(l["value"], l["pcie_inclusive"])
PY
tail -3 gpurun_out/b11/bench_pcie.err
python __graft_entry__.py smoke 2>&1 | tail -2
