#!/bin/bash
# round 3, call 9: evidence for the shipped kernel source (PMC passes, kernel-trace stats) and the round's bench lines
bash experiments/gpu_calls/gpu_pmc_r3.sh
O=gpurun_out/r3_09
mkdir -p $O
cp gpurun_out/r3_pmc/r03_pmc_lzma64k.json gpurun_out/r3_pmc/r03_pmc_dict8m.json profiles/ 2>/dev/null
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"; tail -2 $O/bench_default.err
python bench.py --distinct 0 --no-cpu-baseline --other-configs none > $O/bench_distinct0.json 2> $O/bench_distinct0.err; echo "distinct0 rc=$?"
python bench.py --pcie --no-cpu-baseline --other-configs none > $O/bench_pcie.json 2> $O/bench_pcie.err; echo "pcie rc=$?"
python bench.py --gpus 1 --inproc --no-cpu-baseline > $O/bench_inproc.json 2> $O/bench_inproc.err; echo "inproc rc=$?"
for f in default distinct0 pcie inproc; do python - <<PY
import json
try:
    l=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print("$f", l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"].get("traffic"), l.get("pcie_inclusive",{}).get("value"), {k:v["value"] for k,v in l.get("other_configs",{}).items()})
except Exception as e: print("$f", "ERR", e)
PY
done
