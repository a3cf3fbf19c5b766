#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/b19
mkdir -p $O
BENCH="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -- $BENCH > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/trace.log 2>&1
echo "trace rc=$?"
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
python - <<'PY'
import csv,glob
from collections import defaultdict
for c in ("FETCH_SIZE","WRITE_SIZE"):
    fs=glob.glob("gpurun_out/b19/pmc_%s/*/*counter_collection.csv"%c)
    if not fs: print(c,"no csv"); continue
    agg=defaultdict(float); dur=0
    for r in csv.DictReader(open(fs[0])):
        if "decode_fast_asm" not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]]+=float(r["Counter_Value"]); dur=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
    print(c, dict(agg), dur)
PY
head -2 $O/kernel_stats.csv
