#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/b16
mkdir -p $O
timeout 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_WRITE_SIZE -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify > $O/pmc_WRITE_SIZE.log 2>&1
echo "pmc WRITE_SIZE rc=$?"
python - <<'PY'
import csv,glob
from collections import defaultdict
fs=glob.glob("gpurun_out/b16/pmc_WRITE_SIZE/*/*counter_collection.csv")
if fs:
    agg=defaultdict(float)
    for r in csv.DictReader(open(fs[0])):
        if "decode_fast_asm" in r["Kernel_Name"]: agg[r["Counter_Name"]]+=float(r["Counter_Value"])
    print("WRITE_SIZE", dict(agg))
else: print("WRITE_SIZE no csv")
PY
python bench.py --steps 5 --warmup 1 --pcie > $O/bench_lzma64k.json 2> $O/bench_lzma64k.err; tail -c 600 $O/bench_lzma64k.json
python bench.py --steps 3 --warmup 1 --config dict8m --no-cpu-baseline > $O/bench_dict8m.json 2> $O/bench_dict8m.err; head -c 300 $O/bench_dict8m.json; echo
python bench.py --steps 3 --warmup 1 --config xz --no-cpu-baseline > $O/bench_xz.json 2> $O/bench_xz.err; head -c 300 $O/bench_xz.json; echo
