#!/bin/bash
# the library with time-sliced launches: PMC passes of the final kernel source (one attempt each), kernel-trace stats, the default bench
# line, then what slicing buys -- stream counts that are not a whole number of chip-fulls, and the lc + lp = 4 class
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
G=gpurun_out/r3_final; O=$G/pmc; mkdir -p $G
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
  i=$((i+1))
  timeout 100 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/lzma64k/pass_$i -- python bench.py --config lzma64k --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O.pass_$i.log 2>&1
  echo "pass $i rc=$? csv=$(find $O/lzma64k/pass_$i -name '*counter_collection.csv' 2>/dev/null | wc -l)"
done
if [ "$(find $O/lzma64k -name '*counter_collection.csv' | wc -l)" -ge 3 ]; then
  python tools/make_pmc_profile.py lzma64k $O/lzma64k $G/r03_pmc_lzma64k.json > /dev/null 2>&1 && cp $G/r03_pmc_lzma64k.json profiles/ && echo "pmc profile written"
fi
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $G/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $G/trace.log 2>&1; echo "trace rc=$?"
find $G/trace -name "*kernel_stats.csv" -exec cp {} $G/r03_kernel_trace_stats.csv \; ; head -2 $G/r03_kernel_trace_stats.csv | cut -c1-200; rm -rf $G/trace/*/*.db
python bench.py > $G/bench_default.json 2> $G/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
l=json.loads(open("$G/bench_default.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"].get("traffic"), l["roofline_issue"]["frac"], {k:(v["value"], v["roofline"].get("traffic")) for k,v in l["other_configs"].items()})
PY
run() {  # label n env...
  label=$1; n=$2; shift 2
  env "$@" timeout 300 python bench.py --streams $n --steps 3 --warmup 1 --no-cpu-baseline --other-configs none > $G/s_$label.json 2> $G/s_$label.err
  python - <<PY
import json
try:
    l=json.loads(open("$G/s_$label.json").read().strip().splitlines()[-1]); print("$label", l["value"], "GB/s", l["ms_per_step"], "ms/step, kernel", l["roofline"]["kernel_ms"], "bit_exact", l.get("bit_exact"))
except Exception as e:
    print("$label failed", e)
PY
}
for n in 4100 5000 6144; do run plain_$n $n MILZMA_SLICE=0; run sliced_$n $n; done
for mode in 0 auto; do MILZMA_SLICE=$mode python experiments/ab_bench.py --steps 3 --props 4,0,2 lzma_rs_amd/libmilzma.so > $G/lc4_$mode.txt 2>&1; echo "lc4 4096 streams, MILZMA_SLICE=$mode: $(tail -1 $G/lc4_$mode.txt)"; done
