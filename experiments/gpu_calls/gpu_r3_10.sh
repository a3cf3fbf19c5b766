#!/bin/bash
# round 3, call 10: the two PMC passes that hung in call 9 (rocprofv3 --pmc is flaky on this pool: each pass under its own short
# timeout, up to three attempts), then the round's default bench line with the recorded traffic in place
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_pmc
mkdir -p $O
pass() {  # cfg index counters...
  cfg=$1; i=$2; shift 2
  for attempt in 1 2 3; do
    rm -rf $O/$cfg/pass_$i
    timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/${cfg}_pass_$i.log 2>&1
    rc=$?
    n=$(find $O/$cfg/pass_$i -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "$cfg pass $i attempt $attempt rc=$rc csv=$n"
    [ "$n" -gt 0 ] && break
  done
}
pass lzma64k 3 SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
pass dict8m 1 FETCH_SIZE
for cfg in lzma64k dict8m; do python tools/make_pmc_profile.py $cfg $O/$cfg $O/r03_pmc_$cfg.json > $O/${cfg}_summary.txt 2>&1; tail -c 300 $O/${cfg}_summary.txt; cp $O/r03_pmc_$cfg.json profiles/; done
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
mkdir -p gpurun_out/r3_10
( time python bench.py ) > gpurun_out/r3_10/bench_default.json 2> gpurun_out/r3_10/bench_default.err; echo "default rc=$?"; tail -2 gpurun_out/r3_10/bench_default.err
python - <<PY
import json
l=json.loads(open("gpurun_out/r3_10/bench_default.json").read().strip().splitlines()[-1])
print(l["value"], l["ms_per_step"], l["roofline"]["kernel_ms"], l["roofline"].get("traffic"), l.get("roofline_issue",{}).get("frac"), {k:(v["value"], v["roofline"].get("traffic"), (v.get("roofline_issue") or {}).get("frac")) for k,v in l.get("other_configs",{}).items()})
PY
