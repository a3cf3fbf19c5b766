#!/bin/bash
# Round 6: the evidence on the kernel that ships.  Usage: r6_evidence.sh [configs...] (default: lzma64k dict8m xz unknown_size)
#  * PMC passes, one counter set per pass on the cached batch (MILZMA_BENCH_CACHE), each under its own timeout, repeated if rocprofv3 hangs
#      1 FETCH_SIZE   2 WRITE_SIZE   3 SQ instruction counts + WAVE_CYCLES + WAIT_ANY   (lzma64k only:) 4 the wait split   5 the pipes' counters
#      (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_INST_CYCLES_SALU, SQ_ACTIVE_INST_*)   6 the instruction cache
#  * rocprofv3 --kernel-trace --stats of the default bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
export MILZMA_BENCH_CACHE=/tmp/milzma_bench_cache
CFGS=${@:-lzma64k dict8m xz unknown_size}
O=gpurun_out/r6_ev; rm -rf $O; mkdir -p $O
# the batches, once (cached for every pass below)
for cfg in $CFGS; do
  a="--config $cfg"; [ $cfg = unknown_size ] && a="--unknown-size"
  timeout 300 python bench.py $a --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/warm_$cfg.json 2>$O/warm_$cfg.err
  tail -c 200 $O/warm_$cfg.json
done
pass() {  # cfg index counters...
  cfg=$1; i=$2; shift 2
  a="--config $cfg"; [ $cfg = unknown_size ] && a="--unknown-size"
  for attempt in 1 2 3; do
    rm -rf $O/$cfg/pass_$i
    timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- python bench.py $a --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/${cfg}_pass_$i.log 2>&1
    rc=$?
    n=$(find $O/$cfg/pass_$i -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "$cfg pass $i attempt $attempt rc=$rc csv=$n"
    [ "$n" -gt 0 ] && break
  done
}
for cfg in $CFGS; do
  pass $cfg 1 FETCH_SIZE
  pass $cfg 2 WRITE_SIZE
  if [ $cfg != unknown_size ]; then
    pass $cfg 3 SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
  fi
  if [ $cfg = lzma64k ]; then
    pass $cfg 4 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
    pass $cfg 5 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_MISC
    pass $cfg 6 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL
  fi
  python tools/make_pmc_profile.py $cfg $O/$cfg $O/r06_pmc_$cfg.json > /dev/null 2>$O/make_$cfg.err && cp $O/r06_pmc_$cfg.json profiles/
done
# the time-sliced kernel with every unit parked at every quantum (MILZMA_SLICE=2): what its spilled park / unpark code costs
#   -- scratch traffic as a share of the vector-memory instructions, and the cycles a ready instruction waits -- next to the ordinary launch's passes 3 / 4
if echo $CFGS | grep -q lzma64k; then
  for attempt in 1 2 3; do
    rm -rf $O/sliced/pass_1
    MILZMA_SLICE=2 timeout 100 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_FLAT --kernel-trace --output-format csv -d $O/sliced/pass_1 -- python bench.py --config lzma64k --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/sliced_pass_1.log 2>&1
    n=$(find $O/sliced/pass_1 -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "sliced pass attempt $attempt csv=$n"
    [ "$n" -gt 0 ] && break
  done
  python - <<'PY' > $O/r06_sliced_pmc.txt 2>&1
import csv, glob, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); ms = {}
for p in glob.glob("gpurun_out/r6_ev/sliced/pass_1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if "decode_fast_asm" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"].split("(")[0], r["Dispatch_Id"])
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ms[k] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("MILZMA_SLICE=2 (every fast launch time-sliced, every unit parked at every quantum), bench.py --config lzma64k --steps 1, one --pmc pass:")
for k, c in rows.items():
    print(k[0], "dispatch", k[1], "%.2f ms" % ms[k], {n: int(v) for n, v in sorted(c.items())})
PY
  cat $O/r06_sliced_pmc.txt | head -5
fi
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $O/trace_bench.json 2>$O/trace_bench.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_trace_stats.csv \;
rm -rf $O/*/pass_*/*/*.db $O/trace/*/*.db 2>/dev/null
find $O -name "*.csv" -size +2M -delete
tail -1 $O/trace_bench.json | cut -c1-300
ls $O/*.json
