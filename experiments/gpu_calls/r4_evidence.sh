#!/bin/bash
# Round 4: the evidence on the binary that ships.  Usage: r4_evidence.sh [configs...] (default: lzma64k dict8m xz)
#  * rocprofv3 --kernel-trace --stats of the default bench command                          -> gpurun_out/r4_ev/trace
#  * PMC passes, one counter set per pass, each under its own timeout and repeated if it hangs -> profiles/r04_pmc_<config>.json
#      1 FETCH_SIZE   2 WRITE_SIZE   3 SQ instruction counts + WAVE_CYCLES + WAIT_ANY   4 (lzma64k only) the wait split:
#      SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_* (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFGS=${@:-lzma64k dict8m xz}
O=gpurun_out/r4_ev; rm -rf $O; mkdir -p $O
pass() {  # cfg index counters...
  cfg=$1; i=$2; shift 2
  for attempt in 1 2 3; do
    rm -rf $O/$cfg/pass_$i
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/${cfg}_pass_$i.log 2>&1
    rc=$?
    n=$(find $O/$cfg/pass_$i -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "$cfg pass $i attempt $attempt rc=$rc csv=$n"
    [ "$n" -gt 0 ] && break
  done
}
for cfg in $CFGS; do
  pass $cfg 1 FETCH_SIZE
  pass $cfg 2 WRITE_SIZE
  pass $cfg 3 SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
  [ $cfg = lzma64k ] && pass $cfg 4 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_BRANCH
  python tools/make_pmc_profile.py $cfg $O/$cfg $O/r04_pmc_$cfg.json > /dev/null 2>$O/make_$cfg.err && cp $O/r04_pmc_$cfg.json profiles/
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $O/trace_bench.json 2>$O/trace_bench.err
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r04_kernel_trace_stats.csv \;
rm -rf $O/*/pass_*/*/*.db $O/trace/*/*.db 2>/dev/null
find $O -name "*.csv" -size +2M -delete
tail -1 $O/trace_bench.json | cut -c1-400
