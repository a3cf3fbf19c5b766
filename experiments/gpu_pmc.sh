#!/bin/bash
# PMC passes for the roofline evidence (separate passes per counter set; no trace domains besides --kernel-trace)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/pmc
mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_[A-Z0-9_]*\|TCC_EA_[A-Z0-9_]*\|FETCH_SIZE\|WRITE_SIZE\|TCC_BUBBLE[A-Z_]*\|TCC_HIT[_a-z]*\|TCC_MISS[_a-z]*" | sort -u | tr '\n' ' ' > $O/counters_available.txt
echo >> $O/counters_available.txt
hipcc --offload-arch=gfx950 -O2 experiments/pmc_calib.hip -o /tmp/pmc_calib || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/calib_$c -- /tmp/pmc_calib > $O/calib_$c.log 2>&1
  echo "calib $c rc=$?"
done
BENCH="python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-verify"
for cfg in lzma64k dict8m; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/${cfg}_$c -- $BENCH --config $cfg > $O/${cfg}_$c.log 2>&1
    echo "$cfg $c rc=$?"
  done
done
timeout 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_BRANCH SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/lzma64k_SQ -- $BENCH > $O/lzma64k_SQ.log 2>&1
echo "SQ rc=$?"
timeout 400 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $O/lzma64k_SQ2 -- $BENCH > $O/lzma64k_SQ2.log 2>&1
echo "SQ2 rc=$?"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lzma64k_trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/lzma64k_trace.log 2>&1
echo "trace rc=$?"
for d in $O/*/; do
  f=$(find $d -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py "" $f > ${d%/}.summary.json 2>/dev/null
done
find $O -name "*kernel_stats.csv" | head -3
find $O -name "*kernel_stats.csv" -exec cp {} $O/lzma64k_kernel_stats.csv \;
# keep only the small summaries (the raw csvs can be large)
find $O -name "*counter_collection.csv" -size +20M -delete
ls -la $O | head -40
