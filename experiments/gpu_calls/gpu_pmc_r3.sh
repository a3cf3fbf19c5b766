#!/bin/bash
# Evidence for the shipped kernel source: rocprofv3 kernel-trace stats + PMC passes (FETCH_SIZE | WRITE_SIZE | SQ set) of bench.py's own
# batches for configs[1] and configs[2] (512 distinct streams each, as benched).  --pmc passes carry --kernel-trace only.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_pmc
mkdir -p $O
for cfg in lzma64k dict8m; do
  B="python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none"
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY"; do
    i=$((i+1))
    timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- $B > $O/${cfg}_pass_$i.log 2>&1
    echo "$cfg pass $i ($set) rc=$?"
  done
  python tools/make_pmc_profile.py $cfg $O/$cfg $O/r03_pmc_$cfg.json > $O/${cfg}_summary.txt 2>&1; tail -c 600 $O/${cfg}_summary.txt
done
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $O/trace.log 2>&1
echo "trace rc=$?"
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_trace_stats.csv \;
head -4 $O/r03_kernel_trace_stats.csv
tail -c 400 $O/trace.log
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
du -sh $O
