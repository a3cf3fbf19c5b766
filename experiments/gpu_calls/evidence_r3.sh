#!/bin/bash
# evidence for the FINAL kernel source (biased window offset + end-aligned last window): PMC passes (each under
# its own timeout, up to 2 attempts: rocprofv3 --pmc hangs now and then on this pool), kernel-trace stats, then the bench lines.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3_pmc
rm -rf $O; mkdir -p $O gpurun_out/r3_evidence
pass() {  # cfg index counters...
  cfg=$1; i=$2; shift 2
  for attempt in 1 2; do
    rm -rf $O/$cfg/pass_$i
    timeout 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/${cfg}_pass_$i.log 2>&1
    rc=$?
    n=$(find $O/$cfg/pass_$i -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "$cfg pass $i attempt $attempt rc=$rc csv=$n"
    [ "$n" -gt 0 ] && break
  done
}
for cfg in lzma64k dict8m; do
  pass $cfg 1 FETCH_SIZE
  pass $cfg 2 WRITE_SIZE
  pass $cfg 3 SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY
  python tools/make_pmc_profile.py $cfg $O/$cfg $O/r03_pmc_$cfg.json > $O/${cfg}_summary.txt 2>&1; tail -c 300 $O/${cfg}_summary.txt; cp $O/r03_pmc_$cfg.json profiles/
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --other-configs none > $O/trace.log 2>&1
echo "trace rc=$?"
find $O/trace -name "*kernel_stats.csv" -exec cp {} $O/r03_kernel_trace_stats.csv \;
head -3 $O/r03_kernel_trace_stats.csv
rm -rf $O/*/pass_*/*/*.db $O/trace/*/*.db 2>/dev/null
G=gpurun_out/r3_evidence
( time python bench.py ) > $G/bench_default.json 2> $G/bench_default.err; echo "default rc=$?"
python bench.py --distinct 0 --other-configs none --no-cpu-baseline > $G/bench_distinct0.json 2> $G/bench_distinct0.err; echo "distinct0 rc=$?"
python bench.py --pcie --other-configs none --no-cpu-baseline > $G/bench_pcie.json 2> $G/bench_pcie.err; echo "pcie rc=$?"
python bench.py --gpus 1 --inproc --other-configs none --no-cpu-baseline > $G/bench_inproc.json 2> $G/bench_inproc.err; echo "inproc rc=$?"
python - <<PY
import json
for n in ("default","distinct0","pcie","inproc"):
    try:
        l=json.loads(open("gpurun_out/r3_evidence/bench_%s.json"%n).read().strip().splitlines()[-1])
        print(n, l["value"], l["ms_per_step"], l["roofline"].get("kernel_ms"), l["roofline"].get("traffic"), (l.get("roofline_issue") or {}).get("frac"), l.get("pcie_inclusive"), {k:(v["value"], v["roofline"].get("traffic"), (v.get("roofline_issue") or {}).get("frac")) for k,v in l.get("other_configs",{}).items()})
    except Exception as e:
        print(n, "failed", e)
PY
du -sh gpurun_out
