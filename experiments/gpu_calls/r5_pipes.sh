#!/bin/bash
# Round 5, first call: what binds the symbol loop, in hardware units.
#  1. build/pipe_peaks: wall-clock issue rates of the vector / scalar / branch pipes per SIMD (experiments/microbench/pipe_peaks.hip)
#  2. the same micro-benchmark under rocprofv3 --pmc: what SQ_ACTIVE_INST_VALU / SQ_INST_CYCLES_SALU count per instruction at the pipes' peaks
#  3. the shipped kernel under two new counter sets (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_INST_CYCLES_SALU, SQ_IFETCH ... ; the instruction cache)
#     on the bench batch cached by experiments/ab_bench.py (one generation per call, every pass under its own timeout)
#  4. dead-instruction probes on the shipped loop: one dead scalar / vector / never-taken branch per adaptive decision (3.07 per byte)
#  5. the whole GPU suite on the library as round 4 left it
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_pipes; rm -rf $O; mkdir -p $O
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 > $O/clocks_idle.txt
timeout 120 build/pipe_peaks > $O/pipe_peaks.txt 2>&1; echo "rc=$?" >> $O/pipe_peaks.txt
cat $O/pipe_peaks.txt
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/peaks_pmc_$i -- build/pipe_peaks --waves 4 --reps 1 --iters 2000 > $O/peaks_pmc_$i.log 2>&1
  echo "peaks pmc $i rc=$?"
done
# the bench batch, once (cached under /tmp for the passes below)
timeout 300 python experiments/ab_bench.py --worker --steps 2 | tee $O/ab_default.txt
pass() {  # index counters...
  i=$1; shift
  for attempt in 1 2; do
    rm -rf $O/kernel_pmc_$i
    timeout 100 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/kernel_pmc_$i -- python experiments/ab_bench.py --worker --steps 1 > $O/kernel_pmc_$i.log 2>&1
    rc=$?
    n=$(find $O/kernel_pmc_$i -name "*counter_collection.csv" 2>/dev/null | wc -l)
    echo "kernel pass $i attempt $attempt rc=$rc csv=$n"
    [ "$n" -gt 0 ] && break
  done
}
pass 5 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU
pass 6 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU
L=lzma_rs_amd/libmilzma.so
V=""
for v in pads1 padv1 padb1; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 600 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_pads.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/suite.txt
rm -rf $O/*/*/*.db 2>/dev/null
find $O -name "*.csv" -size +4M -delete
python - <<'EOF'
import csv, glob, collections, os
O = "gpurun_out/r5_pipes"
for d in sorted(glob.glob(O + "/*_pmc_*")):
    if not os.path.isdir(d):
        continue
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per = collections.OrderedDict()
        for r in csv.DictReader(open(path)):
            k = (r["Dispatch_Id"], r["Kernel_Name"][:60])
            per.setdefault(k, collections.OrderedDict())
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per[k]["_ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        with open(d + ".summary.txt", "w") as f:
            for k, v in per.items():
                f.write("%s %s %s\n" % (k[0], k[1], " ".join("%s=%.6g" % kv for kv in v.items())))
        print(open(d + ".summary.txt").read()[-3000:])
EOF
