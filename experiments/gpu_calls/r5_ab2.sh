#!/bin/bash
# Round 5, third call: pipe_peaks third batch (v_cndmask on vcc behind a v_cmp, v_readlane with the lane in m0) + queued instructions per shadow
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab2; rm -rf $O; mkdir -p $O
for w in 1 4 8; do timeout 200 build/pipe_peaks --waves $w --only "v_c" > $O/peaks3_cnd_w$w.txt 2>&1; timeout 100 build/pipe_peaks --waves $w --only "readlane" >> $O/peaks3_cnd_w$w.txt 2>&1; timeout 100 build/pipe_peaks --waves $w --only "the same" >> $O/peaks3_cnd_w$w.txt 2>&1; done
cat $O/peaks3_cnd_w*.txt | grep -v "^#" | cut -c1-120
L=lzma_rs_amd/libmilzma.so
V=""
for v in sh5 sh6 sh8 sh12; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V $L $V | tee $O/ab_text.txt
