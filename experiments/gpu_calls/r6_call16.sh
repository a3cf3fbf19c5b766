#!/bin/bash
# Round 6: what the VDIRECT bit block costs next to the scalar one (experiments/microbench/pipe_peaks.hip, the two blocks as dependent chains, 1 .. 8 waves per SIMD)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r6_call16; rm -rf $O; mkdir -p $O
mkdir -p /tmp/mb
hipcc --offload-arch=gfx950 -O2 experiments/microbench/pipe_peaks.hip -o /tmp/mb/pipe_peaks 2>/dev/null || exit 1
for v in "v_sub_co_u32" "v_min_u32" "VDIRECT bit block" "scalar bit block" "v_lshrrev_b32" "s_addc_u32" "v_readfirstlane"; do
  timeout 300 /tmp/mb/pipe_peaks --only "$v" | grep -v "^#" | tee -a $O/pipe_peaks_vdirect.txt
done
