#!/bin/bash
# Round 5, second call: (1) the price list of the loop's instruction forms (pipe_peaks, second batch), (2) first kernel A/B of the round --
# forms that move a decision's work between the vector and the scalar pipe: s1 = single decisions all scalar,
# forma2 = tree walks in form A with shadows, r11s = range >> 11 scalar, fball = form B everywhere, k24s = 2^24 in an SGPR, sh2 / sh6 =
# two / six queued instructions per shadow --, (3) the GPU suite on the host library with ADVICE r4's fixes (slab stride, slab init per
# launch, RESUME validation, waves released on failed streamed launches) incl. the new mixed-literal-row-class test.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_ab1; rm -rf $O; mkdir -p $O
timeout 200 build/pipe_peaks --waves 4 > $O/pipe_peaks_w4.txt 2>&1; echo "rc=$?" >> $O/pipe_peaks_w4.txt
timeout 200 build/pipe_peaks --waves 8 > $O/pipe_peaks_w8.txt 2>&1
cat $O/pipe_peaks_w4.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_literal_row or growable or wrong_guesses or lclp_above" 2>&1 | tail -15 | tee $O/new_tests.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in s1 forma2 k24s r11s fball sh2 sh6; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
