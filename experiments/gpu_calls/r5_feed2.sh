#!/bin/bash
# Round 5: fed input: GPU tests; then what the loop's FEED test costs -- `nofeed` is the loop without it (the kernel of profiles/r05_bench_default.json),
# p0 .. p7 the shipped loop at every phase of its code in the fetch lines (p6 = the shipped library)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r5_feed2; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_feed.py -q 2>&1 | tail -25 | tee $O/feed.txt
L=lzma_rs_amd/libmilzma.so
V=""
for v in nofeed p0 p1 p2 p3 p4 p5 p7; do V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
timeout 1500 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab_text.txt
