#!/bin/bash
# Round 4: (1) the floor -- the shipped kernel at 1 / 2 / 3 / 4 waves per SIMD (1024 .. 4096 streams of 1 MiB): the
# per-stream latency no scheduling can beat; (2) alternating A/Bs of the prepared variants (tools/build_variants.py) against the shipped library.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_floor; mkdir -p $O
L=lzma_rs_amd/libmilzma.so
for n in 1024 2048 3072 4096; do
  echo -n "streams $n: "; timeout 200 python experiments/ab_bench.py --steps 3 --streams $n $L
done | tee $O/floor.txt
V=""
for v in "$@"; do [ -f lzma_rs_amd/variants/libmilzma_$v.so ] && V="$V lzma_rs_amd/variants/libmilzma_$v.so"; done
if [ -n "$V" ]; then
  timeout 900 python experiments/ab_bench.py --steps 4 $L $V $L $V | tee $O/ab.txt
fi
