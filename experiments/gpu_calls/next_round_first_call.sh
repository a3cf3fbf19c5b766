#!/bin/bash
# What round 3 ran out of GPU minutes for; meant to be the first call of the next round (about 10 minutes):
#  1. PMC passes of the CURRENT kernel source for configs[1] and configs[2] (each pass under its own timeout, up to 3 attempts: rocprofv3
#     --pmc hangs about every other time on this pool), profiles/r03_pmc_*.json rewritten for this hash (they carry an `also_valid_for`
#     attestation until then);
#  2. an alternating A/B of the model registers pinned to v20..v54 (MILZMA_GEN_PINV=20; one sample in round 3 said -0.9 %):
#  3. the same for the split literal table (MILZMA_GEN_LITSPLIT=1: two scalar shifts less per literal-row swap; bit-exact on the emulator,
#     -0.14 scalar instructions per byte on text, -1.74 on random data), on text and on random data.
#  4. FIVE waves per SIMD: the loop generated with all its registers below v96 and the kernel built for 5 waves per SIMD compiles to 96 VGPRs /
#     occupancy 5, 7 KiB LDS (round 3; bit-exact; 5120 streams in one round: 18.16 GB/s, +2.9 % chip throughput).  Here: the GPU suite on
#     it and the large-batch figures with the time-sliced launch on 5120 persistent waves.  Then the time-sliced launch runs 5120 persistent waves: the decision chain's micro-benchmark says +11 % for
#     batches of >= 5120 streams (nothing for 4096).  GPU suite on the variant first, then 4096 / 8192 / 32768 streams against the shipped library.
#     Build the variants first:  python3 tools/build_variants.py "pinv20:PINV=20" "litsplit:LITSPLIT=1" "w5:VBASE=40,PINV=1,VROW8=7,NOPB4=1:+-DMILZMA_WAVES_PER_SIMD=5"
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/next_pmc; rm -rf $O; mkdir -p $O
pass() {  # cfg index counters...
  cfg=$1; i=$2; shift 2
  for attempt in 1 2 3; do
    rm -rf $O/$cfg/pass_$i
    timeout 110 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$cfg/pass_$i -- python bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-verify --other-configs none > $O/${cfg}_pass_$i.log 2>&1
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
  python tools/make_pmc_profile.py $cfg $O/$cfg $O/r03_pmc_$cfg.json > /dev/null 2>&1 && cp $O/r03_pmc_$cfg.json profiles/
done
rm -rf $O/*/pass_*/*/*.db 2>/dev/null
V=lzma_rs_amd/variants/libmilzma_pinv20.so
[ -f $V ] && python experiments/ab_bench.py --steps 4 lzma_rs_amd/libmilzma.so $V lzma_rs_amd/libmilzma.so $V | tee gpurun_out/next_pinv_ab.txt
W=lzma_rs_amd/variants/libmilzma_litsplit.so
[ -f $W ] && python experiments/ab_bench.py --steps 4 lzma_rs_amd/libmilzma.so $W lzma_rs_amd/libmilzma.so $W | tee gpurun_out/next_litsplit_ab.txt
[ -f $W ] && python experiments/ab_bench.py --steps 3 --kind random lzma_rs_amd/libmilzma.so $W | tee -a gpurun_out/next_litsplit_ab.txt
X=$PWD/lzma_rs_amd/variants/libmilzma_w5.so
if [ -f $X ]; then
  MILZMA_LIB=$X timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
  for n in 4096 8192 32768; do
    for lib in lzma_rs_amd/libmilzma.so $X; do
      MILZMA_LIB=$PWD/${lib#$PWD/} timeout 300 python bench.py --streams $n --steps 2 --warmup 1 --no-cpu-baseline --other-configs none 2>/dev/null | python -c "
import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n streams', '$lib'.split('/')[-1], l['value'], 'GB/s', l['ms_per_step'], 'ms', l['bit_exact'])"
    done
  done | tee gpurun_out/next_w5.txt
fi
