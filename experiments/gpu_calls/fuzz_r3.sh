#!/bin/bash
# differential fuzz of the final library against the oracle (damaged / truncated / concatenated .lzma, LZMA2, .xz), both kernels
cd $GRAFT_REPO_ROOT
G=gpurun_out/r3_fuzz; mkdir -p $G
for seed in 31 32 33; do
  timeout 200 python experiments/parity_fuzz.py --seed $seed --rounds 6 > $G/fuzz_asm_$seed.txt 2>&1; echo "asm seed $seed rc=$?"; tail -2 $G/fuzz_asm_$seed.txt
done
timeout 200 python experiments/parity_fuzz.py --seed 41 --rounds 4 --kernel generic > $G/fuzz_generic_41.txt 2>&1; echo "generic rc=$?"; tail -2 $G/fuzz_generic_41.txt
