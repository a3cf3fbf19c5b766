#!/bin/bash
# GPU session 1 (round 2): sensitivity of the kernel time to dead S / V / branch instructions, per-wave timing, clocks
mkdir -p gpurun_out/b1
V=lzma_rs_amd/variants
(while true; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor edge|fclk" | tr '\n' ' '; echo; sleep 1; done) > gpurun_out/b1/clocks.txt 2>&1 &
SMI=$!
python experiments/ab_bench.py --steps 3 $V/libmilzma_base.so $V/libmilzma_pads1.so $V/libmilzma_padv1.so $V/libmilzma_padb1.so $V/libmilzma_pads2.so $V/libmilzma_padv2.so $V/libmilzma_norms.so > gpurun_out/b1/ab.txt 2>&1
python experiments/ab_bench.py --steps 2 --wavetime $V/libmilzma_wt.so > gpurun_out/b1/wavetime.txt 2>&1
python experiments/ab_bench.py --steps 2 --streams 3072 $V/libmilzma_base.so > gpurun_out/b1/occ.txt 2>&1
python experiments/ab_bench.py --steps 2 --streams 2048 $V/libmilzma_base.so >> gpurun_out/b1/occ.txt 2>&1
python experiments/ab_bench.py --steps 2 --streams 5120 $V/libmilzma_base.so >> gpurun_out/b1/occ.txt 2>&1
kill $SMI
cat gpurun_out/b1/ab.txt gpurun_out/b1/wavetime.txt gpurun_out/b1/occ.txt
tail -3 gpurun_out/b1/clocks.txt
