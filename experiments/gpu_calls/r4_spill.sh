#!/bin/bash
# Round 4: lc + lp >= 4 in the asm loop (HBM variant, launch class kFastSpill).  GPU suite, then the property classes side by side.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_spill; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $O/suite.txt
( echo "## asm loop: lc3 = LP0 variant; everything else with lc + lp >= 4 = HBM variant (4096 x 1 MiB, greedy-parse streams of tests/lzma_enc.py)";
  timeout 900 python experiments/lclp_bench.py --streams 4096 --size 1048576 --distinct 64 3,0,2 4,0,2 0,4,0 2,2,4 8,0,2 4,4,0 5,2,4 8,4,4 2>/dev/null;
  echo "## generic kernel for lc + lp >= 4 (MILZMA_SPILL=generic: round 3's path)";
  MILZMA_SPILL=generic timeout 900 python experiments/lclp_bench.py --streams 4096 --size 1048576 --distinct 64 4,0,2 8,0,2 2>/dev/null ) | tee $O/lclp_classes.txt
