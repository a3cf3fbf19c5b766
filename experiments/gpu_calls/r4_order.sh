#!/bin/bash
# Round 4: micro-benchmark of instruction orders / forms of a tree decision (experiments/microbench/order_slots.hip, built into build/ before the call).
# $@: nothing = first batch (orders), one argument = second batch (forms), two = third batch (normalisation stub)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4_order; mkdir -p $O
timeout 300 build/order_slots "$@" > $O/order_slots$#.txt 2>&1; echo "rc=$?" >> $O/order_slots$#.txt
cat $O/order_slots$#.txt
