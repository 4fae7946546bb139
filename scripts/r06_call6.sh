#!/bin/bash
# Round 6, sixth GPU call: the product-free reproducer (scripts/ubench/lost_writes.hip) under the load that
# made the sweep fail -- eight processes on the GPU, 128 busy host threads on 16 cores -- and without it.
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r06_lost
mkdir -p $O
nproc > $O/host.txt
# control: two copies, idle host
for k in 1 2; do ( timeout 200 scripts/ubench/lost_writes 150 $k > $O/control_$k.txt 2>&1 ) & done; wait
tail -n 2 $O/control_*.txt
# load: 112 busy loops + eight copies
BURN=()
for i in $(seq 112); do ( while :; do :; done ) & BURN+=($!); done
for k in $(seq 1 8); do ( timeout 700 scripts/ubench/lost_writes 400 $((100 + k)) > $O/load_$k.txt 2>&1 ) & P[$k]=$!; done
for k in $(seq 1 8); do wait ${P[$k]}; done
kill "${BURN[@]}" 2>/dev/null
grep -h "iteration\|wrong cells\|damaged\|fault\|hip" $O/load_*.txt | cut -c1-300 | head -60
