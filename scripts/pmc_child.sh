#!/bin/bash
# Counter passes over the histogram launches: bash scripts/pmc_child.sh <tag> <queries> [env...]
#   8M: bash scripts/pmc_child.sh r06pmc8 80000      1M: bash scripts/pmc_child.sh r06pmc1 10000
TAG=$1; Q=$2
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
rocprofv3 -L > $O/avail.txt 2>&1
B="python bench.py --no-extras --no-cpu-baseline --no-scoring --queries $Q --steps 5 --warmup 2"
pick() { for c in "$@"; do grep -qw "$c" $O/avail.txt && echo -n "$c "; done; }
P1=$(pick SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS)
P2=$(pick SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD)
P3=$(pick SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAIT_INST_VMEM SQ_INSTS_FLAT)
P4=$(pick TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum)
P5=$(pick TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum)
P6=$(pick TCC_REQ_sum TCC_READ_sum TCC_EA0_RD_UNCACHED_32B_sum TCC_TAG_STALL_sum)
P7=FETCH_SIZE
P8=WRITE_SIZE
i=0
# (round 6: the TCP_* pass aborts inside rocprofv3 on this image and the wrapped command then hangs -- left out,
# the TCC request counters with it; every pass under its own timeout)
for P in "$P1" "$P2" "$P3" "$P7" "$P8"; do
  i=$((i+1))
  [ -z "$P" ] && continue
  echo "pass $i: $P" >> $O/passes.txt
  timeout -k 10 ${PMC_TIMEOUT:-420} rocprofv3 --kernel-trace --pmc $P -d $O/p$i -o pmc --output-format csv -- $B > /dev/null 2> $O/p$i.err || echo "pass $i failed" >> $O/passes.txt
done
python scripts/pmc_child_table.py $O/table.md $O/p1 $O/p2 $O/p3 $O/p4 $O/p5 $O/p6 $O/p7 $O/p8 > /dev/null 2> $O/table.err
for d in $O/p?; do f=$(find $d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && gzip -c $f > $O/$(basename $d)_cc.csv.gz; rm -rf $d; done
head -c 6000 $O/table.md
