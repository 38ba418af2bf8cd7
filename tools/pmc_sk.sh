#!/bin/bash
# usage: pmc_sk.sh <binary> <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
B=$R/$1; TAG=$2
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" ; do
  i=$((i+1))
  rm -rf /tmp/pmc_$TAG_$i
  timeout -k 10 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_${TAG}_$i -- $B 256 5000000 1 > /tmp/pmc_${TAG}_$i.log 2>&1 || { echo "pass $i failed"; tail -3 /tmp/pmc_${TAG}_$i.log; }
done
cd $R
for i in 1 2; do python tools/prof_summary.py /tmp/pmc_${TAG}_$i | grep -E "sketch_kernel" | sed 's/  */ /g' | sed 's/^.*sketch_kernel/sketch_kernel/' ; done > gpurun_out/pmc_$TAG.txt
