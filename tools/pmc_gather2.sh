#!/bin/bash
# Second PMC set for the gather kernel: vector-memory pipeline (TA/TCP) pressure.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TOTAL_CACHE_ACCESSES" \
 "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_TCP_TA_DATA_STALL_CYCLES" \
 "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD SQ_LEVEL_WAVES SQ_WAVES SQ_BUSY_CU_CYCLES" \
 "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_TOTAL_WAVEFRONTS" \
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_TAG_STALL TCC_REQ" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --no-cpu --steps 2 "$@" > $OUT/g$i.log 2>&1
  tail -2 $OUT/g$i.log | cut -c1-200
done
cd $R
python tools/prof_summary.py $OUT | grep -E "gather_kernel|^==" > $R/gpurun_out/pmc_$TAG.summary.txt
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.summary.txt
