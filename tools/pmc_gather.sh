#!/bin/bash
# PMC passes for the gather kernel (run on the GPU box from the repo root):
#   tools/pmc_gather.sh <tag> [bench args...]
# One rocprofv3 run per counter group (no tracing flags next to --pmc).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_VMEM" \
 "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" \
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_32B TCC_EA0_RDREQ_64B TCC_EA0_RDREQ_128B" \
 "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ_DRAM" \
 "GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --no-cpu --steps 2 "$@" > $OUT/g$i.log 2>&1
done
cd $R
python tools/prof_summary.py $OUT | grep -E "gather_kernel|^==" > $R/gpurun_out/pmc_$TAG.summary.txt
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.summary.txt
