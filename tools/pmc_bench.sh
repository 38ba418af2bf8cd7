#!/bin/bash
# PMC passes over the default bench (run on the GPU box from the repo root):
#   tools/pmc_bench.sh <tag> [bench args...]
# One rocprofv3 run per counter group (no tracing flags next to --pmc); FETCH_SIZE and WRITE_SIZE
# in passes of their own (MI355X_MICROARCH.md, rocprofv3 PMC slots); TCC_EA0_RDREQ_DRAM / _WRREQ_DRAM beside the plain request
# counters: requests "destined for DRAM" (do they differ from all requests, i.e. do they leave the Infinity Cache's hits out?).  Output: per-kernel sums in
# gpurun_out/pmc_<tag>.summary.txt for the kernels of the query path.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=/tmp/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT $R/gpurun_out
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "FETCH_SIZE" "WRITE_SIZE" \
 "TCC_EA0_RDREQ TCC_EA0_RDREQ_128B TCC_HIT TCC_MISS" \
 "TCC_EA0_RDREQ_DRAM TCC_EA0_WRREQ_DRAM TCC_EA0_WRREQ TCC_EA0_WRREQ_64B" \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" \
 "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --no-legs --no-overlap --steps 2 "$@" > $OUT/g$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/g$i.log; exit 1; }
done
cd $R
python tools/prof_summary.py $OUT | grep -E "gather_kernel|lookup_kernel|lookup_rows_kernel|block_kernel|probe_kernel|order_kernel|sketch_kernel|hits_|^==" | sed 's/  */ /g' > $R/gpurun_out/pmc_$TAG.summary.txt
rm -rf $OUT
