#!/bin/bash
# SQ counters of the sketch kernel (run on the GPU box from the repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-sk}
OUT=$R/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
i=0
for grp in \
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
 "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/g$i -- python3 $R/bench.py --no-cpu --steps 2 --genomes 20000 > $OUT/g$i.log 2>&1
done
cd $R
python tools/prof_summary.py $OUT | grep -E "sketch_kernel|^==" > $R/gpurun_out/pmc_$TAG.summary.txt
rm -rf $OUT
cat $R/gpurun_out/pmc_$TAG.summary.txt
