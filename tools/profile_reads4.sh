#!/bin/bash
# Kernel-trace summary of the short-read query path at configs[4]'s shape (run on the GPU box from the repo root):
#   tools/profile_reads4.sh r05
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_reads4_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reads4 -- python3 $R/tools/bench_reads4.py --reads 524288 > $OUT/reads4.json 2> $OUT/reads4.log
cd $R
{ echo "== tools/bench_reads4.py --reads 524288 under rocprofv3 --kernel-trace --stats (variants: counter rows, hit lists at 4 capacities; 2 x 8 batches each)"; grep variant $OUT/reads4.json | cut -c1-200; python tools/prof_summary.py $OUT/reads4; } > gpurun_out/${TAG}_reads4_kernel_trace_summary.txt
rm -rf $OUT
head -30 gpurun_out/${TAG}_reads4_kernel_trace_summary.txt | cut -c1-200
