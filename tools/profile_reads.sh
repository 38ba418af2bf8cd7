#!/bin/bash
# Kernel-trace summaries of the short-read path and of the record framing (run on the GPU box
# from the repo root):  tools/profile_reads.sh r01
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_reads_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/reads -- python3 $R/tools/bench_reads.py > $OUT/reads.json 2> $OUT/reads.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ingest -- python3 $R/tools/bench_ingest.py > $OUT/ingest.json 2> $OUT/ingest.log
cd $R
{ echo "== tools/bench_reads.py under rocprofv3 --kernel-trace --stats"; tail -1 $OUT/reads.json; python tools/prof_summary.py $OUT/reads; } > gpurun_out/${TAG}_reads_kernel_trace_summary.txt
{ echo "== tools/bench_ingest.py under rocprofv3 --kernel-trace --stats"; tail -1 $OUT/ingest.json; python tools/prof_summary.py $OUT/ingest; } > gpurun_out/${TAG}_ingest_kernel_trace_summary.txt
rm -rf $OUT
head -14 gpurun_out/${TAG}_reads_kernel_trace_summary.txt | cut -c1-180; head -12 gpurun_out/${TAG}_ingest_kernel_trace_summary.txt | cut -c1-180
