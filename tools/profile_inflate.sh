#!/bin/bash
# Kernel-trace summary of the device inflate (run on the GPU box from the repo root):  tools/profile_inflate.sh r05
# 1024 gzip -6 genome files with each file's whole window in LDS, 2048 with its last 8 KB there, the gzip levels, and
# the `niqki` program on a list of gzip files.
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r05}
OUT=$R/gpurun_out/prof_inflate_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for cfg in "1024 0" "2048 1"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_$1 -- python3 $R/tools/bench_inflate.py --files $1 --window $2 --distinct 8 --reps 3 > $OUT/rate_$1.json 2> $OUT/rate_$1.log
done
cd $R
{
  echo "== tools/bench_inflate.py under rocprofv3 --kernel-trace --stats: 5 Mbp FASTA files, gzip -6 (3.24 : 1), one launch each, 3 launches per run"
  for n in 1024 2048; do
    echo "-- $n files:"; grep files $OUT/rate_$n.json | cut -c1-400
    python3 tools/prof_summary.py $OUT/kt_$n | grep -i "kernel \|inflate\|== "
  done
  echo "== not traced: other gzip levels (1024 files, whole window) and file counts (window's last 8 KB)"
  for lv in 1 9; do timeout 200 python3 tools/bench_inflate.py --files 1024 --window 0 --level $lv --distinct 8 --reps 2 2>/dev/null | cut -c1-330; done
  for n in 1024 4096; do timeout 200 python3 tools/bench_inflate.py --files $n --window 1 --distinct 8 --reps 2 2>/dev/null | cut -c1-330; done
} > gpurun_out/${TAG}_inflate_kernel_trace_summary.txt
rm -rf $OUT
cat gpurun_out/${TAG}_inflate_kernel_trace_summary.txt | cut -c1-220
