#!/bin/bash
# Round profile (run on the GPU box from the repo root):  tools/profile_round.sh r01
# 1) rocprofv3 --kernel-trace --stats of the default bench command
# 2) separate --pmc passes (no tracing flags) for the gather kernel's HBM traffic:
#    FETCH_SIZE, WRITE_SIZE, TCC_EA0_RDREQ_128B (MI355X_MICROARCH.md: on gfx950
#    FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled)
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
OUT=$R/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $R/bench.py --no-cpu > $OUT/kt.json 2> $OUT/kt.log
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_128B; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $OUT/$c -- python3 $R/bench.py --no-cpu --steps 2 > $OUT/$c.json 2> $OUT/$c.log
done
cd $R
python tools/prof_summary.py $OUT/kt > gpurun_out/${TAG}_kernel_trace_summary.txt
cp $(find $OUT/kt -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_rocprofv3_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_128B; do python tools/prof_summary.py $OUT/$c | grep -E "^==|gather_kernel|probe_kernel|order_kernel|sketch_kernel|hits_|build_kernel"; done > gpurun_out/${TAG}_pmc_summary.txt
cp $OUT/kt.json gpurun_out/${TAG}_bench_under_rocprof.json
rm -rf $OUT
cat gpurun_out/${TAG}_pmc_summary.txt | cut -c1-200; head -12 gpurun_out/${TAG}_kernel_trace_summary.txt | cut -c1-170
