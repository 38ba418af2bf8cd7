cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt5; timeout -k 10 800 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt5 -- python3 $R/bench.py --genomes 500000 --no-legs --steps 4 --warmup 1 > $R/gpurun_out/r05_500k_trace_$1.json 2> /tmp/kt5.log || { tail -5 /tmp/kt5.log; exit 1; }
cd $R; python3 tools/prof_summary.py /tmp/kt5 > gpurun_out/r05_500k_kernel_trace_$1.txt; head -16 gpurun_out/r05_500k_kernel_trace_$1.txt | cut -c1-190
