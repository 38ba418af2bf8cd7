#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 > /dev/null 2> /tmp/kt.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/kt | grep -E "gather_kernel|lookup_kernel|block_kernel|probe_kernel|order_kernel|sketch_kernel|hits_|^==" > gpurun_out/kt_d_summary.txt; cat gpurun_out/kt_d_summary.txt | cut -c1-170
bash tools/pmc_bench.sh d || exit 1
cat gpurun_out/pmc_d.summary.txt | grep -E "gather_kernel|lookup|sketch_kernel<1024, 32" | cut -c1-200
