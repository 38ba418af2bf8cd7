#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "prepass_with_table or more_than_four or north_star" > gpurun_out/pp_tests.txt 2>&1 || { tail -40 gpurun_out/pp_tests.txt; exit 1; }
tail -1 gpurun_out/pp_tests.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra 2> gpurun_out/ab_err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f  gather %.3f ms/launch  frac %.3f parity %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['cpu_baseline']['parity']))
"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/kt2
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt2 -- python3 $R/bench.py --no-cpu --no-extra --steps 6 --warmup 2 > /dev/null 2> /tmp/kt2.log || { tail -5 /tmp/kt2.log; exit 1; }
cd $R; python3 tools/prof_summary.py /tmp/kt2 | grep -E "gather_kernel|lookup|probe_kernel|order_kernel" | cut -c1-170
