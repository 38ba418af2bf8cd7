#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_group.py -x -q > gpurun_out/pp_tests.txt 2>&1 || { tail -40 gpurun_out/pp_tests.txt; exit 1; }
tail -1 gpurun_out/pp_tests.txt
for i in 1 2 3; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-extra --no-cpu 2> gpurun_out/ab_err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.3f  gather %.3f ms/launch  frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
"
done
