#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/var.txt
for v in 0 0; do
  echo "=== variant $v" >> gpurun_out/var.txt
  NIQKI_GATHER_VARIANT=$v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-extra 2> gpurun_out/var_err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.2f  gather %.3f ms/launch  frac %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))
" >> gpurun_out/var.txt || { tail -5 gpurun_out/var_err.txt >> gpurun_out/var.txt; cat gpurun_out/var.txt; exit 1; }
done
cat gpurun_out/var.txt
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q > gpurun_out/pt.log 2>&1 || { tail -30 gpurun_out/pt.log; exit 1; }
tail -1 gpurun_out/pt.log
