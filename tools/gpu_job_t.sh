#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/ord.txt
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_group.py tests/test_gpu_fullsize.py tests/test_gpu_paging.py -x -q > gpurun_out/pt.log 2>&1 || { tail -40 gpurun_out/pt.log; exit 1; }
tail -1 gpurun_out/pt.log
for shape in "--steps 10 --warmup 3" "--shard-of 8 --batch 32768 --ring 2 --steps 4 --warmup 1" "--shard-of 8 --steps 10 --warmup 2" ; do
  echo "=== $shape" >> gpurun_out/ord.txt
  timeout -k 10 300 python bench.py $shape --no-cpu --no-extra 2> gpurun_out/ord_err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ms/step %.2f  value %.0f gather %.3f ms/launch x %d  frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['avg_launch_ms'], d['roofline']['launches'], d['roofline']['frac']))
" >> gpurun_out/ord.txt || { tail -5 gpurun_out/ord_err.txt >> gpurun_out/ord.txt; cat gpurun_out/ord.txt; exit 1; }
done
cat gpurun_out/ord.txt
