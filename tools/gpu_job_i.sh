#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python bench.py > gpurun_out/bench_i.json 2> gpurun_out/bench_i.err; rc=$?
echo "bench rc=$rc"; tail -5 gpurun_out/bench_i.err
[ $rc -ne 0 ] && exit $rc
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/bench_i.json"))
print("value %.0f ms/step %.2f gather %.3f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]))
print(json.dumps(j["sketch_kernel"]))
print(json.dumps(j["cpu_baseline"]))
print(json.dumps(j["extra_workloads"], indent=1)[:3500])
PY
timeout -k 10 600 python bench.py --shard-of 8 --no-cpu > gpurun_out/bench_i_shard8.json 2> gpurun_out/bench_i_shard8.err; rc=$?
echo "shard-of rc=$rc"; tail -3 gpurun_out/bench_i_shard8.err
[ $rc -ne 0 ] && exit $rc
python3 -c "
import json
j = json.load(open('gpurun_out/bench_i_shard8.json'))
print(j['value'], j['ms_per_step'], json.dumps(j['shard_emulation']), j['kernels'])"
NIQKI_FORCE_DIST=1 timeout -k 10 600 python bench.py --no-cpu --no-extra > gpurun_out/bench_i_dist1.json 2> gpurun_out/bench_i_dist1.err; rc=$?
echo "force-dist rc=$rc"; tail -3 gpurun_out/bench_i_dist1.err
python3 -c "
import json
j = json.load(open('gpurun_out/bench_i_dist1.json'))
print(j['value'], j['ms_per_step'], j['config']['parallelism'], j['kernels'], j['config']['hits_per_query'])"
