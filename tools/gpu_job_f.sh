#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 0 1; do
NIQKI_LOOKUP_PREPASS=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_f$v.json 2> gpurun_out/bench_f$v.err || exit 1
python3 - $v <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_f%s.json" % sys.argv[1]))
print("prepass", sys.argv[1], "value %.0f ms/step %.2f gather ms/launch %.3f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]), j["kernels"])
PY
done
