#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rc=0

[ $rc -ne 0 ] && exit $rc
for v in "" 400 100; do
  NIQKI_BENCH_QFAM=$v timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_c_$v.json 2> gpurun_out/bench_c_$v.err || exit 1
  python3 - "$v" <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_c_%s.json" % sys.argv[1]))
print("qfam", sys.argv[1] or "all", "value %.0f ms/step %.2f gather ms/launch %.3f frac %.3f T %.0f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"], j["roofline"]["gathered_ids_per_query"]), j["kernels"], "kmers %.0f" % j["sketch_kernel"]["gkmers_per_s"])
PY
done
