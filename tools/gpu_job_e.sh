#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/pytest_e.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_e.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err || exit 1
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/bench_e.json"))
print("value %.0f ms/step %.2f gather ms/launch %.3f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]), j["kernels"], "kmers %.0f" % j["sketch_kernel"]["gkmers_per_s"], j["config"]["index_build_s"])
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 > /dev/null 2> /tmp/kt.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/kt | grep -E "gather_kernel|lookup_kernel|pad_fill|build_kernel|probe_kernel|order_kernel|sketch_kernel|hits_|^==" > gpurun_out/kt_e_summary.txt; cat gpurun_out/kt_e_summary.txt | cut -c1-170
