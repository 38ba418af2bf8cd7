#!/bin/bash
# round-2 GPU job B: full GPU suite, bench A/B of the look-up pre-pass, kernel trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_b.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -8 gpurun_out/pytest_b.log
[ $rc -ge 100 ] && exit $rc
hipcc -O3 --offload-arch=gfx950 tools/ubench_sector.hip -o /tmp/ubench_sector 2>/dev/null && timeout -k 10 120 /tmp/ubench_sector > gpurun_out/ubench_sector.txt 2>&1
cat gpurun_out/ubench_sector.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_b_pre.json 2> gpurun_out/bench_b_pre.err || exit 1
NIQKI_LOOKUP_PREPASS=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_b_nopre.json 2> gpurun_out/bench_b_nopre.err || exit 1
python3 - <<'PY'
import json
for n in ("pre", "nopre"):
    j = json.load(open("gpurun_out/bench_b_%s.json" % n))
    print(n, "value %.0f ms/step %.2f gather ms/launch %.3f frac %.3f" % (j["value"], j["ms_per_step"], j["roofline"]["avg_launch_ms"], j["roofline"]["frac"]), j["kernels"])
PY
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --steps 5 > /dev/null 2> /tmp/kt.log || exit 1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/kt > gpurun_out/kt_b_summary.txt; head -16 gpurun_out/kt_b_summary.txt | cut -c1-180
