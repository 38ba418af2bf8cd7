#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in r1 10 r1 10; do
lib=$GRAFT_REPO_ROOT/niqki_amd/lib/ab/libniqki_$v.so
[ $v = 11 ] && lib=$GRAFT_REPO_ROOT/niqki_amd/lib/libniqki_hip.so
NIQKI_LIB=$lib NIQKI_LOOKUP_PREPASS=0 timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu > gpurun_out/bench_g$v.json 2> gpurun_out/bench_g$v.err || exit 1
python3 - $v <<'PY'
import json, sys
j = json.load(open("gpurun_out/bench_g%s.json" % sys.argv[1]))
print("roll32/masked", sys.argv[1], "value %.0f sketch ms %.3f kmers %.1f" % (j["value"], j["kernels"]["sketch"]["ms"] / 10, j["sketch_kernel"]["gkmers_per_s"]))
PY
done
