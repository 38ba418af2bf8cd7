#!/bin/bash
# The device split in two by compute-unit masks (run on the GPU box from the repo root): the default bench's step
# with the handle's stream on the low X compute units and the sketch lane on the others, and the serial step on X
# compute units alone (what the gather launch takes there).  Output: gpurun_out/cu_split.txt, one JSON line per run.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/cu_split.txt
mkdir -p $R/gpurun_out; : > $OUT
run() { echo "== $*" >> $OUT; timeout -k 10 200 python3 $R/bench.py --no-legs --no-pmc --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 | python3 -c '
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps({k:d.get(k) for k in ("value","ms_per_step","kernels","serial_step","kernels_beside_each_other")}))' >> $OUT || exit 1; }
run
for x in "$@"; do run --gather-cus $x || exit 1; done
for x in "$@"; do run --gather-cus $x --no-overlap || exit 1; done
run
