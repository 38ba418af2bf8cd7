#!/bin/bash
# round-2 GPU job A: tests, dump fixture, sector microbenchmark (+ PMC), baseline bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_a.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_a.log
[ $rc -ge 100 ] && exit $rc
python tools/make_dump_fixture.py || exit 1
hipcc -O3 --offload-arch=gfx950 tools/ubench_sector.hip -o /tmp/ubench_sector 2>/dev/null || exit 1
timeout -k 10 120 /tmp/ubench_sector > gpurun_out/ubench_sector.txt 2>&1 || exit 1
cat gpurun_out/ubench_sector.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RD[A-Z0-9_]*\|TCC_EA0_WR[A-Z0-9_]*\|TCC_REQ[A-Z0-9_]*\|TCC_HIT[A-Z0-9_]*\|TCC_MISS[A-Z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/tcc_counters.txt
for c in FETCH_SIZE TCC_EA0_RDREQ_32B TCC_EA0_RDREQ TCC_EA0_RDREQ_128B; do
  timeout -k 10 120 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -- /tmp/ubench_sector > /dev/null 2>&1 || exit 1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  echo "== $c" >> $GRAFT_REPO_ROOT/gpurun_out/ubench_sector_pmc.txt
  python3 - "$f" >> $GRAFT_REPO_ROOT/gpurun_out/ubench_sector_pmc.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    print(r.get("Dispatch_Id"), r.get("Kernel_Name", "")[:30], r.get("Counter_Name"), r.get("Counter_Value"))
PY
done
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_a.json 2> gpurun_out/bench_a.err
echo "bench rc=$?"; cat gpurun_out/bench_a.json | cut -c1-1500
