#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/pytest_m.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -15 gpurun_out/pytest_m.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
