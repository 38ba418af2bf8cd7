#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_group.py -m gpu -q -x > gpurun_out/pytest_h.log 2>&1
rc=$?; echo "pytest group rc=$rc"; tail -30 gpurun_out/pytest_h.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/pytest_h2.log 2>&1
rc=$?; echo "pytest all rc=$rc"; tail -8 gpurun_out/pytest_h2.log
