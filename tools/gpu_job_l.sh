#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x > gpurun_out/pytest_l.log 2>&1
rc=$?; echo "pytest rc=$rc"; tail -30 gpurun_out/pytest_l.log
