#!/usr/bin/env python3
"""Writes gpurun_out/ours_cli_dump.gz on a GPU box: the dump file the `niqki` host program
produces for the CLI golden inputs (-I fof.txt -S 10 -J 0.1 -D ...), several gzip members as
ParallelGzWriter emits them.  Copy it to tests/golden/ours_cli_dump.gz: the CPU suite lets the
real reference binary load it (tests/test_oracle_golden.py), the GPU suite checks that the
program still writes the same payload (tests/test_cli_gpu.py)."""
import json
import os
import pathlib
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import niqki_amd  # noqa: E402
from conftest import make_cli_workdir  # noqa: E402

meta = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_meta.json")))
out = os.path.join(ROOT, "gpurun_out", "ours_cli_dump.gz")
os.makedirs(os.path.dirname(out), exist_ok=True)
with tempfile.TemporaryDirectory() as td:
    td = make_cli_workdir(pathlib.Path(td), niqki_amd, meta)
    subprocess.check_call([os.path.join(ROOT, "niqki_amd", "bin", "niqki"), "-I", "fof.txt", "-S", "10", "-J", "0.1",
                           "-O", "tmp.gz", "-D", out], cwd=td)
print(out, os.path.getsize(out), "bytes")
