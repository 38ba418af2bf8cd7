#!/usr/bin/env python3
"""Where a SHORT-READ gather workgroup's time goes: tools/bench_reads4.py under a measurement build of the library
(nq_query.hip compiled with -DNQ_GATHER_CLOCK, linked into tools/bin/libniqki_hip_clk.so, named by NIQKI_EXP_LIB: a
100 MHz clock read of thread 0 at the phase boundaries of nq::gather_kernel).  Condenses the clocks of the LAST gather
launch (65 536 one-read workgroups): mean microseconds per phase.
    NIQKI_EXP_LIB=tools/bin/libniqki_hip_clk.so python tools/gather_clock_reads.py --reads 262144"""
import ctypes
import os
import runpy
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from niqki_amd import capi  # noqa: E402

capi._LIB = os.path.abspath(os.environ["NIQKI_EXP_LIB"])
L = ctypes.CDLL(capi._LIB)
buf = torch.zeros(65536 * 64, dtype=torch.int64, device="cuda")
assert L.nq_debug_gather_clock(ctypes.c_void_p(buf.data_ptr())) == 0
sys.argv = [os.path.join(ROOT, "tools", "bench_reads4.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 64)
a = a[a[:, 0] != 0]
t = a[:, :6].astype(np.float64) / 100.0          # microseconds
names = ["entry -> walk (counters zeroed)", "walk (mask tests, look-ups, buckets)", "barrier", "counter scan -> hit list", "rank + store"]
print("workgroups %d" % len(a))
for k in range(5):
    d = t[:, k + 1] - t[:, k]
    print("  %-42s mean %7.2f us  p10 %7.2f  p90 %7.2f" % (names[k], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
w = a[:, 16:20].astype(np.float64) / 100.0 - t[:, [1]]
print("  walk per wave (4 waves): first done %.2f us, last %.2f" % (np.sort(w, axis=1)[:, 0].mean(), np.sort(w, axis=1)[:, 3].mean()))
tot = t[:, 5] - t[:, 0]
print("  workgroup mean %.2f us; kernel span %.1f us" % (tot.mean(), t[:, 5].max() - t[:, 0].min()))
