#!/usr/bin/env python3
"""Where a gather workgroup's time goes (measurement build: make -C niqki_amd/csrc HIPFLAGS+=-DNQ_GATHER_CLOCK,
which adds a 100 MHz clock read of thread 0 at the phase boundaries of nq::gather_kernel).  Runs bench.py with the
given arguments, then condenses the clocks of the last gather launch: mean microseconds per phase, and per CU
the idle gap between one workgroup's end and the next one's start.
  python tools/gather_clock.py --shard-of 8 --no-cpu --no-extra --steps 3"""
import ctypes
import os
import runpy
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from niqki_amd import capi  # noqa: E402

L = ctypes.CDLL(capi._LIB)
buf = torch.zeros(1 << 20, dtype=torch.int64, device="cuda")
assert L.nq_debug_gather_clock(ctypes.c_void_p(buf.data_ptr())) == 0
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 64)
a = a[a[:, 0] != 0]
t = a[:, :9].astype(np.float64) / 100.0          # microseconds
names = ["entry->walk0", "walk0", "barrier0", "scan0(+zero)", "to walk1", "walk1", "barrier1", "scan1"]
print("workgroups %d" % len(a))
last = 8 if (a[:, 8] != 0).all() else 4
for k in range(last):
    d = t[:, k + 1] - t[:, k]
    print("  %-14s mean %7.2f us  p10 %7.2f  p90 %7.2f" % (names[k], d.mean(), np.percentile(d, 10), np.percentile(d, 90)))
for k, base in ((0, 16), (1, 32)):
    if last < 8 and k:
        break
    w = a[:, base:base + 16].astype(np.float64) / 100.0 - t[:, [1 + 4 * k]]
    srt = np.sort(w, axis=1)
    print("  walk%d per wave: first done %6.2f us, median %6.2f, last %6.2f (mean over workgroups); wave 0 %6.2f" % (
        k, srt[:, 0].mean(), srt[:, 8].mean(), srt[:, 15].mean(), w[:, 0].mean()))
    print("    by wave index: " + " ".join("%.1f" % x for x in w.mean(axis=0)))
tot = t[:, last] - t[:, 0]
print("  workgroup      mean %7.2f us" % tot.mean())
cu = a[:, 15].astype(np.uint64) & np.uint64(0xFFFFFFFF0000FF00)     # XCC id, SE / SH / CU id of HW_ID
gaps, spans = [], []
for c in np.unique(cu):
    m = np.nonzero(cu == c)[0]
    o = m[np.argsort(t[m, 0])]
    g = t[o[1:], 0] - t[o[:-1], last]
    gaps += list(g)
    spans.append((t[o[-1], last] - t[o[0], 0], len(o)))
gaps = np.array(gaps)
print("  CUs seen %d, workgroups per CU %.1f, gap between workgroups on a CU: mean %.2f us  p10 %.2f  p90 %.2f" % (
    len(spans), np.mean([n for _, n in spans]), gaps.mean(), np.percentile(gaps, 10), np.percentile(gaps, 90)))
print("  kernel span (first start to last end) %.1f us" % (t[:, last].max() - t[:, 0].min()))
