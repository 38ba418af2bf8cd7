"""Rate of the device inflate alone (niqki_gunzip, nq_inflate.hip): N gzip'd synthetic FASTA genomes in one launch.
python tools/bench_inflate.py [--files 1024] [--len 5000000] [--distinct 16] [--level 6]"""
import argparse
import gzip
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import niqki_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=1024)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--distinct", type=int, default=16)
    ap.add_argument("--level", type=int, default=6)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--window", type=int, default=-1, help="option inflate_window: -1 by the number of files, 0 whole window in LDS, 1 its last 8 KB")
    a = ap.parse_args()
    e = niqki_amd.Engine(K=31, S=10, W=10, H=4)
    e.set_option("inflate_window", a.window)
    t0 = time.time()
    plain, zipped = [], []
    for i in range(a.distinct):
        g = niqki_amd.synth_genome_host(11, i, 0, 0, a.len)
        f = b">g%d\n" % i + b"\n".join(bytes(g[k:k + 70]) for k in range(0, len(g), 70)) + b"\n"
        plain.append(f)
        zipped.append(gzip.compress(f, compresslevel=a.level, mtime=0))
    t_make = time.time() - t0
    files = [zipped[i % a.distinct] for i in range(a.files)]
    sizes = [len(plain[i % a.distinct]) for i in range(a.files)]
    # host inflate of the same bytes, one thread (zlib through Python: the per-core rate the readers have)
    t0 = time.time()
    for z in zipped:
        gzip.decompress(z)
    t_cpu = (time.time() - t0) / a.distinct
    e.profile(True)
    out = None
    best = None
    for r in range(a.reps):
        e.profile_reset()
        out, status, produced, members, _ = e.gunzip(files, sizes, check_outside=False)
        ms, n = e.profile_read(niqki_amd.capi.KC_INFLATE)
        best = ms if best is None else min(best, ms)
    assert not status.any(), status
    assert all(out[i] == plain[i % a.distinct] for i in range(0, a.files, max(1, a.files // 37)))
    raw = float(sum(sizes))
    wire = float(sum(len(f) for f in files))
    print(json.dumps({"files": a.files, "inflate_window": a.window, "files_in_flight": [e.stat("inflate_files_in_flight"), e.stat("inflate_files_in_flight_8k")], "genome_bp": a.len, "gzip_level": a.level, "kernel_ms": round(best, 3),
                      "files_per_s": round(a.files / (best / 1e3), 1), "raw_GBps": round(raw / best / 1e6, 2),
                      "wire_GBps": round(wire / best / 1e6, 2), "ratio": round(raw / wire, 3),
                      "zlib_one_thread_files_per_s": round(1 / t_cpu, 2), "make_inputs_s": round(t_make, 1),
                      "per_file": {k: v // a.files for k, v in e.gunzip_stats().items()}}))
    e.close()


if __name__ == "__main__":
    main()
