#!/usr/bin/env python3
"""profiles/gather_traffic.json from the PMC summaries of tools/profile_round.sh: HBM bytes per launch of the
gather path (gather kernel + look-up pre-pass + locality probe) = 2 x FETCH_SIZE + WRITE_SIZE, in KB, per
dispatch (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md).
    python tools/make_gather_traffic.py profiles/r02_bench_pmc_default_summary.txt [profiles/r02_bench_pmc_noprepass_summary.txt]"""
import json
import re
import sys


def per_dispatch(text, kernel_re, counter):
    tot = 0.0
    names = []
    for line in text.splitlines():
        m = re.match(r"(.*?) %s dispatches (\d+) sum (\S+) per_dispatch (\S+)" % counter, line)
        if m and re.search(kernel_re, m.group(1)):
            tot += float(m.group(4))
            names.append(m.group(1).strip())
    return tot, names


def launch_bytes(text):
    out = {}
    for key, rx in (("gather", r"gather_kernel<"), ("lookup", r"lookup(_rows)?_kernel<"), ("probe", r"probe_kernel<")):
        f, names = per_dispatch(text, rx, "FETCH_SIZE")
        w, _ = per_dispatch(text, rx, "WRITE_SIZE")
        out[key] = {"kernels": names, "FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "bytes": (2 * f + w) * 1024}
    out["total_bytes"] = sum(v["bytes"] for k, v in out.items() if isinstance(v, dict))
    r, _ = per_dispatch(text, r"gather_kernel<|lookup(_rows)?_kernel<|probe_kernel<", "TCC_EA0_RDREQ")
    out["TCC_EA0_RDREQ_x128B"] = r * 128
    return out


def main():
    d = launch_bytes(open(sys.argv[1]).read())
    j = {"source": sys.argv[1] + " (rocprofv3 --pmc, one counter group per pass, per dispatch of 4096 queries)",
         "index_genomes": 100000, "query_batch": 4096, "tile_genomes": 50048,
         "note": "gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes (MI355X_MICROARCH.md, HBM), so reads = 2 x FETCH_SIZE; "
                 "padded buckets, tiles striped in blocks of 32 genomes, queries in locality order, look-up pre-pass over packed "
                 "table rows (default)",
         "default": d, "traffic_bytes_per_launch": d["total_bytes"]}
    if len(sys.argv) > 2:
        n = launch_bytes(open(sys.argv[2]).read())
        j["lookups_inside_the_gather_kernel"] = n
    json.dump(j, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
