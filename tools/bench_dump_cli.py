#!/usr/bin/env python3
"""`niqki -I fof -D dump.gz` and `niqki -L dump.gz -Q fof` end to end: N synthetic genomes of --len bases as FASTA files
in the page cache, the dump written (parallel gzip members) and loaded by the host program.  Prints one JSON line."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "niqki_amd", "bin", "niqki")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=8192)
    ap.add_argument("--len", type=int, default=200_000)
    ap.add_argument("--dir", default="/dev/shm/niqki_dump_bench")
    ap.add_argument("--reference", action="store_true", help="also load the dump with the reference's own program (oracle/_ref/niqki_ref)")
    a = ap.parse_args()
    import niqki_amd
    shutil.rmtree(a.dir, ignore_errors=True)
    os.makedirs(a.dir)
    try:
        names = []
        for g in range(a.genomes):
            seq = niqki_amd.synth_genome_host(11, g // 16, g % 16, 0 if g % 16 == 0 else 20 + 40 * (g % 16), a.len)
            rows = np.frombuffer(seq[: len(seq) // 70 * 70], np.uint8).reshape(-1, 70)
            body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
            fn = os.path.join(a.dir, "g%05d.fa" % g)
            open(fn, "wb").write(b">g%05d\n" % g + body)
            names.append(fn)
        open(os.path.join(a.dir, "fof.txt"), "w").write("\n".join(names) + "\n")
        open(os.path.join(a.dir, "q.txt"), "w").write("\n".join(names[:64]) + "\n")
        env = dict(os.environ, NIQKI_HOST_TIMING="1")
        res = {"genomes": a.genomes, "len": a.len}

        def run(tag, binary, cli):
            t0 = time.time()
            r = subprocess.run([binary] + cli, cwd=a.dir, capture_output=True, text=True, timeout=3000, env=env)
            res[tag + "_s"] = round(time.time() - t0, 3)
            res.setdefault("timing", {})[tag] = [l for l in r.stderr.splitlines() if l.startswith("[niqki timing]")]
            if r.returncode != 0:
                print(r.stdout[-1500:], r.stderr[-1500:], file=sys.stderr)
                raise SystemExit(1)
        run("index_only", BIN, ["-I", "fof.txt", "-J", "0.1", "-O", "o0.gz"])
        run("index_dump", BIN, ["-I", "fof.txt", "-D", "dump.gz", "-J", "0.1", "-O", "o1.gz"])
        res["dump_file_GB"] = round(os.path.getsize(os.path.join(a.dir, "dump.gz")) / 1e9, 3)
        run("load_query", BIN, ["-L", "dump.gz", "-Q", "q.txt", "-J", "0.1", "-O", "o2.gz"])
        run("index_query", BIN, ["-I", "fof.txt", "-Q", "q.txt", "-J", "0.1", "-O", "o3.gz"])
        res["same_hits"] = open(os.path.join(a.dir, "o2.gz"), "rb").read() == open(os.path.join(a.dir, "o3.gz"), "rb").read()
        res["dump_s"] = round(res["index_dump_s"] - res["index_only_s"], 3)
        res["load_s"] = round(res["load_query_s"] - (res["index_query_s"] - res["index_only_s"]), 3)
        ref = os.path.join(ROOT, "oracle", "_ref", "niqki_ref")
        if a.reference and os.path.exists(ref):
            run("reference_loads_our_dump", ref, ["-L", "dump.gz", "-Q", "q.txt", "-J", "0.1", "-O", "o4.gz"])
        print(json.dumps(res))
    finally:
        shutil.rmtree(a.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
