#!/usr/bin/env python3
"""End-to-end timing of the `niqki` host program (niqki_amd/bin/niqki) on files:
N synthetic genomes written as FASTA (70 columns, optionally gzip level 1), then
  niqki -I fof -Q fof -J 0.1            (whole-file mode)
and, with --reads R, a FASTA of R 150-base reads through -l (lines mode) against
that index.  Prints one JSON line with wall times and rates.  File bytes are the
inputs of the measurement, so they are written to --dir first (default /dev/shm:
page cache speed, i.e. the storage is taken out of the picture)."""
import argparse
import gzip
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


def write_fasta(path, name, seq, gz, level=1):
    rows = np.frombuffer(seq[: len(seq) // 70 * 70], np.uint8).reshape(-1, 70)
    body = np.concatenate([rows, np.full((rows.shape[0], 1), 10, np.uint8)], axis=1).tobytes()
    tail = bytes(seq[len(seq) // 70 * 70:])
    data = b">" + name.encode() + b"\n" + body + (tail + b"\n" if tail else b"")
    if gz:
        with gzip.open(path, "wb", compresslevel=level) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
    return len(data)


def _write_one(job):
    g, fn, length, gz, level = job
    import niqki_amd
    seq = niqki_amd.synth_genome_host(11, g // 16, g % 16, 0 if g % 16 == 0 else 20 + 40 * (g % 16), length)
    return write_fasta(fn, "g%05d" % g, seq, gz, level)


def cpu_quota():
    try:
        threads = len(os.sched_getaffinity(0))
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            threads = max(1, min(threads, int(round(float(q[0]) / float(q[1])))))
        return threads
    except (OSError, ValueError, IndexError, AttributeError):
        return os.cpu_count() or 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genomes", type=int, default=512)
    ap.add_argument("--len", type=int, default=5_000_000)
    ap.add_argument("--reads", type=int, default=0)
    ap.add_argument("--gz", action="store_true")
    ap.add_argument("--gz-level", type=int, default=6, help="gzip level of the input files (gzip's own default is 6)")
    ap.add_argument("--host-inflate-too", action="store_true",
                    help="--gz: the index + query run once more with NIQKI_HOST_NO_GPU_INFLATE=1 (every file inflated by the reader threads)")
    ap.add_argument("--dir", default="/dev/shm/niqki_cli_bench")
    ap.add_argument("--extra", default="", help="extra CLI options, space separated")
    ap.add_argument("--reference", type=int, default=0,
                    help="also run the reference's own program (oracle/_ref/niqki_ref: CPU; niqki_ref_gpu: its operators bound to "
                         "the C ABI) on the first N files")
    args = ap.parse_args()
    import niqki_amd
    shutil.rmtree(args.dir, ignore_errors=True)
    os.makedirs(args.dir)
    try:
        names = [os.path.join(args.dir, "g%05d.fa%s" % (g, ".gz" if args.gz else "")) for g in range(args.genomes)]
        jobs = [(g, fn, args.len, args.gz, args.gz_level) for g, fn in enumerate(names)]
        if args.gz and args.genomes >= 32:   # (compressing is the slow part of making the inputs)
            import multiprocessing as mp
            with mp.get_context("fork").Pool(min(16, cpu_quota())) as pool:
                raw_bytes = sum(pool.map(_write_one, jobs, chunksize=4))
        else:
            raw_bytes = sum(_write_one(j) for j in jobs)
        open(os.path.join(args.dir, "fof.txt"), "w").write("\n".join(names) + "\n")
        extra = args.extra.split()
        res = {"genomes": args.genomes, "len": args.len, "gz": args.gz, "fasta_bytes": raw_bytes}
        if args.gz:
            res["gz_level"] = args.gz_level
            res["file_bytes"] = sum(os.path.getsize(n) for n in names)

        env = dict(os.environ, NIQKI_HOST_TIMING="1")
        phases = {}

        def run(tag, cli, more_env=None):
            t0 = time.time()
            r = subprocess.run([BIN] + cli + extra, cwd=args.dir, capture_output=True, text=True, timeout=1800,
                               env=dict(env, **(more_env or {})))
            dt = time.time() - t0
            # the program's own phase clocks: "[niqki timing] N files: total T s, ..." / "... lines mode: N entries in T s"
            phases[tag] = [float(x.split(" s")[0]) for line in r.stderr.splitlines() if line.startswith("[niqki timing]")
                           for x in [line.split("total ")[-1] if "total " in line else line.split(" in ")[-1]]]
            res.setdefault("timing_lines", {})[tag] = [l for l in r.stderr.splitlines() if l.startswith("[niqki timing]")]
            if r.returncode != 0:
                print(r.stdout[-2000:], r.stderr[-2000:], file=sys.stderr)
                raise SystemExit(1)
            res[tag + "_s"] = round(dt, 3)
            return dt
        # the first process on a fresh box pays the driver's cold start: not part of the measurement
        open(os.path.join(args.dir, "one.txt"), "w").write(names[0] + "\n")
        run("warmup", ["-I", "one.txt", "-J", "0.1", "-O", "o0.gz"])
        res["startup_s"] = round(run("startup", ["-I", "one.txt", "-J", "0.1", "-O", "o0.gz"]), 3)
        dt = run("index_only", ["-I", "fof.txt", "-J", "0.1", "-O", "o1.gz"])
        res["index_first_pass_genomes_per_s"] = round(args.genomes / phases["index_only"][0], 1)   # the first process to read the files
        run("index_query", ["-I", "fof.txt", "-Q", "fof.txt", "-J", "0.1", "-O", "o2.gz"])
        # rates from the program's phase clocks (process start-up is reported as startup_s)
        t_index, t_query = phases["index_query"][0], phases["index_query"][1]
        res["index_phase_s"], res["query_phase_s"] = t_index, t_query
        res["index_genomes_per_s"] = round(args.genomes / t_index, 1)
        res["index_fasta_GBps"] = round(raw_bytes / t_index / 1e9, 3)
        res["query_genomes_per_s"] = round(args.genomes / t_query, 1)
        # the query phase's own split: "(copy + frame X, sketch + insert/query Y, output Z)"
        import re
        m = re.search(r"\(copy \+ frame ([0-9.e+-]+), sketch \+ insert/query ([0-9.e+-]+), output ([0-9.e+-]+)\)",
                      res["timing_lines"]["index_query"][1])
        if m:
            cp, dv, ou = (float(x) for x in m.groups())
            # (copy_and_frame: what the main thread still waits for -- the copy of batch i+1 runs under batch i)
            res["query_phase_split_s"] = {"copy_and_frame": cp, "sketch_and_query": dv, "output": ou}
        if args.gz and args.host_inflate_too:
            run("index_query_host_inflate", ["-I", "fof.txt", "-Q", "fof.txt", "-J", "0.1", "-O", "o2h.gz"], {"NIQKI_HOST_NO_GPU_INFLATE": "1"})
            res["host_inflate"] = {"index_genomes_per_s": round(args.genomes / phases["index_query_host_inflate"][0], 1),
                                   "query_genomes_per_s": round(args.genomes / phases["index_query_host_inflate"][1], 1),
                                   "outputs_equal": open(os.path.join(args.dir, "o2.gz"), "rb").read() == open(os.path.join(args.dir, "o2h.gz"), "rb").read()}
        # the REFERENCE's own program on the same files (oracle/_ref, where it travelled with the repo): its CPU path,
        # and the same program with compute_sketch / insert_sketch / query_sketch bound to the C ABI
        # (oracle/ref_gpu_ops.cpp) -- wall time of `-I fof -Q fof` on the first files, threads = the CPUs this job has
        ref_cpu = os.path.join(ROOT, "oracle", "_ref", "niqki_ref")
        ref_gpu = os.path.join(ROOT, "oracle", "_ref", "niqki_ref_gpu")
        if not args.gz and args.reference and os.path.exists(ref_cpu) and os.path.exists(ref_gpu):
            n_ref = min(args.genomes, args.reference)
            open(os.path.join(args.dir, "ref.txt"), "w").write("\n".join(names[:n_ref]) + "\n")
            threads = cpu_quota()
            renv = dict(os.environ, OMP_NUM_THREADS=str(threads), NIQKI_REF_GPU_REPORT="1")
            ref = {"files": n_ref, "threads": threads}
            for tag, binary in (("reference_cpu", ref_cpu), ("reference_on_c_abi", ref_gpu)):
                t0 = time.time()
                r = subprocess.run([binary, "-I", "ref.txt", "-Q", "ref.txt", "-J", "0.1", "-O", tag + ".gz"] + extra, cwd=args.dir,
                                   capture_output=True, text=True, timeout=1800, env=renv)
                dt = time.time() - t0
                if r.returncode != 0:
                    ref[tag] = {"error": (r.stderr or r.stdout)[-300:]}
                    continue
                ref[tag] = {"wall_s": round(dt, 3), "genomes_per_s_index_plus_query": round(2 * n_ref / dt, 1)}
                rep = [l for l in r.stderr.splitlines() if l.startswith("niqki_ref_gpu:")]
                if rep:
                    ref[tag]["report"] = rep[-1]
            t0 = time.time()
            r = subprocess.run([BIN, "-I", "ref.txt", "-Q", "ref.txt", "-J", "0.1", "-O", "ours_ref.gz"] + extra, cwd=args.dir,
                               capture_output=True, text=True, timeout=1800, env=env)
            ref["niqki_host_program"] = {"wall_s": round(time.time() - t0, 3), "genomes_per_s_index_plus_query": round(2 * n_ref / (time.time() - t0), 1)}
            same = None
            try:
                a = gzip.open(os.path.join(args.dir, "reference_cpu.gz")).read().decode().split("\n")
                b = gzip.open(os.path.join(args.dir, "reference_on_c_abi.gz")).read().decode().split("\n")
                c = gzip.open(os.path.join(args.dir, "ours_ref.gz")).read().decode().split("\n")
                norm = lambda ls: sorted(tuple(sorted(t for t in l.split(" ") if t)) for l in ls if l.strip())  # noqa: E731
                same = norm(a) == norm(b) == norm(c)
            except (OSError, ValueError):
                pass
            ref["outputs_equal"] = same
            res["reference_program"] = ref
        if args.reads:
            rng = np.random.default_rng(3)
            src = niqki_amd.synth_genome_host(11, 0, 0, 0, args.len)
            st = rng.integers(0, args.len - 150, args.reads)
            with open(os.path.join(args.dir, "reads.fa"), "wb") as f:
                for i in range(0, args.reads, 65536):
                    blk = [b">r%d\n" % (i + j) + bytes(src[s:s + 150]) + b"\n" for j, s in enumerate(st[i:i + 65536])]
                    f.write(b"".join(blk))
            run("index_lines", ["-I", "fof.txt", "-l", "reads.fa", "-S", "12", "-W", "10", "-J", "0.1", "-O", "o3.gz"])
            res["lines_phase_s"] = phases["index_lines"][-1]
            res["reads_per_s"] = round(args.reads / phases["index_lines"][-1], 1)
        print(json.dumps(res))
    finally:
        shutil.rmtree(args.dir, ignore_errors=True)


if __name__ == "__main__":
    main()
