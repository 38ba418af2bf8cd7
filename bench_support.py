"""bench_support.py -- the parts of bench.py that are not the measurement itself: the deterministic workload
specification (which synthetic genome is indexed genome g / query q), the counter passes a default run takes of itself
(rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU over short copies of the same command), the launcher of
`python bench.py --gpus N`, the `roofline` object of the line, and what the host says about its CPUs.
Nothing here touches oracle/ (the CPU baseline and the parity legs stay in bench.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
BENCH_K = 31   # the bench's k-mer length (BASELINE.json: K=31 S=15 W=12)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class Budget:
    """Wall-clock budget of one bench.py run (--budget-s, counted from the start of the process): the headline -- index
    build, warm-up, the K timed steps, the roofline steps, the CPU baseline on its sample -- always runs; every other leg
    starts only while its estimated time is left, a child process gets min(its own limit, what is left), and what was
    dropped is named in the line.  A leg that hangs inside this process is cut by the watchdog (arm()), which prints the
    line as it stands and ends the process."""

    def __init__(self, seconds, t0=None):
        self.t0 = time.time() if t0 is None else t0
        self.seconds = float(seconds)
        self.dropped = []
        self.timeline = []
        self._mark = self.t0
        self._timer = None

    def left(self):
        return self.t0 + self.seconds - time.time()

    def elapsed(self):
        return time.time() - self.t0

    def want(self, name, estimate_s):
        """True if `estimate_s` seconds are left for the leg `name`; else the leg is recorded as dropped."""
        if self.left() >= estimate_s:
            return True
        self.dropped.append({"leg": name, "needs_s": estimate_s, "left_s": round(self.left(), 1)})
        log("[bench] budget: %s dropped (needs ~%.0f s, %.0f s left)" % (name, estimate_s, self.left()))
        return False

    def lap(self, name):
        """closes the timeline entry `name` (seconds since the last lap)"""
        now = time.time()
        self.timeline.append([name, round(now - self._mark, 2)])
        self._mark = now

    def child(self, name, cmd, own_timeout_s, estimate_s, **popen_kw):
        """Runs cmd as a child process (a session of its own, killed as a group) for at most min(own_timeout_s, what is
        left); returns (returncode, stdout bytes) or None when it was dropped, timed out or could not start."""
        import signal
        import subprocess
        if not self.want(name, estimate_s):
            return None
        limit = max(1.0, min(float(own_timeout_s), self.left()))
        try:
            pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True, **popen_kw)
        except OSError:
            self.dropped.append({"leg": name, "error": "could not start"})
            return None
        try:
            out, _ = pr.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(pr.pid, signal.SIGKILL)
            except OSError:
                pass
            pr.communicate()
            self.dropped.append({"leg": name, "killed_after_s": round(limit, 1)})
            log("[bench] budget: %s killed after %.0f s" % (name, limit))
            return None
        return pr.returncode, out

    def arm(self, grace_s, emit):
        """watchdog: grace_s after the deadline emit() is called from a timer thread and the process ends (a leg of this
        process that hangs -- a kernel that never finishes -- must not cost the run its line)"""
        import threading

        def fire():
            log("[bench] budget: %.0f s past the deadline, printing the line as it stands" % grace_s)
            try:
                emit(True)
            finally:
                os._exit(0)
        self._timer = threading.Timer(max(1.0, self.left() + grace_s), fire)
        self._timer.daemon = True
        self._timer.start()

    def disarm(self):
        if self._timer is not None:
            self._timer.cancel()
            self._timer = None

    def record(self):
        return {"budget_s": self.seconds, "used_s": round(self.elapsed(), 1), "dropped": self.dropped, "timeline_s": self.timeline}


def genome_spec(g, n_fam, fam_size):
    """Indexed genome g: family g // fam_size, member g % fam_size; member 0 is the
    ancestor, the others carry substitution rates spread geometrically over
    0.1 % .. 5 % so that in-family Jaccard spans ~0.05 .. 0.95."""
    fam = g // fam_size
    mem = g % fam_size
    rate = np.where(mem == 0, 0, np.round(16 * (820 / 16) ** ((mem - 1) / max(fam_size - 2, 1)))).astype(np.uint32)
    return fam.astype(np.uint32), mem.astype(np.uint32), rate


def query_spec(q, n_fam):
    """Query q: a fresh mutant (member id >= 2^20) of a pseudo-random indexed
    family; every 10th query comes from a family that is not indexed."""
    h = (q.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15)) >> np.uint64(33)
    fam = (h % np.uint64(n_fam)).astype(np.uint32)
    fam = np.where(q % 10 == 9, n_fam + q, fam).astype(np.uint32)
    mem = ((1 << 20) + q).astype(np.uint32)
    rate = (16 + (h >> np.uint64(8)) % np.uint64(400)).astype(np.uint32)
    return fam, mem, rate


def measure_counters(args, budget=None):
    """Hardware counters of a short run of this same command under rocprofv3 (one --pmc pass per counter, no
    trace flags, child processes started before this process touches the GPU):
      * HBM-side bytes per gather launch (gather kernel + look-up pre-pass + locality probe):
        2 x FETCH_SIZE (gfx950 tallies 128-byte requests at 64 bytes, MI355X_MICROARCH.md) + WRITE_SIZE, in KB;
      * vector instructions per k-mer of the sketch kernel: SQ_INSTS_VALU (wave instructions, all XCDs)
        x 64 lanes over the k-mers its launches of that run rolled.
    None if anything goes wrong."""
    import csv
    import glob
    import re
    import shutil
    import signal
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None
    rx = re.compile(r"gather_kernel<|lookup(_rows)?_kernel<|probe_kernel<")
    # not under another profiler: the nested rocprofv3 would inherit its preload / tool variables
    if ("ROCP_TOOL_LIBRARIES" in os.environ or any(k.startswith("ROCPROF_") for k in os.environ)
            or "rocprof" in os.environ.get("LD_PRELOAD", "")):
        return None
    kb = {}
    launches = None
    valu = None
    c_steps, c_warm = 2, 1
    t0 = time.time()
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
        out = tempfile.mkdtemp(prefix="niqki_pmc_", dir="/tmp")
        try:
            cmd = ["rocprofv3", "--pmc", counter, "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--pmc-child", "--steps", str(c_steps), "--warmup", str(c_warm),
                   "--genomes", str(args.genomes), "--batch", str(args.batch), "--len", str(args.len),
                   "--family", str(args.family), "--seed", str(args.seed), "--ring", str(args.ring)]
            env = dict(os.environ, TMPDIR="/tmp")
            # a session of its own: on a timeout the whole group goes (rocprofv3 AND the bench under it)
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                  start_new_session=True)
            try:
                rc = pr.wait(timeout=240 if budget is None else max(5.0, min(240.0, budget.left() - 60.0)))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except OSError:
                    pass
                pr.wait()
                return None
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if rc != 0 or not files:
                if counter == "SQ_INSTS_VALU":
                    break                  # (the traffic passes stand on their own)
                return None
            total, n_gather, v_sum, v_n = 0.0, 0, 0.0, 0
            with open(files[0], newline="") as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") != counter:
                        continue
                    if counter == "SQ_INSTS_VALU":
                        if "sketch_kernel<" in name:
                            v_sum += float(row["Counter_Value"])
                            v_n += 1
                    elif rx.search(name):
                        total += float(row["Counter_Value"])
                        n_gather += "gather_kernel<" in name
            if counter == "SQ_INSTS_VALU":
                # every sketch launch of the child: the index build (all genomes) and its warm-up + timed query batches
                # (K = 31: the bench's only k-mer length -- config.K below; a launch of `len` bases rolls len - K k-mers,
                # src/niqki_index.cpp:342.  The child runs --no-legs, so its sketch launches are exactly these; the
                # exact re-run of a genome whose filtered pass left a slot empty, 1 in ~20 000, is inside the
                # numerator and not the denominator: < 0.01 %.)
                kmers = (args.genomes + (c_steps + c_warm) * args.batch) * max(args.len - BENCH_K, 0)
                if v_n and kmers:
                    valu = {"valu_per_kmer": v_sum * 64.0 / kmers, "launches": v_n}
                continue
            if n_gather == 0:
                return None
            kb[counter] = total / n_gather
            launches = n_gather
        except (OSError, subprocess.SubprocessError, KeyError, ValueError):
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    log("[bench] HBM traffic of the gather path: FETCH_SIZE %.0f KB, WRITE_SIZE %.0f KB per launch (%d launches); sketch kernel %s "
        "vector instructions per k-mer (%.0f s)" % (kb["FETCH_SIZE"], kb["WRITE_SIZE"], launches,
                                                    ("%.2f" % valu["valu_per_kmer"]) if valu else "n/a", time.time() - t0))
    return {"bytes_per_launch": (2 * kb["FETCH_SIZE"] + kb["WRITE_SIZE"]) * 1024,
            "fetch_kb": kb["FETCH_SIZE"], "write_kb": kb["WRITE_SIZE"],
            "valu": valu,
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes over %d launches of this same command, run by "
                      "bench.py before its timed run: 2 x FETCH_SIZE (gfx950) + WRITE_SIZE of gather_kernel, the look-up "
                      "pre-pass and the locality probe" % launches}


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command under
    torch.distributed.run as a CHILD process (a session of its own, killed as a group on a time-out), relay
    its stdout -- the one JSON line of rank 0 -- and return its exit code.  Called before this process has
    imported torch or made any HIP call: a process that has initialised the GPU must not be replaced or
    forked into a launcher on this pool."""
    import signal
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL and the ipc transport need on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    limit = float(os.environ.get("NIQKI_BENCH_LAUNCH_TIMEOUT", "3000"))
    log("[bench] --gpus %d without a launcher: starting %s" % (n, " ".join(cmd[1:10]) + " ..."))
    pr = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, start_new_session=True)
    try:
        out, _ = pr.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(pr.pid, signal.SIGKILL)
        except OSError:
            pass
        pr.wait()
        log("[bench] the %d ranks did not finish within %.0f s: killed" % (n, limit))
        return 124
    except KeyboardInterrupt:
        try:
            os.killpg(pr.pid, signal.SIGTERM)
        except OSError:
            pass
        pr.wait()
        return 130
    lines = [ln for ln in out.decode(errors="replace").splitlines() if ln.startswith("{")]
    if lines:
        sys.stdout.write(lines[-1] + "\n")
        sys.stdout.flush()
    elif pr.returncode == 0:
        log("[bench] the ranks exited 0 without a JSON line")
        return 1
    return pr.returncode


def roofline_record(achieved_alg, traffic, traffic_source, live, gather_ms, launches, alg_bytes, layout_min, copy_gbs, measured_in,
                    T, n_q_local, pipeline):
    """The `roofline` object of the line.  Three byte counts per launch over the same launch time:
      * `achieved` / `frac`: the bytes this layout CANNOT avoid (2-byte ids of the touched buckets, one 8-byte table entry per
        slot and tile, the sketch in, the 2-byte counter row out) -- a lower bound of what HBM moved, so a fraction that can
        never flatter the kernel;
      * `achieved_algorithmic` / `frac_algorithmic`: SURVEY.md 8(d)'s formula 4T + 20F, which counts 4 bytes per id as the
        reference stores them and every query's lines for itself: it can pass 1 (the layout stores 2-byte ids);
      * `traffic` / `traffic_frac`: what the L2s requested from the fabric (rocprofv3 --pmc: 2 x FETCH_SIZE + WRITE_SIZE,
        MI355X_MICROARCH.md HBM section) -- Infinity-Cache hits and lines that several XCDs fetch are inside, so an UPPER
        bound of the HBM-proper fraction; `copy_ceiling_frac` holds it against the streaming copy measured in the run.
    No counter of this device separates the Infinity Cache's hits from HBM reads (TCC_EA0_RDREQ_DRAM counts requests
    "destined for DRAM", cache or not: profiles/r06_bench_pmc_default_summary.txt), so the truth lies between `frac` and
    `traffic_frac`."""
    n = max(1, launches)
    t_launch = gather_ms / n * 1e-3
    alg_l, lay_l = alg_bytes / n, layout_min / n
    real_gbs = traffic / t_launch / 1e9 if (traffic and t_launch) else None
    lay_gbs = lay_l / t_launch / 1e9 if t_launch else None
    rec = {
        "kernel": "nq::gather_kernel (gather-histogram, rank 0's slot shard) incl. its look-up pre-pass and probe / order passes",
        "bound": "hbm", "achieved": lay_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (lay_gbs / HBM_PEAK_GBS) if lay_gbs else None,
        "frac_basis": "layout_min_bytes_per_launch / avg_launch_ms / peak: the bytes this layout cannot avoid (a lower bound of what HBM moved)",
        "achieved_algorithmic": achieved_alg,
        "frac_algorithmic": achieved_alg / HBM_PEAK_GBS,
        "frac_layout_min": (lay_gbs / HBM_PEAK_GBS) if lay_gbs else None,
        "traffic": traffic, "traffic_source": traffic_source,
        "traffic_gbs": real_gbs,
        "traffic_frac": (real_gbs / HBM_PEAK_GBS) if real_gbs else None,
        "traffic_over_layout_min": (traffic / lay_l) if (traffic and lay_l) else None,
        "traffic_fetch_kb": live["fetch_kb"] if live else None, "traffic_write_kb": live["write_kb"] if live else None,
        "copy_gbs": copy_gbs,
        "copy_ceiling_frac": (real_gbs / copy_gbs) if (real_gbs and copy_gbs) else None,
        "note": ("frac counts only bytes the layout must move; frac_algorithmic counts SURVEY.md 8d's 4T + 20F (4-byte ids, shared lines "
                 "counted per query) and can pass 1; traffic_frac is what the L2s requested from the fabric, Infinity-Cache hits "
                 "included (an upper bound of the HBM fraction; it can pass copy_ceiling).  traffic_over_layout_min: half-empty 128-byte "
                 "bucket lines are the difference.  What bounds the launch: the lines it fetches -- without any of its LDS atomics it "
                 "is 3 % faster (profiles/r06_one_tile_cost_model.txt)"
                 + ("; with the next batch's sketch kernel beside the gather the two share the CUs, the gather launch time here "
                    "is not a roofline figure" if pipeline else "")),
        "algorithmic_bytes_per_launch": alg_l,
        "layout_min_bytes_per_launch": lay_l,
        "launches": launches, "avg_launch_ms": gather_ms / n,
        "measured_in": measured_in or "the timed steps",
        "gathered_ids_per_query": T / max(1, n_q_local),
    }
    return rec


def host_cpu_info():
    """What this process may use of the host: logical CPUs, physical cores, the affinity mask and the cgroup's CPU
    quota (a container is often handed a share of the node: more threads than that only take turns)."""
    phys = set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pid = line.split(":")[1].strip()
            elif line.startswith("core id"):
                cid = line.split(":")[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = os.cpu_count()
    return {"logical_cpus": os.cpu_count(), "physical_cores": len(phys) or None, "affinity_cpus": aff,
            "cgroup_cpu_quota": quota}
