"""CPU (never on the GPU box): the host-side concurrency and I/O code under ThreadSanitizer and AddressSanitizer + UBSan.

The code under test is the PRODUCT's, compiled as it is:
  * niqki_amd/csrc/nq_combiner.h -- the leader / follower batching behind niqki_*_shared -- on a fake single-caller
    engine (tests/host_san/combiner_stress.cpp): 64 threads, random capacities, engine errors, allocation failures;
  * the whole `niqki` host program (niqki_amd/host: reader threads, the two-deep batch pipeline, the lines-mode reader
    and writer threads, the multi-member gzip writer / reader, the option parser, the FASTA packer) linked -- through
    `make -C niqki_amd/host SAN=thread|address ENGINE=...` -- onto tests/host_san/fake_engine.cpp, which answers the C
    ABI on the CPU with the parity oracle.  The runs must be sanitizer-clean AND give the reference CLI's golden texts,
    so the host logic is checked here against the reference as well (src/niqki_index.cpp:391-401, :461-500, :523-566).
The oracle itself (oracle/niqki_oracle.c) runs under -fsanitize=address,undefined inside the second binary."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, make_cli_workdir

HS = os.path.join(ROOT, "tests", "host_san")
BINDIR = os.path.join(HS, "bin")
HOST = os.path.join(ROOT, "niqki_amd", "host")
ENGINE = "../../tests/host_san/fake_engine.cpp ../../oracle/niqki_oracle.c"
SAN_ENV = {"thread": {"TSAN_OPTIONS": "halt_on_error=1 second_deadlock_stack=1"},
           # (the host program keeps its reader buffers and its Index to the end of the process, like the reference's
           # main never deletes its Index: leak checking would only list those)
           "address": {"ASAN_OPTIONS": "detect_leaks=0 abort_on_error=0", "UBSAN_OPTIONS": "halt_on_error=1 print_stacktrace=1"}}
FLAGS = {"thread": ["-fsanitize=thread"], "address": ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]}


def newer(target, sources):
    return os.path.exists(target) and all(os.path.getmtime(target) >= os.path.getmtime(s) for s in sources)


@pytest.fixture(scope="module", params=["thread", "address"])
def san(request):
    return request.param


@pytest.fixture(scope="module")
def stress_bin(san):
    out = os.path.join(BINDIR, "combiner_stress_" + san)
    src = [os.path.join(HS, "combiner_stress.cpp"), os.path.join(ROOT, "niqki_amd", "csrc", "nq_combiner.h")]
    if not newer(out, src):
        os.makedirs(BINDIR, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + FLAGS[san] + [src[0], "-o", out])
    return out


@pytest.fixture(scope="module")
def gzio_bin(san):
    out = os.path.join(BINDIR, "gzio_stress_" + san)
    src = [os.path.join(HS, "gzio_stress.cpp"), os.path.join(HOST, "gzio.h")]
    if not newer(out, src):
        os.makedirs(BINDIR, exist_ok=True)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-pthread"] + FLAGS[san] + [src[0], "-o", out, "-lz", "-ldl"])
    return out


@pytest.fixture(scope="module")
def niqki_bin(san):
    out = os.path.join(BINDIR, "niqki_fake_" + san)
    src = [os.path.join(HOST, f) for f in ("niqki_main.cpp", "index_host.cpp", "index_host.h", "file_reader.h", "gzio.h", "seqio.h", "Makefile")] + \
          [os.path.join(HS, "fake_engine.cpp"), os.path.join(ROOT, "oracle", "niqki_oracle.c"),
           os.path.join(ROOT, "niqki_amd", "csrc", "nq_pack.h"), os.path.join(ROOT, "include", "niqki_hip.h")]
    if not newer(out, src):
        subprocess.check_call(["make", "-C", HOST, "SAN=" + san, "ENGINE=" + ENGINE, "OUT=" + os.path.relpath(out, HOST)],
                              stdout=subprocess.DEVNULL)
    return out


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, native, gold):
    _, meta = gold
    return make_cli_workdir(tmp_path_factory.mktemp("san_cli"), native, meta)


def run(binary, san, td, args, ok=(0,), env=None):
    e = dict(os.environ, **SAN_ENV[san])
    e["NIQKI_HOST_GPU_INFLATE_MIN"] = "1"   # short lists of gzip files too go to the (stand-in) device inflate and back
    e.update(env or {})
    r = subprocess.run([binary] + args, cwd=td, capture_output=True, text=True, timeout=900, env=e)
    assert r.returncode in ok, "exit %d\n%s\n%s" % (r.returncode, r.stdout[-1500:], r.stderr[-6000:])
    assert "Sanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-6000:]
    return r


def gunzip(p):
    return gzip.open(p, "rb").read()


def test_combiner_under_sanitizer(stress_bin, san):
    """64 threads x 5 phases on one combiner: no data race / no invalid access, every answer right, the single-caller
    engine never entered twice, the capacity retry runs, inserts hand out every id once -- also with engine errors
    and with allocations inside submit() failing."""
    r = subprocess.run([stress_bin, "64", "60" if san == "thread" else "150"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, **SAN_ENV[san]))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-6000:]
    assert "combiner ok" in r.stdout and "Sanitizer" not in r.stderr


def test_host_program_goldens_under_sanitizer(niqki_bin, san, workdir, gold):
    """The reference CLI's golden texts and dump bytes from the sanitizer build of the host program: index + query +
    dump (whole-file mode, packed FASTA, parallel gzip writer), -G, load + query (streamed gzip reader), matrix,
    lines mode (-i / -l: reader and writer threads), the framing oddities."""
    _, meta = gold
    cli = meta["cli"]
    out = run(niqki_bin, san, workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "hits.gz", "-D", "idx.dump"]).stdout
    assert "Number of indexed genomes" in out and "12 |" in out
    assert gunzip(workdir / "hits.gz").decode() == cli["hits"]
    raw = gunzip(workdir / "idx.dump")
    assert len(raw) == cli["dump_len"] and hashlib.md5(raw).hexdigest() == cli["dump_md5"]
    assert (workdir / "idx.dump").read_bytes().count(b"\x1f\x8b\x08") >= 2          # several gzip members, written in parallel
    out = run(niqki_bin, san, workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-G", "40000", "-J", "0.1", "-O", "hits_G.gz",
                                        "-D", "idxG.dump"]).stdout
    assert "I chosed H=%d" % cli["dumpG_header"][2] in out
    assert gunzip(workdir / "hits_G.gz").decode() == cli["hits_G"]
    assert hashlib.md5(gunzip(workdir / "idxG.dump")).hexdigest() == cli["dumpG_md5"]
    run(niqki_bin, san, workdir, ["-L", "idx.dump", "-Q", "fof.txt", "-O", "hits_loaded.gz"])
    assert gunzip(workdir / "hits_loaded.gz").decode() == cli["hits_loaded"]
    run(niqki_bin, san, workdir, ["-M", "fof.txt", "-S", "10", "-O", "matrix.gz"])
    assert gunzip(workdir / "matrix.gz").decode() == cli["matrix"]
    run(niqki_bin, san, workdir, ["-i", "reads.fa", "-l", "reads.fa", "-S", "10", "-W", "10", "-J", "0.2", "-O", "lines.gz"])
    assert gunzip(workdir / "lines.gz").decode() == cli["lines"]
    (workdir / "nasty.fa").write_bytes(cli["nasty_fa_input"].encode("latin1"))
    (workdir / "nasty.fq").write_bytes(cli["nasty_fq_input"].encode("latin1"))
    run(niqki_bin, san, workdir, ["-I", "fof.txt", "-l", "nasty.fa", "-S", "10", "-J", "0", "-O", "nasty_fa.gz"])
    assert gunzip(workdir / "nasty_fa.gz").decode("latin1") == cli["nasty_fa"]
    run(niqki_bin, san, workdir, ["-I", "fof.txt", "-l", "nasty.fq", "-S", "10", "-J", "0", "-O", "nasty_fq.gz"])
    assert gunzip(workdir / "nasty_fq.gz").decode("latin1") == cli["nasty_fq"]
    run(niqki_bin, san, workdir, ["-i", "nasty.fa", "-Q", "fof.txt", "-S", "10", "-J", "0.02", "-O", "nasty_idx.gz"])
    assert gunzip(workdir / "nasty_idx.gz").decode("latin1") == cli["nasty_idx"]


def test_host_pipeline_many_files_and_streams_under_sanitizer(niqki_bin, san, workdir, gold, native):
    """What makes the threads meet: a list of 300 files (several batches: the reader pool runs ahead of the consumer,
    buffers are recycled, the next batch is prefetched) with plain, gzip and multi-member gzip inputs on 8 reader
    threads; a lines-mode file of several pieces (reader thread -> engine -> writer thread); errors that end a run
    early (a missing file in the list is skipped, a damaged gzip stops the run with a message, a missing list exits)."""
    _, meta = gold
    names = (workdir / "fof.txt").read_text().split()
    big = []
    for rep in range(25):
        for n in names:
            dst = "r%02d_%s%s" % (rep, n, "" if rep % 3 == 0 else ".gz")
            raw = (workdir / n).read_bytes()
            if not (workdir / dst).exists():
                if rep % 3 == 0:
                    (workdir / dst).write_bytes(raw)
                elif rep % 3 == 1:
                    cut = len(raw) // 3
                    (workdir / dst).write_bytes(gzip.compress(raw[:cut], 1) + gzip.compress(raw[cut:], 6) + b"\0" * 7)
                else:
                    (workdir / dst).write_bytes(gzip.compress(raw, 1))
            big.append(dst)
    (workdir / "big.txt").write_text("\n".join(big[:150] + ["no_such_file.fa"] + big[150:]) + "\n")
    env = {"NIQKI_HOST_THREADS": "8"}
    run(niqki_bin, san, workdir, ["-I", "big.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "big.gz"], env=env)
    got = gunzip(workdir / "big.gz").decode().splitlines()
    exp = meta["cli"]["hits"].splitlines()
    assert len(got) == len(exp) == 12
    for g, e in zip(got, exp):
        eh = dict(t.rsplit(":", 1) for t in e.split(" ")[1:] if t)
        gh = [t.rsplit(":", 1) for t in g.split(" ")[1:] if t]
        assert len(gh) == 25 * len(eh)
        for name, val in gh:
            assert eh[name.split("_", 1)[1].replace(".gz", "")] == val
    # the list as the QUERY side (every copy answers like its original), zlib instead of libdeflate
    run(niqki_bin, san, workdir, ["-I", "fof.txt", "-Q", "big.txt", "-S", "10", "-J", "0.1", "-O", "bigq.gz"],
        env=dict(env, NIQKI_HOST_ZLIB_ONLY="1", NIQKI_HOST_NO_PACK="1"))
    gotq = gunzip(workdir / "bigq.gz").decode().splitlines()
    assert len(gotq) == 300
    by_name = {e.split(" ")[0]: e.split(" ", 1)[1] for e in exp}
    for g, dst in zip(gotq, big):
        qname, rest = (g.split(" ", 1) + [""])[:2]
        assert qname == dst and rest == by_name[dst.split("_", 1)[1].replace(".gz", "")], dst
    # lines mode over several pieces
    rng = np.random.default_rng(9)
    genome = native.synth_genome_host(meta["seed"], 0, 0, 0, 40000)
    n = 60_000 if san == "address" else 30_000
    with open(workdir / "many.fa", "wb") as f:
        for i, s in enumerate(rng.integers(0, 40000 - 150, n)):
            f.write(b">read%d\n" % i + bytes(genome[s:s + 150]) + b"\n")
    run(niqki_bin, san, workdir, ["-I", "fof.txt", "-l", "many.fa", "-S", "8", "-W", "8", "-J", "0", "-O", "many.gz"])
    lines = gunzip(workdir / "many.gz").decode().split("\n")
    assert len(lines) == n + 1 and [l.split(" ", 1)[0] for l in lines[:n]] == [">read%d" % i for i in range(n)]
    assert all(l.count(":") == 12 for l in lines[:n])          # (J = 0: every indexed genome is listed, src/niqki_index.cpp:662)
    # early ends
    (workdir / "broken.fa.gz").write_bytes(gzip.compress((workdir / names[0]).read_bytes(), 1)[:-200])
    (workdir / "broken.txt").write_text("\n".join(names[:3] + ["broken.fa.gz"] + names[3:]) + "\n")
    r = run(niqki_bin, san, workdir, ["-I", "broken.txt", "-S", "10", "-O", "broken.gz"], ok=(0, 1), env=env)
    assert "broken.fa.gz" in (r.stdout + r.stderr) or r.returncode == 1
    r = run(niqki_bin, san, workdir, ["-I", "no_such_list.txt", "-S", "10", "-O", "nolist.gz"], ok=(0, 1))
    assert "Unable to open the file" in r.stdout
    r = run(niqki_bin, san, workdir, ["stray"], ok=(1,))
    assert "Bad usage!!!" in r.stdout
    r = run(niqki_bin, san, workdir, ["--gpus", "2", "-I", "fof.txt", "-S", "10", "-O", "mg.gz"], ok=(1,))   # the fake engine has no groups
    assert "whole-range handles only" in r.stderr          # (an engine error on the way out: message, exit code 1, no hang)



def test_parallel_gzip_writer_and_reader_under_sanitizer(gzio_bin, san, tmp_path):
    """The dump files' writer and reader (size-tagged gzip members compressed, written and inflated side by side:
    niqki_amd/host/gzio.h) on blocks of every size around the 8 MB piece, with libdeflate and with zlib's codec: the
    same bytes back through the parallel reader and through zlib's gzread; damage is an exception."""
    for env, size in (({}, "full"), ({"NIQKI_HOST_ZLIB_ONLY": "1"}, "small")):
        r = subprocess.run([gzio_bin, str(tmp_path), size], capture_output=True, text=True, timeout=900, env=dict(os.environ, **SAN_ENV[san], **env))
        assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout[-500:] + r.stderr[-3000:]
