"""GPU: the `niqki` host program (niqki_amd/bin/niqki, C++17 on the C ABI)
against the text / dump outputs the reference's own CLI produced for the same
files (tests/golden/reference_meta.json "cli", made by oracle/make_goldens.py)."""
import gzip
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, family_spec, make_cli_workdir

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "niqki_amd", "bin", "niqki")


@pytest.fixture(scope="module")
def workdir(tmp_path_factory, native, gold):
    _, meta = gold
    return make_cli_workdir(tmp_path_factory.mktemp("cli"), native, meta)


# (lists of gzip files as short as these tests' are inflated by the reader threads unless told otherwise: the device path)
os.environ["NIQKI_HOST_GPU_INFLATE_MIN"] = "1"


def run(td, args):
    assert os.path.exists(BIN), "niqki_amd/bin/niqki missing: run __graft_entry__.build()"
    r = subprocess.run([BIN] + args, cwd=td, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return r.stdout


def gunzip(path):
    return gzip.open(path, "rb").read()


def assert_same_text(got, exp):
    """Identical tokens; Jaccard values within 1e-6 (they are equal: same %g of exact counts)."""
    gl, el = got.split("\n"), exp.split("\n")   # not splitlines(): a header may hold a '\r'
    assert len(gl) == len(el)
    for a, b in zip(gl, el):
        if a == b:
            continue
        ta, tb = a.replace("\t", " ").split(" "), b.replace("\t", " ").split(" ")
        assert len(ta) == len(tb), (a, b)
        for x, y in zip(ta, tb):
            if x == y:
                continue
            nx, vx = x.rsplit(":", 1) if ":" in x else ("", x)
            ny, vy = y.rsplit(":", 1) if ":" in y else ("", y)
            assert nx == ny and abs(float(vx) - float(vy)) <= 1e-6, (x, y)


def test_index_query_dump(workdir, gold):
    _, meta = gold
    out = run(workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "hits.gz", "-D", "idx.dump"])
    assert "Number of indexed genomes" in out and "12 |" in out
    assert_same_text(gunzip(workdir / "hits.gz").decode(), meta["cli"]["hits"])
    raw = gunzip(workdir / "idx.dump")
    assert len(raw) == meta["cli"]["dump_len"]
    assert hashlib.md5(raw).hexdigest() == meta["cli"]["dump_md5"]


def test_genome_size_option(workdir, gold):
    """-G: select_best_H after the constructor; the dump header carries the chosen H."""
    _, meta = gold
    out = run(workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-G", "40000", "-J", "0.1", "-O", "hits_G.gz",
                        "-D", "idxG.dump"])
    assert "I chosed H=%d" % meta["cli"]["dumpG_header"][2] in out
    assert_same_text(gunzip(workdir / "hits_G.gz").decode(), meta["cli"]["hits_G"])
    raw = gunzip(workdir / "idxG.dump")
    assert np.frombuffer(raw[:24], np.uint32).tolist() == meta["cli"]["dumpG_header"]
    assert hashlib.md5(raw).hexdigest() == meta["cli"]["dumpG_md5"]


def test_record_framing_oddities(workdir, gold):
    """Hand-made FASTA / FASTQ files with every framing oddity (CRLF, blank lines, consecutive
    headers, '>' inside lines, first line without '>', 0xFF at a line start, no final newline,
    FASTQ record without quality lines): same text as the reference CLI, lines mode both ways."""
    _, meta = gold
    cli = meta["cli"]
    (workdir / "nasty.fa").write_bytes(cli["nasty_fa_input"].encode("latin1"))
    (workdir / "nasty.fq").write_bytes(cli["nasty_fq_input"].encode("latin1"))
    run(workdir, ["-I", "fof.txt", "-l", "nasty.fa", "-S", "10", "-J", "0", "-O", "nasty_fa.gz"])
    assert_same_text(gunzip(workdir / "nasty_fa.gz").decode("latin1"), cli["nasty_fa"])
    run(workdir, ["-I", "fof.txt", "-l", "nasty.fq", "-S", "10", "-J", "0", "-O", "nasty_fq.gz"])
    assert_same_text(gunzip(workdir / "nasty_fq.gz").decode("latin1"), cli["nasty_fq"])
    run(workdir, ["-i", "nasty.fa", "-Q", "fof.txt", "-S", "10", "-J", "0.02", "-O", "nasty_idx.gz"])
    assert_same_text(gunzip(workdir / "nasty_idx.gz").decode("latin1"), cli["nasty_idx"])


def test_gzip_inputs_and_many_files(workdir, gold):
    """gzip-compressed genome files and a list longer than one GPU batch give the same
    hits as the plain files."""
    import shutil
    _, meta = gold
    names = (workdir / "fof.txt").read_text().split()
    big = []
    for rep in range(25):   # 300 files > kWholeBatchFiles
        for n in names:
            dst = "r%02d_%s.gz" % (rep, n)
            if not (workdir / dst).exists():
                raw = (workdir / n).read_bytes()
                if rep % 5 == 1:     # a multi-member file (what `cat a.gz b.gz` makes), bytes behind the last member
                    cut = len(raw) // 3
                    (workdir / dst).write_bytes(gzip.compress(raw[:cut], 1) + gzip.compress(raw[cut:], 6) + b"\0" * 7)
                elif rep % 5 == 2:   # BGZF, what bgzip writes: members of <= 64 KB that carry their size, an empty one last
                    import struct
                    import zlib
                    out = bytearray()
                    for a in list(range(0, len(raw), 65280)) + [len(raw)]:
                        piece = raw[a:a + 65280] if a < len(raw) else b""
                        c = zlib.compressobj(6, zlib.DEFLATED, -15)
                        body = c.compress(piece) + c.flush()
                        out += (b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", 18 + len(body) + 8 - 1) + body +
                                struct.pack("<II", zlib.crc32(piece) & 0xFFFFFFFF, len(piece)))
                    (workdir / dst).write_bytes(bytes(out))
                else:
                    with gzip.open(workdir / dst, "wb", compresslevel=1) as f:
                        f.write(raw)
            big.append(dst)
    (workdir / "big.txt").write_text("\n".join(big) + "\n")
    run(workdir, ["-I", "big.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "big.gz"])
    got = gunzip(workdir / "big.gz").decode().splitlines()
    exp = meta["cli"]["hits"].splitlines()
    assert len(got) == len(exp) == 12
    for g, e in zip(got, exp):
        # every hit of the plain run appears 25 times (once per copy), same value
        eh = dict(t.rsplit(":", 1) for t in e.split(" ")[1:] if t)
        gh = [t.rsplit(":", 1) for t in g.split(" ")[1:] if t]
        assert len(gh) == 25 * len(eh)
        for name, val in gh:
            base = name.split("_", 1)[1][:-3]
            assert eh[base] == val
    # the host program inflates whole files with libdeflate where the system has it: zlib's answer is the same
    r = subprocess.run([BIN, "-I", "big.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "big_zlib.gz"], cwd=workdir,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, NIQKI_HOST_ZLIB_ONLY="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert gunzip(workdir / "big_zlib.gz") == gunzip(workdir / "big.gz")
    # gzip files cross PCIe as they are and are inflated on the device (nq_inflate.hip; the multi-member ones come back
    # refused and go through zlib here): inflating everything on the host gives the same text
    r = subprocess.run([BIN, "-I", "big.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "big_host.gz"], cwd=workdir,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, NIQKI_HOST_NO_GPU_INFLATE="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert gunzip(workdir / "big_host.gz") == gunzip(workdir / "big.gz")
    # BGZF files go to the device whatever the list's length (their members are a wavefront's job each), other gzip
    # files of a short list to the reader threads: the default policy, same text
    (workdir / "bgzf.txt").write_text("\n".join(b for b in big if int(b[1:3]) % 5 in (2, 3)) + "\n")
    env_default = {k: v for k, v in os.environ.items() if k != "NIQKI_HOST_GPU_INFLATE_MIN"}
    outs = []
    for env in (env_default, dict(env_default, NIQKI_HOST_NO_GPU_INFLATE="1")):
        r = subprocess.run([BIN, "-I", "bgzf.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "bgzf.gz"], cwd=workdir,
                           capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(gunzip(workdir / "bgzf.gz"))
    assert outs[0] == outs[1] and len(outs[0]) > 100
    # a damaged file (a flipped byte in the middle of the stream; a cut one) ends the run the same way either way
    raw = bytearray((workdir / big[0]).read_bytes())
    raw[len(raw) // 2] ^= 0x10
    (workdir / "damaged.fa.gz").write_bytes(bytes(raw))
    (workdir / "cut.fa.gz").write_bytes((workdir / big[2]).read_bytes()[:-40])
    for victim in ("damaged.fa.gz", "cut.fa.gz"):
        (workdir / "bad.txt").write_text("\n".join(big[:40] + [victim] + big[40:80]) + "\n")
        outs = []
        for env in ({}, {"NIQKI_HOST_NO_GPU_INFLATE": "1"}):
            r = subprocess.run([BIN, "-I", "bad.txt", "-S", "10", "-J", "0.1", "-O", "bad.gz"], cwd=workdir,
                               capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
            outs.append((r.returncode, r.stderr.strip().splitlines()[-1:]))
        assert outs[0] == outs[1] and outs[0][0] != 0 and victim in outs[0][1][0], outs
    # three shards, each staging (and prefetching) its share of every batch: the same text
    r = subprocess.run([BIN, "--gpus", "3", "-I", "big.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "big_mg.gz"], cwd=workdir,
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, NIQKI_SHARDS_ON_ONE_DEVICE="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert gunzip(workdir / "big_mg.gz") == gunzip(workdir / "big.gz")
    # ... and as the QUERY list: three batches through the host program's copy / compute pipeline (the bytes of
    # batch i+1 cross to the device under batch i's kernels), every copy answers like its original
    run(workdir, ["-I", "fof.txt", "-Q", "big.txt", "-S", "10", "-J", "0.1", "-O", "bigq.gz"])
    got = gunzip(workdir / "bigq.gz").decode().splitlines()
    assert len(got) == len(big) == 300
    by_name = {e.split(" ")[0]: e.split(" ", 1)[1] if " " in e else "" for e in exp}
    for g, dst in zip(got, big):
        qname, rest = (g.split(" ", 1) + [""])[:2]
        assert qname == dst + ":" or qname.startswith(dst), (qname, dst)
        orig = dst.split("_", 1)[1][:-3]
        key = next(k for k in by_name if k.startswith(orig))
        assert rest == by_name[key], dst


@pytest.mark.parametrize("fmt", ["fa", "fq"])
def test_lines_mode_streams_many_pieces(workdir, native, gold, fmt):
    """A read file several pieces long (reader thread -> GPU -> writer thread): names, order and
    hits equal those of the record-level ABI path on the same reads."""
    _, meta = gold
    fam, mem, rate = family_spec(4, 8)
    genomes = [native.synth_genome_host(meta["seed"], int(f), int(m), int(r), 40000)
               for f, m, r in zip(fam[:12], mem[:12], rate[:12])]
    rng = np.random.default_rng(17)
    n = 150_000
    which = rng.integers(0, 12, n)
    start = rng.integers(0, 40000 - 150, n)
    reads = [genomes[w][s:s + 150] for w, s in zip(which, start)]
    path = workdir / ("many.%s" % fmt)
    with open(path, "wb") as f:
        for i, r in enumerate(reads):
            if fmt == "fa":
                f.write(b">read%d\n" % i + bytes(r) + b"\n")
            else:
                f.write(b"@read%d\n" % i + bytes(r) + b"\n+\n" + b"I" * 150 + b"\n")
    assert path.stat().st_size > (20 << 20)   # > 2 pieces of 8 MB
    run(workdir, ["-I", "fof.txt", "-l", path.name, "-S", "10", "-W", "10", "-J", "0.05", "-O", "many.gz"])
    got = gunzip(workdir / "many.gz").decode().split("\n")
    e = native.Engine(K=31, S=10, W=10, H=4, J=0.05)
    e.insert(e.sketch(genomes))
    names = (workdir / "fof.txt").read_text().split()
    off, hc, hg = e.query_sequences(reads)
    e.close()
    assert len(got) == n + 1 and got[-1] == ""
    head = b">" if fmt == "fa" else b"@"
    for i in range(0, n, 997):   # every 997th line in full, all names below
        exp = (head + b"read%d" % i).decode() + " " + "".join(
            "%s:%g " % (names[g], c / 1024) for c, g in zip(hc[off[i]:off[i + 1]], hg[off[i]:off[i + 1]]))
        assert got[i] == exp, i
    assert [g.split(" ", 1)[0] for g in got[:n]] == [(head + b"read%d" % i).decode() for i in range(n)]
    assert sum(len(g.split(" ")) - 2 for g in got[:n]) == int(off[-1])   # total hits


def test_reference_example_data(tmp_path, gold):
    """BASELINE.json configs[0]: the nine E. coli genomes the reference ships as its example
    (data fixtures in tests/golden/ecoli), default parameters (K=31 S=15 W=12): --matrix gives the
    README's matrix, index + query the reference CLI's hits, the dump its bytes."""
    _, meta = gold
    exp = meta["ecoli_cli"]
    edir = os.path.join(ROOT, "tests", "golden", "ecoli")
    run(edir, ["-M", "file_of_file.txt", "-O", str(tmp_path / "m.gz")])
    assert_same_text(gunzip(tmp_path / "m.gz").decode(), exp["matrix"])
    assert "ecoli01p.fa.gz\t1\t0.967773\t0.938019\t" in exp["matrix"]       # README.md:118-128
    run(edir, ["-I", "file_of_file.txt", "-Q", "file_of_file.txt", "-J", "0.8", "-O", str(tmp_path / "h.gz"),
               "-D", str(tmp_path / "d.gz")])
    assert_same_text(gunzip(tmp_path / "h.gz").decode(), exp["hits_J0.8"])
    h = hashlib.md5()
    n = 0
    with gzip.open(tmp_path / "d.gz", "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
            n += len(blk)
    assert n == exp["dump_len"] and h.hexdigest() == exp["dump_md5"]


def test_matrix(workdir, gold):
    _, meta = gold
    run(workdir, ["-M", "fof.txt", "-S", "10", "-O", "matrix.gz"])
    assert_same_text(gunzip(workdir / "matrix.gz").decode(), meta["cli"]["matrix"])


def test_lines_mode(workdir, gold):
    _, meta = gold
    run(workdir, ["-i", "reads.fa", "-l", "reads.fa", "-S", "10", "-W", "10", "-J", "0.2", "-O", "lines.gz"])
    assert_same_text(gunzip(workdir / "lines.gz").decode(), meta["cli"]["lines"])


def test_load_then_query(workdir, gold):
    _, meta = gold
    if not (workdir / "idx.dump").exists():
        run(workdir, ["-I", "fof.txt", "-S", "10", "-J", "0.1", "-O", "tmp.gz", "-D", "idx.dump"])
    run(workdir, ["-L", "idx.dump", "-Q", "fof.txt", "-O", "hits_loaded.gz"])
    assert_same_text(gunzip(workdir / "hits_loaded.gz").decode(), meta["cli"]["hits_loaded"])


def test_bad_usage(workdir):
    r = subprocess.run([BIN, "stray"], cwd=workdir, capture_output=True, text=True)
    assert r.returncode == 1 and "Bad usage!!!" in r.stdout
    r = subprocess.run([BIN, "-K", "x"], cwd=workdir, capture_output=True, text=True)
    assert r.returncode == 1 and "requires a numeric argument" in r.stderr


def test_dump_file_equals_the_committed_fixture(workdir, gold):
    """tests/golden/ours_cli_dump.gz is the multi-member gzip dump this host program wrote on a
    GPU box (tools/make_dump_fixture.py); tests/test_oracle_golden.py lets the REAL reference
    binary load it in the build container (the reference never travels to the GPU box).  Here:
    the program still writes those bytes (same members after gunzip, same payload md5)."""
    _, meta = gold
    run(workdir, ["-I", "fof.txt", "-S", "10", "-J", "0.1", "-O", "tmp2.gz", "-D", "ours.dump"])
    fixture = os.path.join(GOLD, "ours_cli_dump.gz")
    assert os.path.exists(fixture), "run tools/make_dump_fixture.py on a GPU box and commit its output"
    assert gunzip(workdir / "ours.dump") == gunzip(fixture)
    assert (workdir / "ours.dump").read_bytes().count(b"\x1f\x8b\x08") >= 2   # really several gzip members


@pytest.mark.parametrize("gpus", [3, 8])
def test_host_program_sharded_over_gpus(workdir, gold, gpus, monkeypatch):
    """`niqki --gpus N`: the index cut by sketch-slot range over N shards (here all on the one GPU
    of the box: NIQKI_SHARDS_ON_ONE_DEVICE, the exchange then runs on device copies instead of RCCL),
    every shard framing and sketching its share of each batch.  Same hits, matrix, lines-mode text and
    dump bytes as the reference CLI's goldens, i.e. as one GPU."""
    monkeypatch.setenv("NIQKI_SHARDS_ON_ONE_DEVICE", "1")
    _, meta = gold
    g = ["--gpus", str(gpus)]
    run(workdir, g + ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "mg_hits.gz", "-D", "mg.dump"])
    assert_same_text(gunzip(workdir / "mg_hits.gz").decode(), meta["cli"]["hits"])
    raw = gunzip(workdir / "mg.dump")
    assert len(raw) == meta["cli"]["dump_len"] and hashlib.md5(raw).hexdigest() == meta["cli"]["dump_md5"]
    run(workdir, g + ["-L", "mg.dump", "-Q", "fof.txt", "-O", "mg_loaded.gz"])
    assert_same_text(gunzip(workdir / "mg_loaded.gz").decode(), meta["cli"]["hits_loaded"])
    run(workdir, g + ["-M", "fof.txt", "-S", "10", "-O", "mg_matrix.gz"])
    assert_same_text(gunzip(workdir / "mg_matrix.gz").decode(), meta["cli"]["matrix"])
    run(workdir, g + ["-i", "reads.fa", "-l", "reads.fa", "-S", "10", "-W", "10", "-J", "0.2", "-O", "mg_lines.gz"])
    assert_same_text(gunzip(workdir / "mg_lines.gz").decode(), meta["cli"]["lines"])
    (workdir / "nasty.fa").write_bytes(meta["cli"]["nasty_fa_input"].encode("latin1"))
    run(workdir, g + ["-I", "fof.txt", "-l", "nasty.fa", "-S", "10", "-J", "0", "-O", "mg_nasty.gz"])
    assert_same_text(gunzip(workdir / "mg_nasty.gz").decode("latin1"), meta["cli"]["nasty_fa"])
    if (workdir / "many.fa").exists():   # 150 000 reads in several pieces (written by the streaming test above)
        run(workdir, ["-I", "fof.txt", "-l", "many.fa", "-S", "10", "-W", "10", "-J", "0.002", "-O", "sg_many.gz"])
        run(workdir, g + ["-I", "fof.txt", "-l", "many.fa", "-S", "10", "-W", "10", "-J", "0.002", "-O", "mg_many.gz"])
        one = gunzip(workdir / "sg_many.gz")
        assert gunzip(workdir / "mg_many.gz") == one and one.count(b":") > 1000


def test_sketch_size_16(workdir):
    """-S 16: the reference's uint32-counter branch (src/niqki_index.cpp:668-682) -- self hits count
    2^16 -- and its uint16 matrix counters, which wrap to 0 on the diagonal (:572).  Goldens from the
    reference CLI (oracle/make_goldens_s16.py)."""
    import json
    exp = json.load(open(os.path.join(GOLD, "reference_s16.json")))["cli_s16"]
    run(workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "s16_hits.gz"])
    assert_same_text(gunzip(workdir / "s16_hits.gz").decode(), exp["hits"])
    run(workdir, ["-M", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "s16_matrix.gz"])
    assert_same_text(gunzip(workdir / "s16_matrix.gz").decode(), exp["matrix"])
    assert "syn00.fa\t0\t0.970917" in exp["matrix"]


def test_paged_index_matrix_and_dump(workdir, gold):
    """--resident-mib: the index paged through 1 MiB of device memory (several pages of slots at S = 10): the
    matrix text, the hits and the dump bytes are those of the reference's CLI (src/niqki_index.cpp:570-628, :42-59)."""
    _, meta = gold
    run(workdir, ["-M", "fof.txt", "-S", "10", "-O", "matrix_pg.gz", "--resident-mib", "1"])
    assert_same_text(gunzip(workdir / "matrix_pg.gz").decode(), meta["cli"]["matrix"])
    run(workdir, ["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "hits_pg.gz", "-D", "idx_pg.dump", "--resident-mib", "1"])
    assert_same_text(gunzip(workdir / "hits_pg.gz").decode(), meta["cli"]["hits"])
    raw = gunzip(workdir / "idx_pg.dump")
    assert len(raw) == meta["cli"]["dump_len"] and hashlib.md5(raw).hexdigest() == meta["cli"]["dump_md5"]


def test_sketch_size_16_sharded(workdir, monkeypatch):
    """`niqki --gpus 2 -S 16`: every shard counts 2^15 slots, the sums reach 2^16 (u32 in the exchange)."""
    import json
    monkeypatch.setenv("NIQKI_SHARDS_ON_ONE_DEVICE", "1")
    exp = json.load(open(os.path.join(GOLD, "reference_s16.json")))["cli_s16"]
    run(workdir, ["--gpus", "2", "-I", "fof.txt", "-Q", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "s16g_hits.gz"])
    assert_same_text(gunzip(workdir / "s16g_hits.gz").decode(), exp["hits"])
    run(workdir, ["--gpus", "2", "-M", "fof.txt", "-S", "16", "-W", "8", "-J", "0.1", "-O", "s16g_matrix.gz"])
    assert_same_text(gunzip(workdir / "s16g_matrix.gz").decode(), exp["matrix"])


REF_GPU = os.path.join(ROOT, "oracle", "_ref", "niqki_ref_gpu")


def _normalised(text):
    """the lines as sorted bags of tokens, themselves sorted: what does not depend on the order in which threads
    were handed their genome ids (ties between equal counts are listed by descending id) or wrote their lines"""
    return sorted(tuple(sorted(t for t in line.split(" ") if t)) for line in text.split("\n") if line.strip())


@pytest.mark.skipif(not os.path.exists(REF_GPU), reason="oracle/_ref/niqki_ref_gpu is built where /root/reference exists (oracle/Makefile)")
def test_the_reference_program_itself_on_the_c_abi(workdir, gold):
    """The drop-in, end to end: the REFERENCE's own main, option parser, file readers, `omp parallel` record loops and
    writers (oracle/_ref/libniqki_ref.so + src/niqki.cpp, compiled where they lie) with compute_sketch / insert_sketch /
    query_sketch bound to libniqki_hip.so (oracle/ref_gpu_ops.cpp: symbol precedence, no reference file modified --
    INTEGRATION.md's minimal patch).  One thread: the reference CLI's golden texts byte for byte.  Eight threads (the
    reference then hands out genome ids in thread arrival order; the *_shared entry points combine
    the callers into batches): the same hits per query."""
    _, meta = gold

    import re

    def run_ref(args, threads):
        r = subprocess.run([REF_GPU] + args, cwd=workdir, capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, OMP_NUM_THREADS=str(threads), NIQKI_REF_GPU_REPORT="1"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        # the binding took: the operators' calls were answered by the library (else the reference's CPU code
        # would have produced the same text)
        m = re.search(r"niqki_ref_gpu: (\d+) calls .* in (\d+) batches \(largest (\d+)\), (\d+) genomes indexed on the GPU", r.stderr)
        assert m, r.stderr[-2000:]
        calls, batches, largest, indexed = (int(x) for x in m.groups())
        assert calls >= 2 * indexed > 0 and batches >= 1        # a sketch + an insert per indexed record (+ a sketch and a query per query)
        return calls, batches, largest
    run_ref(["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "refgpu_hits.gz"], 1)
    assert_same_text(gunzip(workdir / "refgpu_hits.gz").decode(), meta["cli"]["hits"])
    run_ref(["-i", "reads.fa", "-l", "reads.fa", "-S", "10", "-W", "10", "-J", "0.2", "-O", "refgpu_lines.gz"], 1)
    assert_same_text(gunzip(workdir / "refgpu_lines.gz").decode(), meta["cli"]["lines"])
    # --matrix and --dump walk the reference's own bucket vectors: their counting loop / payload are bound too
    # (the reference's output_matrix still formats the rows)
    run_ref(["-M", "fof.txt", "-S", "10", "-O", "refgpu_matrix.gz"], 1)
    assert_same_text(gunzip(workdir / "refgpu_matrix.gz").decode(), meta["cli"]["matrix"])
    run_ref(["-I", "fof.txt", "-S", "10", "-J", "0.1", "-O", "refgpu_tmp.gz", "-D", "refgpu.dump"], 1)
    raw = gunzip(workdir / "refgpu.dump")
    assert len(raw) == meta["cli"]["dump_len"] and hashlib.md5(raw).hexdigest() == meta["cli"]["dump_md5"]
    # --load: the reference's own constructor reads the dump, the GPU index is rebuilt from its bucket vectors
    run_ref(["-L", "refgpu.dump", "-Q", "fof.txt", "-O", "refgpu_loaded.gz"], 1)
    assert_same_text(gunzip(workdir / "refgpu_loaded.gz").decode(), meta["cli"]["hits_loaded"])
    calls, batches, largest = run_ref(["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "refgpu_hits8.gz"], 8)
    assert _normalised(gunzip(workdir / "refgpu_hits8.gz").decode()) == _normalised(meta["cli"]["hits"])
    calls, batches, largest = run_ref(["-i", "reads.fa", "-l", "reads.fa", "-S", "10", "-W", "10", "-J", "0.2", "-O", "refgpu_lines8.gz"], 8)
    assert _normalised(gunzip(workdir / "refgpu_lines8.gz").decode()) == _normalised(meta["cli"]["lines"])


@pytest.mark.skipif(not os.path.exists(REF_GPU), reason="oracle/_ref/niqki_ref_gpu is built where /root/reference exists (oracle/Makefile)")
def test_the_reference_program_on_its_own_example_data(tmp_path, gold):
    """BASELINE.json configs[0] through the reference's OWN program on the GPU (operators bound to the C ABI,
    oracle/ref_gpu_ops.cpp): the nine E. coli genomes it ships, default parameters (K=31 S=15 W=12) -- `--matrix` gives
    the README's matrix (README.md:118-128), index + query the reference CLI's hits, `--dump` its bytes."""
    _, meta = gold
    exp = meta["ecoli_cli"]
    edir = os.path.join(ROOT, "tests", "golden", "ecoli")

    def run_ref(args):
        r = subprocess.run([REF_GPU] + args, cwd=edir, capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, OMP_NUM_THREADS="1", NIQKI_REF_GPU_REPORT="1"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        assert "answered by libniqki_hip.so" in r.stderr
    run_ref(["-M", "file_of_file.txt", "-O", str(tmp_path / "m.gz")])
    assert_same_text(gunzip(tmp_path / "m.gz").decode(), exp["matrix"])
    run_ref(["-I", "file_of_file.txt", "-Q", "file_of_file.txt", "-J", "0.8", "-O", str(tmp_path / "h.gz"), "-D", str(tmp_path / "d.gz")])
    assert_same_text(gunzip(tmp_path / "h.gz").decode(), exp["hits_J0.8"])
    h = hashlib.md5()
    n = 0
    with gzip.open(tmp_path / "d.gz", "rb") as f:
        for blk in iter(lambda: f.read(1 << 24), b""):
            h.update(blk)
            n += len(blk)
    assert n == exp["dump_len"] and h.hexdigest() == exp["dump_md5"]


def test_lines_mode_many_reads_sharded_equals_one_gpu(tmp_path, native, monkeypatch):
    """Lines mode at a size where the pipeline really runs (several 8 MB pieces, tens of thousands of entries per call,
    the hit lines written as several gzip members): 300 000 reads of 150 bases against 24 genomes -- `niqki --gpus 8`
    (8 slot shards on the one GPU of the box, every shard framing and sketching its run of each piece) writes the
    bytes `niqki` writes on one GPU, one line per read in the file's order."""
    monkeypatch.setenv("NIQKI_SHARDS_ON_ONE_DEVICE", "1")
    names = []
    for g in range(24):
        seq = native.synth_genome_host(11, g // 4, g % 4, 0 if g % 4 == 0 else 30 * (g % 4), 400_000)
        (tmp_path / ("g%02d.fa" % g)).write_bytes(b">g%02d\n" % g + bytes(seq) + b"\n")
        names.append("g%02d.fa" % g)
    (tmp_path / "fof.txt").write_text("\n".join(names) + "\n")
    rng = np.random.default_rng(5)
    src = native.synth_genome_host(11, 0, 0, 0, 400_000)
    n_reads = 300_000
    st = rng.integers(0, 400_000 - 150, n_reads)
    with open(tmp_path / "reads.fa", "wb") as f:
        for i in range(0, n_reads, 65536):
            f.write(b"".join(b">r%d\n" % (i + j) + bytes(src[s:s + 150]) + b"\n" for j, s in enumerate(st[i:i + 65536])))
    outs = {}
    for gpus in (1, 8):
        run(tmp_path, ["--gpus", str(gpus), "-I", "fof.txt", "-l", "reads.fa", "-S", "12", "-W", "10", "-J", "0.01", "-O", "o%d.gz" % gpus])
        outs[gpus] = gunzip(tmp_path / ("o%d.gz" % gpus))
    assert outs[1] == outs[8]
    lines = outs[1].decode().split("\n")
    assert len(lines) == n_reads + 1 and lines[0].startswith(">r0 ") and lines[n_reads - 1].startswith(">r%d " % (n_reads - 1))
