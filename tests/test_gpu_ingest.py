"""GPU: FASTA / FASTQ framing on the device (niqki_stage_raw + niqki_staged_*,
niqki_amd/csrc/nq_ingest.hip) against the oracle's restatement of the reference's
reader loop (oracle/pyoracle.py frame_records <- Index::Biogetline,
src/niqki_index.cpp:890-941), which test_oracle_golden.py pins on the reference
CLI's own output for the nasty files."""
import numpy as np
import pytest

import os

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("NIQKI_FUZZ_SCALE", "1"))   # one-off campaigns: more fuzz seeds

CHUNK = 8192  # nq::kIngestChunk


def rand_seq(rng, n, alphabet=b"ACGT"):
    return bytes(np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n)])


def make_fasta(rng, n_rec, *, max_line=90, crlf=False, trailing_nl=True, first_gt=True, blank=0.05,
               long_lines=False, weird=True):
    out = bytearray()
    nl = b"\r\n" if crlf else b"\n"
    for r in range(n_rec):
        hdr = b">" if (first_gt or r) else b"x"
        hdr += b"rec%d some text" % r
        if weird and rng.random() < 0.2:
            hdr += b" >inner"
        out += hdr + nl
        n_lines = int(rng.integers(0, 6))
        for _ in range(n_lines):
            if rng.random() < blank:
                out += nl
                continue
            ln = int(rng.integers(1, 30000)) if (long_lines and rng.random() < 0.2) else int(rng.integers(1, max_line))
            line = bytearray(rand_seq(rng, ln, b"ACGTacgtN" if weird else b"ACGT"))
            if weird and ln > 3 and rng.random() < 0.3:
                line[int(rng.integers(1, ln))] = ord(">")   # '>' not at a line start is sequence
            out += bytes(line) + nl
    if not trailing_nl and out.endswith(nl):
        del out[-len(nl):]
    return bytes(out)


def make_fastq(rng, n_rec, *, trailing_nl=True, partial=0, max_len=300):
    out = bytearray()
    for r in range(n_rec):
        n = int(rng.integers(0, max_len))
        s = rand_seq(rng, n)
        out += b"@r%d\n" % r + s + b"\n+\n" + b"I" * n + b"\n"
    if partial:
        tail = b"@last\n" + rand_seq(rng, 80) + b"\n+\n" + b"I" * 80 + b"\n"
        out += b"\n".join(tail.split(b"\n")[:partial])
    if not trailing_nl and out.endswith(b"\n"):
        del out[-1:]
    return bytes(out)


def check_whole(native, po, files, types, K=31, S=8):
    e = native.Engine(K=K, S=S, W=10, H=3, J=0.0)
    info, _ = e.stage_raw(files, types, lines=False)
    assert info.n_entry == len(files) and info.consumed == sum(len(f) for f in files)
    recs, entry_rec, hdr_pos = e.staged_records()
    assert entry_rec[0] == 0 and entry_rec[-1] == info.n_rec
    base = 0
    want_sk = []
    p = po.make_params(K, S, 10, 3, 0.0)
    for f, (data, ty) in enumerate(zip(files, types)):
        exp = po.frame_records(data, ty, K)
        got = [(int(hdr_pos[r]) - base, recs[r]) for r in range(entry_rec[f], entry_rec[f + 1]) if len(recs[r]) > K]
        assert [(a, s) for a, _, s in exp] == got, "file %d (%s, %d bytes)" % (f, ty, len(data))
        # every framed record is delimited where the reference would put it
        sk = np.full(1 << S, -1, np.int32)
        for _, _, s in exp:
            po.sketch_accumulate(p, s, sk)
        want_sk.append(po.densify(p, sk)[0])
        base += len(data)
    got_sk = e.staged_sketch()
    for f in range(len(files)):
        # files whose records fill no slot at all stay empty; everything else equals
        # accumulate-all-records-then-densify (one densification per file)
        assert np.array_equal(got_sk[f], want_sk[f]), f
    e.close()


def test_fasta_framing_edge_cases(native, po):
    rng = np.random.default_rng(1)
    files = [
        b"",
        b"\n",
        b">only header",
        b">h\n",
        b">h\nACGT",
        b"ACGTACGTACGTACGTACGTACGTACGTACGTACGTACGT\n>second\n" + rand_seq(rng, 100) + b"\n",   # first line is a header whatever it holds
        b">a\n>b\n>c\n" + rand_seq(rng, 64) + b"\n>d\n\n\n" + rand_seq(rng, 50) + b"\n\n",
        b">h\n" + rand_seq(rng, 40) + b"\n\xffnot a sequence line\n" + rand_seq(rng, 60) + b"\n",     # 0xFF acts like '>'
        make_fasta(rng, 30),
        make_fasta(rng, 30, crlf=True),
        make_fasta(rng, 30, trailing_nl=False),
        make_fasta(rng, 10, first_gt=False),
        make_fasta(rng, 40, long_lines=True),
    ]
    check_whole(native, po, files, ["A"] * len(files))


@pytest.mark.parametrize("size", [CHUNK - 1, CHUNK, CHUNK + 1, 2 * CHUNK, 2 * CHUNK + 31, 3 * CHUNK - 33])
def test_chunk_boundaries(native, po, size):
    """Newlines, '>' and file ends placed on every offset around the 8 KB chunk edge."""
    rng = np.random.default_rng(size)
    files = []
    for shift in range(0, 40, 3):
        body = bytearray(b">h0\n" + rand_seq(rng, size + 64))
        # a header line straddling the chunk edge, a newline right at it, '>' right after it
        edge = CHUNK - 20 + shift
        body[edge:edge + 12] = b"\n>hdr line\n\n"
        body[size - 1 - (shift % 5)] = ord("\n")
        files.append(bytes(body[:size]))
    check_whole(native, po, files, ["A"] * len(files))


@pytest.mark.parametrize("seed", range(6 * SCALE))
def test_framing_fuzz(native, po, seed):
    """Files of random bytes from an alphabet rich in newlines, '>', 0xFF, '\\r' and NULs, as
    FASTA and as FASTQ: records are framed where the reference's reader loop frames them."""
    rng = np.random.default_rng(100 + seed)
    alphabet = np.frombuffer(b"ACGTACGTACGTacgtN\n\n>\xff\r\x00@+ ", np.uint8)
    files, types = [], []
    for i in range(24):
        n = int(rng.integers(0, 40000)) if i % 4 else int(rng.integers(0, 300))
        p_nl = float(rng.choice([0.002, 0.02, 0.2]))
        body = alphabet[rng.integers(0, alphabet.size, n)].copy()
        body[rng.random(n) < p_nl] = ord("\n")
        files.append(bytes(body))
        types.append("A" if i % 2 == 0 else "Q")
    check_whole(native, po, files, types, K=int(rng.integers(3, 32)), S=6)


def test_single_line_genome_and_many_small_files(native, po):
    rng = np.random.default_rng(3)
    big = b">g\n" + rand_seq(rng, 300_000) + b"\n"           # one 300 kbp line
    wrapped = b">g2 wrapped\n" + b"\n".join(rand_seq(rng, 70) for _ in range(3000)) + b"\n"
    small = [b">s%d\n" % i + rand_seq(rng, int(rng.integers(0, 200))) + b"\n" for i in range(300)]
    check_whole(native, po, [big, wrapped] + small, ["A"] * 302, S=10)


def test_fastq_framing(native, po):
    rng = np.random.default_rng(4)
    files = [
        make_fastq(rng, 50),
        make_fastq(rng, 50, trailing_nl=False),
        make_fastq(rng, 20, partial=1),
        make_fastq(rng, 20, partial=2),     # header + sequence of a last record without its '+' lines: still a record
        make_fastq(rng, 20, partial=3),
        make_fastq(rng, 400, max_len=120),  # several chunks
        b"",
        b"@h\nACGT",
    ]
    check_whole(native, po, files, ["Q"] * len(files))
    # FASTA and FASTQ files in one batch
    check_whole(native, po, [files[0], make_fasta(rng, 20), files[5]], ["Q", "A", "Q"])


@pytest.mark.parametrize("ty", ["A", "Q"])
def test_lines_mode_stream(native, po, ty):
    """A file handed over in pieces (final only on the last), entries capped per call:
    the entries, their header lines and their sketches are those of the reference's
    record loop, whatever the piece size is."""
    rng = np.random.default_rng(5)
    data = make_fasta(rng, 300, weird=True) if ty == "A" else make_fastq(rng, 500)
    K, S = 21, 8
    exp = po.frame_records(data, ty, K)
    p = po.make_params(K, S, 10, 3, 0.0)
    exp_sk = np.stack([po.densify(p, po.sketch_accumulate(p, s))[0] for _, _, s in exp])
    for piece, cap in ((len(data) + 10, 1 << 20), (20000, 64), (3000, 7), (9000, 1000)):
        e = native.Engine(K=K, S=S, W=10, H=3, J=0.0)
        pos, want, got_hdr, got_sk = 0, piece, [], []
        for _ in range(100000):
            end = min(len(data), pos + want)
            final = end == len(data)
            info, hdr = e.stage_raw([data[pos:end]], [ty], lines=True, final=final, max_entries=cap)
            if info.n_entry:
                got_sk.append(e.staged_sketch())
                got_hdr += [pos + int(h) for h in hdr]
            if final and info.consumed == end - pos:
                break
            if info.consumed == 0:   # not even one complete record in the piece
                assert not final
                want *= 2
                continue
            pos += int(info.consumed)
            want = piece
        else:
            raise AssertionError("stream did not terminate")
        assert got_hdr == [a for a, _, _ in exp], (piece, cap)
        sk = np.concatenate(got_sk) if got_sk else np.zeros((0, 1 << S), np.int32)
        assert np.array_equal(sk, exp_sk)
        # the header text is the line at the reported offset
        for a, h, _ in exp[:50]:
            j = data.find(b"\n", a)
            assert data[a:j if j >= 0 else len(data)] == h
        e.close()


def test_staged_insert_and_query_equal_the_record_path(native, po):
    rng = np.random.default_rng(6)
    genomes = [rand_seq(rng, 30_000) for _ in range(12)]
    for i in range(1, 12):   # relatives of genome 0
        g = bytearray(genomes[0])
        for j in rng.integers(0, len(g), 40 * i):
            g[j] = ord("ACGT"[int(rng.integers(0, 4))])
        genomes[i] = bytes(g)
    files = [b">g%d\n" % i + b"\n".join(g[a:a + 60] for a in range(0, len(g), 60)) + b"\n" for i, g in enumerate(genomes)]
    a = native.Engine(K=31, S=10, W=12, H=4, J=0.05)
    info, _ = a.stage_raw(files, None)
    a.staged_insert()
    assert a.n_genomes == 12
    b = native.Engine(K=31, S=10, W=12, H=4, J=0.05)
    b.insert(b.sketch(genomes))
    assert np.array_equal(a.get_sketches(0, 12), b.get_sketches(0, 12))
    a.stage_raw(files[:5], None)
    off, hc, hg = a.staged_query(capacity=3)   # forces the capacity retry on the staged batch
    off2, hc2, hg2 = b.query_sequences(genomes[:5])
    assert np.array_equal(off, off2) and np.array_equal(hc, hc2) and np.array_equal(hg, hg2)
    assert off[-1] > 5
    # other calls in between do not disturb the staged batch ...
    a.stage_raw(files[:5], None)
    sk5 = a.staged_sketch()
    a.query(b.get_sketches(0, 12))
    a.densify(np.full((3, 1024), -1, np.int32))
    off3, hc3, hg3 = a.staged_query()
    assert np.array_equal(off3, off2) and np.array_equal(hc3, hc2) and np.array_equal(hg3, hg2)
    assert np.array_equal(a.staged_sketch(), sk5)
    # ... except a host-memory sketch call, which takes the staging buffers: reported, not wrong
    a.sketch(genomes[:2])
    with pytest.raises(native.NiqkiError):
        a.staged_query()
    # lines mode on the same bytes: one entry per record
    reads = b"".join(b">r%d\n" % i + genomes[i % 12][100 * i:100 * i + 150] + b"\n" for i in range(200))
    info, hdr = a.stage_raw([reads], None, lines=True, final=True, max_entries=4096)
    assert info.n_entry == 200 and info.consumed == len(reads)
    off, hc, hg = a.staged_query()
    off2, hc2, hg2 = b.query_sequences([genomes[i % 12][100 * i:100 * i + 150] for i in range(200)])
    assert np.array_equal(off, off2) and np.array_equal(hc, hc2) and np.array_equal(hg, hg2)
    a.close()
    b.close()


def test_stage_errors(native):
    e = native.Engine(K=31, S=8, W=10, H=3, J=0.0)
    with pytest.raises(native.NiqkiError):
        e._ck(e.L.niqki_staged_insert(e.h))          # nothing staged yet
    with pytest.raises(native.NiqkiError):
        e.stage_raw([b">a\nACGT\n", b">b\nACGT\n"], None, lines=True)      # lines mode: one file
    with pytest.raises(native.NiqkiError):
        e.stage_raw([b">a\nACGT\n"], ["X"])
    info, _ = e.stage_raw([], None)
    assert info.n_entry == 0 and info.n_rec == 0
    assert e.staged_sketch().shape == (0, 256)
    e.close()


def test_prefetched_batches_equal_plain_staging(native, po):
    """niqki_stage_raw_prefetch: the bytes of the next batch travel on the copy stream while the staged
    batch is worked on; taken by the matching niqki_stage_raw, dropped by any other; sketches unchanged."""
    rng = np.random.default_rng(16)
    batches = []
    for b in range(4):
        n = int(rng.integers(3, 9))
        batches.append([b">f%d_%d\n" % (b, i) + b"\n".join(
            s[a:a + 70] for s in [rand_seq(rng, int(rng.integers(2_000, 60_000)))] for a in range(0, len(s), 70)) + b"\n"
            for i in range(n)])
    ref = native.Engine(K=31, S=10, W=12, H=4, J=0.05)
    want = []
    for files in batches:
        ref.stage_raw(files, None)
        want.append(ref.staged_sketch())
    e = native.Engine(K=31, S=10, W=12, H=4, J=0.05)
    # the host program's sequence: stage(i) takes prefetch(i), prefetch(i+1) runs beside the work on i
    info, _ = e.stage_raw(batches[0], None, scattered=True, prefetch="this")
    assert info.n_entry == len(batches[0])
    for i in range(4):
        if i:
            info, _ = e.stage_raw(batches[i], None, scattered=True, prefetch="take")
            assert info.n_entry == len(batches[i])
        if i + 1 < 4:
            e.stage_raw(batches[i + 1], None, scattered=True, prefetch="only")
        assert np.array_equal(e.staged_sketch(), want[i]), i
        e.staged_insert()
    assert e.n_genomes == sum(len(b) for b in batches)
    # a prefetch nobody takes: a different batch is staged by copy, the next prefetch replaces an unused one
    e.stage_raw(batches[2], None, scattered=True, prefetch="only")
    e.stage_raw(batches[1], None, scattered=True)
    assert np.array_equal(e.staged_sketch(), want[1])
    e.stage_raw(batches[0], None, scattered=True, prefetch="only")
    e.stage_raw(batches[3], None, scattered=True, prefetch="only")
    e.stage_raw(batches[3], None)                       # contiguous form: not the prefetched pointers
    assert np.array_equal(e.staged_sketch(), want[3])
    e.stage_raw([], None, scattered=True, prefetch="this")
    e.close()
    ref.close()


def test_packed_fasta_files_stage_like_their_raw_bytes(native, po):
    """file_type 'a': a FASTA file handed over as its packed container (niqki_pack_fasta: 2 bits per base in full
    A/C/G/T lines, the rest verbatim).  The device restores the file's bytes before it frames them, so the staged
    records, header positions and sketches must be those of the raw file -- for clean genomes of several line widths,
    files with dirty lines in between, several records, no final newline, and in batches that mix packed and raw
    files; with the bytes copied by the call, and with niqki_stage_raw_prefetch having put them on their way."""
    rng = np.random.default_rng(17)

    def genome(n, width, **kw):
        seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        for at in kw.get("dirty", ()):
            seq[at] = ord("N")
        out = bytearray(kw.get("header", b">g text\n"))
        for a in range(0, n, width):
            out += bytes(seq[a:a + width]) + b"\n"
        if kw.get("no_final_nl"):
            del out[-1:]
        return bytes(out)
    files = [genome(200_000, 70), genome(123_457, 60, no_final_nl=True), genome(90_000, 80, dirty=(5, 40_000, 89_999)),
             genome(50_000, 70) + genome(70_001, 70, header=b">second record\n") + b"\n\n" + genome(20_000, 33, header=b">third\n"),
             genome(40_000, 16), genome(66_000, 1000), genome(300_000, 65536), genome(17_000, 127, header=b"no gt in the first line\n"),
             make_fasta(rng, 30), genome(8000, 70)]
    packed = [native.pack_fasta(f) for f in files]
    assert sum(c is not None for c in packed) >= 8 and packed[8] is None       # (the record soup is not worth packing)
    wire = [bytes(c) if c is not None else f for c, f in zip(packed, files)]
    types = ["a" if c is not None else "A" for c in packed]
    K, S = 31, 8
    ref = native.Engine(K=K, S=S, W=10, H=3, J=0.0)
    info_r, _ = ref.stage_raw(files, ["A"] * len(files))
    recs_r, er_r, hp_r = ref.staged_records()
    sk_r = ref.staged_sketch()
    for mode in ("copy", "prefetch"):
        e = native.Engine(K=K, S=S, W=10, H=3, J=0.0)
        info, _ = e.stage_raw(wire, types, scattered=True, prefetch="this" if mode == "prefetch" else None)
        assert (info.n_entry, info.n_rec, info.seq_bytes) == (info_r.n_entry, info_r.n_rec, info_r.seq_bytes), mode
        assert info.consumed == sum(len(f) for f in files)                      # raw bytes, whatever travelled
        recs, er, hp = e.staged_records()
        assert recs == recs_r and np.array_equal(er, er_r) and np.array_equal(hp, hp_r), mode
        assert np.array_equal(e.staged_sketch(), sk_r), mode
        # a batch of packed files only, then a raw batch on the same handle (buffers change roles)
        sel = [i for i, c in enumerate(packed) if c is not None][:3]
        info2, _ = e.stage_raw([wire[i] for i in sel], ["a"] * 3, scattered=True)
        r2, _, _ = e.staged_records()
        info3, _ = ref.stage_raw([files[i] for i in sel], ["A"] * 3)
        r3, _, _ = ref.staged_records()
        assert r2 == r3 and info2.n_rec == info3.n_rec
        info4, _ = e.stage_raw(files[:2], ["A", "A"], scattered=True, prefetch="this")
        r4, _, _ = e.staged_records()
        info5, _ = ref.stage_raw(files[:2], ["A", "A"])
        r5, _, _ = ref.staged_records()
        assert r4 == r5
        e.close()
    # what is not allowed: a damaged container, lines mode, the one-buffer form
    e = native.Engine(K=K, S=S, W=10, H=3, J=0.0)
    bad = bytearray(wire[0])
    bad[8] ^= 1
    for kw, fl, ty in ((dict(scattered=True), [bytes(bad)], ["a"]), (dict(scattered=True, lines=True), [wire[0]], ["a"]),
                       (dict(), [wire[0]], ["a"])):
        with pytest.raises(native.NiqkiError) as ei:
            e.stage_raw(fl, ty, **kw)
        assert ei.value.code == 1
    e.close()
    ref.close()
