"""GPU: gzip members inflated on the device (niqki_amd/csrc/nq_inflate.hip, niqki_gunzip and the NIQKI_FILE_GZIP files
of niqki_stage_raw) against zlib -- what the reference's reader (zstr::ifstream over zlib's inflate, src/zstr.hpp:190-203,
:236-239) makes of the same bytes.  A file the kernel accepts (status 0) must have exactly zlib's bytes; a file zlib
refuses must be refused; and whatever the bytes say, nothing is written outside the file's own output range."""
import gzip
import struct
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fasta(rng, n_bases, width=70, alphabet=b"ACGT", name=b"seq"):
    s = np.frombuffer(alphabet, np.uint8)[rng.integers(0, len(alphabet), n_bases)].tobytes()
    return b">" + name + b" synthetic\n" + b"\n".join(s[a:a + width] for a in range(0, n_bases, width)) + b"\n"


def gz(data, level=6, **kw):
    return gzip.compress(data, compresslevel=level, mtime=0, **kw)


def member(raw_deflate, data, flags=0, extra=b"", name=b"", comment=b"", hcrc=False):
    """a gzip member around a raw DEFLATE stream, with any of the optional header fields (RFC 1952)"""
    flg = (4 if extra else 0) | (8 if name else 0) | (16 if comment else 0) | (2 if hcrc else 0) | flags
    h = b"\x1f\x8b\x08" + bytes([flg]) + b"\0\0\0\0" + b"\x00\x03"
    if extra:
        h += struct.pack("<H", len(extra)) + extra
    if name:
        h += name + b"\0"
    if comment:
        h += comment + b"\0"
    if hcrc:
        h += struct.pack("<H", zlib.crc32(h) & 0xFFFF)
    return h + raw_deflate + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data) & 0xFFFFFFFF)


def raw_deflate(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, memlevel=8):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, memlevel, strategy)
    return c.compress(data) + c.flush()


class Bits:
    """LSB-first bit writer for hand-made DEFLATE streams"""

    def __init__(self):
        self.acc, self.n, self.out = 0, 0, bytearray()

    def put(self, v, n):
        self.acc |= v << self.n
        self.n += n
        while self.n >= 8:
            self.out.append(self.acc & 0xFF)
            self.acc >>= 8
            self.n -= 8

    def code(self, c, n):   # Huffman codes go in most-significant bit first
        self.put(int(format(c, "0%db" % n)[::-1], 2), n)

    def done(self):
        if self.n:
            self.out.append(self.acc & 0xFF)
        return bytes(self.out)


LEN_BASE = [3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258]
LEN_EXTRA = [0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0]
DIST_BASE = [1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097,
             6145, 8193, 12289, 16385, 24577]
DIST_EXTRA = [0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13]


def fixed_block(ops, final=True, bits=None):
    """ops: bytes objects (literals) and (length, distance) pairs -> one fixed-Huffman block (bits: the writer of
    the blocks before it; the stream's bytes come with the final block)"""
    b = bits if bits is not None else Bits()
    b.put(1 if final else 0, 1)
    b.put(1, 2)

    def lit(sym):
        if sym < 144:
            b.code(0x30 + sym, 8)
        elif sym < 256:
            b.code(0x190 + sym - 144, 9)
        elif sym < 280:
            b.code(sym - 256, 7)
        else:
            b.code(0xC0 + sym - 280, 8)

    for op in ops:
        if isinstance(op, (bytes, bytearray)):
            for c in op:
                lit(c)
        elif op[0] == "sym":          # a raw literal/length symbol
            lit(op[1])
        elif op[0] == "dsym":         # a raw distance symbol
            b.code(op[1], 5)
        else:
            ln, dist = op
            i = max(k for k in range(29) if LEN_BASE[k] <= ln) if ln < 258 else 28
            lit(257 + i)
            b.put(ln - LEN_BASE[i], LEN_EXTRA[i])
            j = max(k for k in range(30) if DIST_BASE[k] <= dist)
            b.code(j, 5)
            b.put(dist - DIST_BASE[j], DIST_EXTRA[j])
    lit(256)
    return b.done() if final else b


def expand(ops):
    out = bytearray()
    for op in ops:
        if isinstance(op, (bytes, bytearray)):
            out += op
        else:
            ln, dist = op
            for _ in range(ln):
                out.append(out[-dist])
    return bytes(out)


def zlib_says(blob):
    """what zlib makes of a gzip file: its bytes, or None when it refuses (gzip.decompress reads every member and
    raises on damage, truncation and trailing garbage)"""
    try:
        return gzip.decompress(blob)
    except Exception:
        return None


@pytest.fixture(scope="module", params=["window_in_lds", "last_8k_in_lds"])
def eng(native, request):
    """both forms of the kernel: the file's whole 32 KB window in LDS, or its last 8 KB with far matches read back from
    the file's own output (what launches of more files than the device runs at once take)"""
    e = native.Engine(K=31, S=10, W=10, H=4)
    e.set_option("inflate_window", 0 if request.param == "window_in_lds" else 1)
    yield e
    e.close()


def check(eng, blobs, expects, sizes=None):
    sizes = [len(x) for x in expects] if sizes is None else sizes
    out, status, produced, members, outside = eng.gunzip(blobs, sizes)
    assert outside == 0
    return out, status, produced, members


def test_fasta_levels_and_alignments(eng):
    rng = np.random.default_rng(1)
    datas, blobs = [], []
    for i, level in enumerate((1, 2, 4, 6, 9, 6, 1, 9, 6, 6, 6)):
        d = fasta(rng, int(rng.integers(1000, 400000)) + i, alphabet=b"ACGT" if i % 3 else b"ACGTN", name=b"g%d" % i)
        d += b"x" * i                      # odd sizes: every file starts at another alignment, in and out
        datas.append(d)
        blobs.append(gz(d, level) + b"")
    out, status, produced, members = check(eng, blobs, datas)
    assert status.tolist() == [0] * len(blobs)
    assert all(a == b for a, b in zip(out, datas))
    assert members.tolist() == [1] * len(blobs) and produced.tolist() == [len(d) for d in datas]


def test_stored_fixed_tiny_and_runs(eng):
    rng = np.random.default_rng(2)
    rnd = rng.integers(0, 256, 200000).astype(np.uint8).tobytes()
    datas = [b"", b"A", b"ACGT\n", b"N" * 100000, b"AC" * 60000, b"ACGTTGCA" * 9000 + b"T", rnd,
             (b"abcdefghijklmnopqrstuvwxyz0123456789_" * 3)[:100] * 700, b"\0" * 70000, bytes(range(256)) * 300]
    blobs = [gz(d, 6) for d in datas]
    blobs += [gz(rnd, 0), gz(fasta(rng, 150000), 0), member(raw_deflate(b"hello hello hello", 9, zlib.Z_FIXED), b"hello hello hello")]
    datas += [rnd, gzip.decompress(blobs[-2]), b"hello hello hello"]
    out, status, _, _ = check(eng, blobs, datas)
    assert status.tolist() == [0] * len(blobs)
    assert all(a == b for a, b in zip(out, datas))


def test_long_codes_and_strategies(eng):
    """skewed byte statistics give codes of up to 15 bits (the canonical walk behind the 10-bit table); Z_HUFFMAN_ONLY
    and Z_RLE streams; small memLevel (many short blocks: the tables are rebuilt every few hundred symbols)"""
    rng = np.random.default_rng(3)
    p = 0.5 ** np.arange(1, 257, dtype=np.float64) + 1e-7
    skew = rng.choice(256, 300000, p=p / p.sum()).astype(np.uint8).tobytes()
    text = (b"the quick brown fox jumps over the lazy dog " * 50 + skew[:5000]) * 20
    datas, blobs = [], []
    for d, lvl, strat, ml in ((skew, 6, zlib.Z_DEFAULT_STRATEGY, 8), (skew, 9, zlib.Z_HUFFMAN_ONLY, 8), (text, 6, zlib.Z_RLE, 8),
                              (text, 9, zlib.Z_DEFAULT_STRATEGY, 1), (skew, 1, zlib.Z_FILTERED, 2), (fasta(rng, 200000), 6, zlib.Z_DEFAULT_STRATEGY, 1)):
        datas.append(d)
        blobs.append(member(raw_deflate(d, lvl, strat, ml), d))
    assert all(zlib_says(b) == d for b, d in zip(blobs, datas))
    out, status, _, _ = check(eng, blobs, datas)
    assert status.tolist() == [0] * len(blobs)
    assert all(a == b for a, b in zip(out, datas))


def test_header_fields_and_members(eng):
    rng = np.random.default_rng(4)
    d1, d2, d3 = fasta(rng, 50000), fasta(rng, 12345, name=b"second"), b"tail\n"
    r1 = raw_deflate(d1)
    blobs = [member(r1, d1, name=b"genome.fa"), member(r1, d1, extra=b"AB\x04\x00abcd", comment=b"a comment"),
             member(r1, d1, extra=bytes(range(200)) * 3, name=b"n", comment=b"c", hcrc=True),
             gz(d1) + gz(d2, 9) + gz(d3, 1),                 # three members
             gz(d1) + gz(b"") + gz(d2)]                      # an empty member in between
    datas = [d1, d1, d1, d1 + d2 + d3, d1 + d2]
    assert all(zlib_says(b) == d for b, d in zip(blobs, datas))
    out, status, _, members = check(eng, blobs, datas)
    assert status.tolist() == [0] * 5 and members.tolist() == [1, 1, 1, 3, 3]
    assert all(a == b for a, b in zip(out, datas))


def test_handmade_streams(eng):
    """matches at the window's limits, overlapping copies of every small distance, length 258, and what must be refused:
    a distance before the start of the data, the symbols the fixed code has but DEFLATE forbids, a reserved block type,
    a stored block whose lengths disagree"""
    rng = np.random.default_rng(5)
    first = rng.integers(65, 91, 32768).astype(np.uint8).tobytes()
    good = [
        [first, (258, 32768), (3, 32768), b"xyz", (258, 1), (200, 2), (100, 3), (70, 63), (70, 64), (70, 65), (258, 32767)],
        [b"a", (258, 1), (258, 1), (258, 259), b"b", (5, 1)],
        [b"abc"] + [(3 + k, 1 + (k % 3)) for k in range(256)],
    ]
    blobs, datas = [], []
    for ops in good:
        d = expand(ops)
        blobs.append(member(fixed_block(ops), d))
        datas.append(d)
    # two blocks, the second reaching back into the first; then a second member (a fresh window)
    ops_a, ops_b = [first[:1000]], [(50, 1000), b"q", (258, 1051)]
    d = expand(ops_a + ops_b)
    blobs.append(member(fixed_block(ops_b, bits=fixed_block(ops_a, final=False)), d))
    datas.append(d)
    assert all(zlib_says(b) == d for b, d in zip(blobs, datas))
    out, status, _, _ = check(eng, blobs, datas)
    assert status.tolist() == [0] * len(blobs)
    assert all(a == b for a, b in zip(out, datas))

    def bad_member(stream, n):
        return member(stream, b"\0" * n)
    bad = [
        bad_member(fixed_block([b"abc", (3, 4)]), 6),                          # distance too far back
        member(fixed_block([b"abcdef"]), b"abcdef") + bad_member(fixed_block([(3, 1)]), 3),   # ... into the previous member
        bad_member(fixed_block([b"ab", ("sym", 286)]), 2),                     # literal/length symbol 286
        bad_member(fixed_block([b"ab", ("sym", 257), ("dsym", 30)]), 5),       # distance symbol 30
        bad_member(b"\x07" + b"\0" * 8, 0),                                    # block type 3
        bad_member(b"\x01\x05\x00\xfa\xfe" + b"hello", 5),                     # stored: LEN / NLEN disagree
    ]
    assert all(zlib_says(b) is None for b in bad)
    _, status, produced, _ = check(eng, bad, [b""] * len(bad), sizes=[16] * len(bad))
    assert all(s != 0 for s in status), status


def test_wrong_sizes_and_trailing_bytes(eng):
    rng = np.random.default_rng(6)
    d = fasta(rng, 30000)
    b = gz(d)
    blobs = [b, b, b, b + b"\0" * 30, b + b"trailing garbage that is long enough", b[:-1], b[:len(b) // 2], b[:12], b"", b"\x1f\x8b"]
    sizes = [len(d) - 1, len(d) + 1, len(d), len(d), len(d), len(d), len(d), len(d), 10, 10]
    out, status, produced, _ = check(eng, blobs, None, sizes=sizes)
    assert status[0] == 7 and status[1] == 12 and status[2] == 0 and out[2] == d
    assert all(s != 0 for s in status[3:]), status
    assert all(int(p) <= s for p, s in zip(produced, sizes))


def test_damaged_streams_are_refused_or_right(eng):
    """bit flips, byte swaps and cuts all over valid files: status 0 only with exactly zlib's bytes, zlib's refusals
    refused, nothing written outside, no hang"""
    rng = np.random.default_rng(7)
    base = [(fasta(rng, 60000), 6), (fasta(rng, 20000, alphabet=b"ACGTNacgtn"), 9), (b"the quick brown fox " * 3000, 6),
            (rng.integers(0, 256, 30000).astype(np.uint8).tobytes(), 6), (fasta(rng, 40000), 1)]
    blobs, sizes = [], []
    for d, lvl in base:
        good = bytearray(gz(d, lvl))
        for k in range(60):
            b = bytearray(good)
            what = k % 4
            if what == 0:
                at = int(rng.integers(0, len(b)))
                b[at] ^= 1 << int(rng.integers(0, 8))
            elif what == 1:
                at = int(rng.integers(10, len(b) - 8))
                b[at] = int(rng.integers(0, 256))
            elif what == 2:
                at = int(rng.integers(10, len(b) - 9))
                b[at], b[at + 1] = b[at + 1], b[at]
            else:
                at = int(rng.integers(10, min(len(b), 400)))   # early damage: the code length tables
                b[at] ^= 0xFF
            blobs.append(bytes(b))
            sizes.append(len(d) + (int(rng.integers(-3, 4)) if k % 7 == 0 else 0))
        for cut in (1, 5, 9, 100):
            blobs.append(bytes(good[:-cut]))
            sizes.append(len(d))
    out, status, produced, _ = check(eng, blobs, None, sizes=sizes)
    n_ok = 0
    for b, s, o, st in zip(blobs, sizes, out, status):
        z = zlib_says(b)
        if st == 0:
            assert z is not None and o == z and len(z) == s
            n_ok += 1
        elif z is not None:
            assert len(z) != s or st in (1, 11), (st, len(z), s)   # only a wrong announced size (or a header field we are stricter about)
    assert n_ok < len(blobs) // 2


def test_many_files_one_launch(eng):
    """more files than the device runs at once (four waves per CU), sizes from empty to a megabyte"""
    rng = np.random.default_rng(8)
    datas = [fasta(rng, int(rng.integers(0, 3000)) * (1 + 300 * (i % 97 == 0)), name=b"f%d" % i) for i in range(1500)]
    blobs = [gz(d, 1 + i % 9) for i, d in enumerate(datas)]
    out, status, _, _ = check(eng, blobs, datas)
    assert not status.any()
    assert all(a == b for a, b in zip(out, datas))


def test_launch_form_follows_the_number_of_files(native):
    """option inflate_window -1: a launch of more files than the device keeps resident with whole windows in LDS takes
    the form with the window's last 8 KB there (twice as many resident); same bytes"""
    e = native.Engine(K=31, S=10, W=10, H=4)
    whole, small = e.stat("inflate_files_in_flight"), e.stat("inflate_files_in_flight_8k")
    assert whole >= 256 and small >= 2 * whole
    rng = np.random.default_rng(10)
    d = fasta(rng, 60000)
    # far matches: the second half repeats the first at distances of 20 to 30 KB, beyond what the small form keeps in LDS
    d2 = d[:30000] + d[2000:28000] + d[:30000]
    for n in (3, int(whole) + 5):
        blobs = [gz(d2 if i % 2 else d, 6) for i in range(n)]
        out, status, _, _, outside = e.gunzip(blobs, [len(d2) if i % 2 else len(d) for i in range(n)])
        assert outside == 0 and not status.any()
        assert all(o == (d2 if i % 2 else d) for i, o in enumerate(out))
    e.close()


def bgzf(data, block=65280, tag=b"BC"):
    """what bgzip / htslib write: gzip members of at most 64 KB, each with its size in a 'B' 'C' extra subfield, an
    empty member at the end (tag = b"NQ": this project's own 4-byte size tag, members of any size)"""
    out = bytearray()
    for a in list(range(0, len(data), block)) + [len(data)]:
        piece = data[a:a + block] if a < len(data) else b""
        body = raw_deflate(piece, 6)
        if tag == b"BC":
            total = 18 + len(body) + 8
            extra = b"BC" + struct.pack("<HH", 2, total - 1)
        else:
            total = 20 + len(body) + 8
            extra = b"NQ" + struct.pack("<HI", 4, total)
        out += member(body, piece, extra=extra)
        assert len(out) and (tag != b"BC" or total <= 65536)
    return bytes(out)


def test_files_of_size_tagged_members(native, po):
    """BGZF files (and files with this project's own member tag): niqki_stage_raw cuts them into their members, one
    wavefront each -- same records and sketches as the files' own bytes; a damaged member sends the whole file back"""
    rng = np.random.default_rng(11)
    e = native.Engine(K=31, S=10, W=10, H=4)
    plain = [fasta(rng, 300000 + 7777 * i, name=b"b%d" % i) for i in range(5)]
    info0, _ = e.stage_raw(plain, ["A"] * 5, scattered=True)
    recs0, ent0, _ = e.staged_records()
    sk0 = e.staged_sketch()
    G = native.capi.FILE_GZIP
    files = [bgzf(plain[0]), bgzf(plain[1], tag=b"NQ", block=1 << 20), gz(plain[2]), bgzf(plain[3], block=4000), plain[4]]
    assert all(zlib_says(f) == p for f, p in zip(files[:4], plain[:4]))
    tys = [ord("A") | G] * 4 + [ord("A")]
    info, _ = e.stage_raw(files, tys, scattered=True)
    recs, ent, _ = e.staged_records()
    assert (info.n_entry, info.n_rec, info.seq_bytes) == (info0.n_entry, info0.n_rec, info0.seq_bytes)
    assert recs == recs0 and np.array_equal(ent, ent0) and np.array_equal(e.staged_sketch(), sk0)
    st = e.gunzip_stats()
    assert st["blocks"] >= len(plain[0]) // 65280 + len(plain[3]) // 4000     # (every member is at least one DEFLATE block)
    bad = bytearray(files[0]); bad[len(bad) // 3] ^= 0x20
    with pytest.raises(native.capi.NiqkiError) as ei:
        e.stage_raw([bytes(bad)] + files[1:], tys, scattered=True)
    assert ei.value.code == native.capi.E_GZIP and e.file_status[:5].tolist()[1:] == [0] * 4 and e.file_status[0] != 0
    e.close()


def test_staged_gzip_files_equal_plain_ones(native, po):
    """niqki_stage_raw with NIQKI_FILE_GZIP files: the same records and sketches as the files' own bytes; mixed with
    raw and packed files; a damaged file is reported through file_status and nothing is staged"""
    rng = np.random.default_rng(9)
    e = native.Engine(K=31, S=10, W=10, H=4)
    plain = [fasta(rng, 40000 + 1000 * i, name=b"g%d" % i) for i in range(6)]
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, bytes(b"ACGT"[j % 4] for j in range(i, i + 80)), b"I" * 80) for i in range(200))
    plain.append(fq)
    types = ["A"] * 6 + ["Q"]
    info0, _ = e.stage_raw(plain, types, scattered=True)
    recs0, ent0, _ = e.staged_records()
    sk0 = e.staged_sketch()
    G = native.capi.FILE_GZIP
    zipped = [gz(p, 1 + i) for i, p in enumerate(plain)]
    for files, tys in ((zipped, [ord(t) | G for t in types]),
                       ([zipped[0], plain[1], zipped[2], plain[3], zipped[4], plain[5], zipped[6]],
                        [ord("A") | G, ord("A"), ord("A") | G, ord("A"), ord("A") | G, ord("A"), ord("Q") | G])):
        for prefetch in (None, "this"):
            info, _ = e.stage_raw(files, tys, scattered=True, prefetch=prefetch)
            recs, ent, _ = e.staged_records()
            assert (info.n_entry, info.n_rec, info.seq_bytes) == (info0.n_entry, info0.n_rec, info0.seq_bytes)
            assert recs == recs0 and np.array_equal(ent, ent0)
            assert np.array_equal(e.staged_sketch(), sk0)
    # refusals: a flipped bit in the middle, two members (the trailer's size is the last member's), a cut file
    bad = bytearray(zipped[1]); bad[len(bad) // 2] ^= 4
    for victim, why in ((bytes(bad), None), (zipped[1] + zipped[2], 7), (zipped[1][:-20], None), (b"\x1f\x8b\x08" + b"\0" * 5, 13)):
        files = list(zipped)
        files[1] = victim
        with pytest.raises(native.capi.NiqkiError) as ei:
            e.stage_raw(files, [ord(t) | G for t in types], scattered=True)
        assert ei.value.code == native.capi.E_GZIP
        st = e.file_status[:7].tolist()
        assert st[1] != 0 and st[:1] + st[2:] == [0] * 6, st
        if why:
            assert st[1] == why
        with pytest.raises(native.capi.NiqkiError):
            e.staged_sketch()           # nothing is staged after a refusal
    e.close()
