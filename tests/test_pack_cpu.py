"""CPU: packed FASTA (niqki_pack_fasta / niqki_unpack_fasta, niqki_amd/csrc/nq_pack.h) -- host code of the product
library that needs no device.  A container must give back EXACTLY the file's bytes (the device pass that does the
same on the GPU is checked against raw staging in tests/test_gpu_ingest.py), files that are not worth packing must be
refused (the caller then sends them raw), and a damaged container must be rejected before its offsets are trusted."""
import struct

import numpy as np
import pytest


def fasta(rng, n_bases, width, *, dirty_every=0, lower_every=0, trailing_nl=True, crlf=False, header=b">g some name\n"):
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_bases)].copy()
    if dirty_every:
        seq[::dirty_every] = ord("N")
    if lower_every:
        seq[3::lower_every] |= 0x20
    out = bytearray(header)
    nl = b"\r\n" if crlf else b"\n"
    for a in range(0, n_bases, width):
        out += bytes(seq[a:a + width]) + nl
    if not trailing_nl and out.endswith(nl):
        del out[-len(nl):]
    return bytes(out)


def test_round_trip_and_ratio(native):
    rng = np.random.default_rng(3)
    for width in (16, 17, 31, 32, 33, 60, 63, 64, 65, 70, 80, 127, 128, 200, 1000, 65536):
        data = fasta(rng, 300_000, width)
        c = native.pack_fasta(data)
        assert c is not None, width
        assert bytes(native.unpack_fasta(c)) == data, width
        assert c.size < len(data) / 3.5, (width, c.size, len(data))       # ~4 x: 2 bits per base, lines byte aligned
    # a last line without newline, lines of other widths in between, several records, blank lines
    parts = [fasta(rng, 50_000, 70, trailing_nl=True), b"\n\n", fasta(rng, 30_011, 60, header=b">second\n"), b">third\n",
             fasta(rng, 10_000, 70, header=b"", trailing_nl=False)]
    data = b"".join(parts)
    c = native.pack_fasta(data)
    assert c is not None and bytes(native.unpack_fasta(c)) == data
    # lines holding anything but A C G T travel verbatim, the others packed: still exact, still smaller
    data = fasta(rng, 400_000, 70, dirty_every=5000, lower_every=7001)
    c = native.pack_fasta(data)
    assert c is not None and bytes(native.unpack_fasta(c)) == data and c.size < len(data) / 2


def test_files_not_worth_packing_are_refused(native):
    rng = np.random.default_rng(4)
    reads = b"".join(b">r%d\n" % i + bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 150)]) + b"\n" for i in range(2000))
    for data in (b"", b">only a header\n", b">x\nACGT\n", reads,                       # tiny; too fragmented (a segment per read)
                 fasta(rng, 100_000, 70, crlf=True),                                  # '\r' in every line: nothing packs
                 fasta(rng, 100_000, 70, dirty_every=50),                             # an N in nearly every line
                 fasta(rng, 100_000, 12)):                                            # lines shorter than 16 bases
        assert native.pack_fasta(data) is None, data[:40]


def test_damaged_containers_are_rejected(native):
    rng = np.random.default_rng(5)
    data = fasta(rng, 100_000, 70)
    c = native.pack_fasta(data)
    assert bytes(native.unpack_fasta(c)) == data
    magic, n_seg, raw_len, pay_off, pay_len = struct.unpack_from("<IIQQQ", c.tobytes(), 0)
    assert magic == 0x4B50514E and n_seg >= 2 and raw_len == len(data) and pay_off % 16 == 0 and pay_off + pay_len == c.size
    bad = []
    for off, fmt, val in ((0, "<I", 0x12345678),            # magic
                          (8, "<Q", raw_len + 1),            # raw length that the segments do not add up to
                          (16, "<Q", c.size + 16),           # payload outside the container
                          (24, "<Q", pay_len + 1),
                          (32 + 8, "<Q", 7),                 # first segment's payload offset
                          (32 + 16, "<I", 0),                # an empty segment
                          (32 + 24 + 16, "<I", 0xFFFFFFF0)):  # a line count that overruns the payload
        d = bytearray(c.tobytes())
        struct.pack_into(fmt, d, off, val)
        bad.append(bytes(d))
    bad.append(c.tobytes()[:40])
    bad.append(c.tobytes()[:-1])
    for d in bad:
        with pytest.raises(native.NiqkiError):
            native.unpack_fasta(np.frombuffer(d, np.uint8))


def test_crafted_widths_are_rejected(native):
    """A table whose line width makes the 32-bit `(width + 3) / 4` or `width + 1` of the device pass wrap (ADVICE round 5:
    count = 1, width = 0xFFFFFFFD gave raw_len 4294967294 with an EMPTY payload) must not pass validation: nothing wider
    than the packer's own limit (0xFFFFFF bases per line) is a container."""
    for width in (0xFFFFFFFD, 0xFFFFFFFE, 0xFFFFFFFF, 0x1000000):
        raw_len = width + 1
        pk_len = ((width + 3) & 0xFFFFFFFF) // 4                       # what 32-bit arithmetic made of it
        for pay_len in {pk_len, (width + 3) // 4 if width < 0x2000000 else 0}:
            d = struct.pack("<IIQQQ", 0x4B50514E, 1, raw_len, 64, pay_len) + struct.pack("<QQII", 0, 0, 1, width)
            d += b"\0" * (64 - len(d)) + b"\0" * min(pay_len, 1 << 23)
            with pytest.raises(native.NiqkiError):
                native.unpack_fasta(np.frombuffer(d, np.uint8))
    # the widest line the packer does write is still a container
    rng = np.random.default_rng(6)
    data = fasta(rng, 0xFFFFFF + 40, 0xFFFFFF)
    c = native.pack_fasta(data)
    assert c is not None and bytes(native.unpack_fasta(c)) == data
