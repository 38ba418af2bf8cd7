"""ctypes access to the parity oracle (oracle/liboracle.so) and, when built, to
the real reference (oracle/_ref/libniqki_ref.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (niqki_amd) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_LIB_PATH = os.path.join(_HERE, "_ref", "libniqki_ref.so")
REF_BIN_PATH = os.path.join(_HERE, "_ref", "niqki_ref")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


class Params(C.Structure):
    _fields_ = [("K", C.c_uint32), ("S", C.c_uint32), ("W", C.c_uint32),
                ("H", C.c_uint32), ("min_score", C.c_uint32), ("H0p1", C.c_uint32)]


class _Index(C.Structure):
    _fields_ = [("p", Params), ("n_genomes", C.c_uint32),
                ("n_buckets", C.c_uint64), ("offsets", u64p), ("gids", u32p)]


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(LIB_PATH) or (
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "niqki_oracle.c"))):
        subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/src") and not os.path.exists(REF_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.nqo_min_score.restype = C.c_uint32
        L.nqo_min_score.argtypes = [C.c_double, C.c_uint32]
        for f in (L.nqo_rev64, L.nqo_unrev64):
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64]
        L.nqo_fingerprint.restype = C.c_int32
        L.nqo_fingerprint.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
        L.nqo_fingerprint_stale.restype = C.c_int32
        L.nqo_fingerprint_stale.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.nqo_select_best_H.restype = C.c_uint32
        L.nqo_select_best_H.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32]
        L.nqo_hash_family.restype = C.c_uint64
        L.nqo_hash_family.argtypes = [C.c_uint64, C.c_uint32]
        L.nqo_sketch_accumulate.restype = C.c_uint64
        L.nqo_sketch_accumulate.argtypes = [C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p]
        L.nqo_densify.restype = C.c_int64
        L.nqo_densify.argtypes = [C.POINTER(Params), C.c_void_p]
        L.nqo_compute_sketch.restype = C.c_int64
        L.nqo_compute_sketch.argtypes = [C.POINTER(Params), C.c_void_p, C.c_uint64, C.c_void_p]
        L.nqo_index_build.restype = C.POINTER(_Index)
        L.nqo_index_build.argtypes = [C.POINTER(Params), C.c_void_p, C.c_uint32]
        L.nqo_index_build_mt.restype = C.POINTER(_Index)
        L.nqo_index_build_mt.argtypes = [C.POINTER(Params), C.c_void_p, C.c_uint32, C.c_int]
        L.nqo_index_free.restype = None
        L.nqo_index_free.argtypes = [C.POINTER(_Index)]
        L.nqo_query_counts.restype = None
        L.nqo_query_counts.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_void_p]
        L.nqo_hits_from_counts.restype = C.c_uint32
        L.nqo_hits_from_counts.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32]
        L.nqo_query_gathered.restype = C.c_uint64
        L.nqo_query_gathered.argtypes = [C.POINTER(_Index), C.c_void_p]
        L.nqo_matrix_range.restype = None
        L.nqo_matrix_range.argtypes = [C.POINTER(_Index), C.c_uint32, C.c_uint32, C.c_void_p]
        L.nqo_dump_bytes.restype = C.c_uint64
        L.nqo_dump_bytes.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint64]
        L.nqo_load_bytes.restype = C.POINTER(_Index)
        L.nqo_load_bytes.argtypes = [C.c_void_p, C.c_uint64, u64p]
        L.nqo_sketch_batch.restype = None
        L.nqo_sketch_batch.argtypes = [C.POINTER(Params), C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
        L.nqo_query_batch.restype = C.c_uint64
        L.nqo_query_batch.argtypes = [C.POINTER(_Index), C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        L.nqo_max_threads.restype = C.c_int
        L.nqo_index_spread.restype = None
        L.nqo_index_spread.argtypes = [C.POINTER(_Index), C.c_int]
        L.nqo_fnv1a64.restype = C.c_uint64
        L.nqo_fnv1a64.argtypes = [C.c_void_p, C.c_uint64]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _seq(seq):
    if isinstance(seq, str):
        seq = seq.encode()
    if isinstance(seq, (bytes, bytearray)):
        return np.frombuffer(bytes(seq), dtype=np.uint8)
    return np.ascontiguousarray(seq, dtype=np.uint8)


def make_params(K=31, S=15, W=12, H=4, J=0.0, genome_size=0.0):
    """genome_size != 0 applies the reference's -G (select_best_H after the constructor)."""
    p = Params(K, S, W, H, lib().nqo_min_score(J, S), 0)
    if genome_size:
        p.H0p1 = H + 1
        p.H = lib().nqo_select_best_H(genome_size, S, W, H)
    return p


def select_best_H(genome_size, S, W, H):
    return lib().nqo_select_best_H(genome_size, S, W, H)


def fingerprint_stale(h, W, H, H0):
    return lib().nqo_fingerprint_stale(h, W, H, H0)


def rev64(x):
    return lib().nqo_rev64(x)


def unrev64(x):
    return lib().nqo_unrev64(x)


def fingerprint(h, W=12, H=4):
    return lib().nqo_fingerprint(h, W, H)


def fnv1a64(arr):
    a = np.ascontiguousarray(arr)
    return lib().nqo_fnv1a64(_ptr(a), a.nbytes)


def sketch_accumulate(p, seq, sk=None):
    s = _seq(seq)
    F = 1 << p.S
    if sk is None:
        sk = np.full(F, -1, dtype=np.int32)
    lib().nqo_sketch_accumulate(C.byref(p), _ptr(s), s.size, _ptr(sk))
    return sk


def densify(p, sk):
    sk = np.ascontiguousarray(sk, dtype=np.int32).copy()
    rc = lib().nqo_densify(C.byref(p), _ptr(sk))
    return sk, rc


def compute_sketch(p, seq):
    s = _seq(seq)
    sk = np.empty(1 << p.S, dtype=np.int32)
    lib().nqo_compute_sketch(C.byref(p), _ptr(s), s.size, _ptr(sk))
    return sk


def sketch_batch(p, seqs, rec_off, threads=0):
    seqs = np.ascontiguousarray(seqs, dtype=np.uint8)
    rec_off = np.ascontiguousarray(rec_off, dtype=np.uint64)
    n = rec_off.size - 1
    out = np.empty((n, 1 << p.S), dtype=np.int32)
    lib().nqo_sketch_batch(C.byref(p), _ptr(seqs), _ptr(rec_off), n, _ptr(out), threads)
    return out


def cpu_threads():
    """CPUs this process may really use: the affinity mask, cut to the cgroup's CPU quota where there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    try:
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            n = max(1, min(n, int(round(float(q[0]) / float(q[1])))))
    except (OSError, ValueError, IndexError):
        pass
    return n


class Index:
    """CSR inverted index built by the oracle from an (n, F) int32 sketch array."""

    def __init__(self, p=None, sketches=None, handle=None, threads=None):
        """threads: builders (slot ranges; the arrays are those of one thread).  None = what this process has
        (cpu_threads()) for a large index, one for a small one; 1 = the plain single-threaded build."""
        self._L = lib()
        if handle is not None:
            self._h = handle
        else:
            sk = np.ascontiguousarray(sketches, dtype=np.int32)
            assert sk.ndim == 2 and sk.shape[1] == (1 << p.S)
            if threads is None:
                threads = cpu_threads() if sk.size >= (1 << 24) else 1
            self._h = self._L.nqo_index_build_mt(C.byref(p), _ptr(sk), sk.shape[0], int(threads))
        if not self._h:
            raise MemoryError("oracle index build failed")
        self.p = self._h.contents.p
        self.n = self._h.contents.n_genomes

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nqo_index_free(self._h)
            self._h = None

    @property
    def n_buckets(self):
        return self._h.contents.n_buckets

    def offsets(self):
        return np.ctypeslib.as_array(self._h.contents.offsets, shape=(self.n_buckets + 1,)).copy()

    def gids(self):
        tot = int(self._h.contents.offsets[self.n_buckets])
        if tot == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(self._h.contents.gids, shape=(tot,)).copy()

    def counts(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.int32)
        out = np.empty(self.n, dtype=np.uint32)
        self._L.nqo_query_counts(self._h, _ptr(sk), _ptr(out))
        return out

    def gathered(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.int32)
        return self._L.nqo_query_gathered(self._h, _ptr(sk))

    def query(self, sk, min_score=None):
        """-> (counts, gids) of the hits, reference order (count desc, gid desc)."""
        c = self.counts(sk)
        ms = self.p.min_score if min_score is None else min_score
        hc = np.empty(self.n, dtype=np.uint32)
        hg = np.empty(self.n, dtype=np.uint32)
        nh = self._L.nqo_hits_from_counts(_ptr(c), self.n, ms, _ptr(hc), _ptr(hg), self.n)
        return hc[:nh].copy(), hg[:nh].copy()

    def spread(self, threads=0):
        """CPU baseline: the CSR arrays first touched by `threads` threads in equal parts (NUMA placement)."""
        self._L.nqo_index_spread(self._h, threads)

    def query_batch(self, sketches, threads=0):
        sk = np.ascontiguousarray(sketches, dtype=np.int32)
        n = sk.shape[0]
        cap = n * self.n
        off = np.empty(n + 1, dtype=np.uint64)
        hc = np.empty(cap, dtype=np.uint32)
        hg = np.empty(cap, dtype=np.uint32)
        tot = self._L.nqo_query_batch(self._h, _ptr(sk), n, _ptr(off), _ptr(hc), _ptr(hg), cap, threads)
        return off, hc[:tot].copy(), hg[:tot].copy()

    def matrix_range(self, begin, end):
        out = np.empty((self.n, end - begin), dtype=np.uint16)
        self._L.nqo_matrix_range(self._h, begin, end, _ptr(out))
        return out

    def dump_bytes(self):
        n = self._L.nqo_dump_bytes(self._h, None, 0)
        buf = np.empty(n, dtype=np.uint8)
        self._L.nqo_dump_bytes(self._h, _ptr(buf), n)
        return buf.tobytes()

    @classmethod
    def load_bytes(cls, data):
        buf = np.frombuffer(data, dtype=np.uint8)
        consumed = C.c_uint64(0)
        h = lib().nqo_load_bytes(_ptr(buf), buf.size, C.byref(consumed))
        if not h:
            raise ValueError("bad dump")
        ix = cls(handle=h)
        ix.names_offset = consumed.value
        return ix


# ---- record framing ----------------------------------------------------------

class _Stream:
    """std::istream over a byte string, as far as getline / peek / eof go."""

    def __init__(self, data):
        self.d, self.pos, self.eofbit, self.failbit = bytes(data), 0, False, False

    def getline(self, old):
        if self.failbit or self.eofbit:          # sentry fails: the string keeps its content
            self.failbit = True
            return old
        if self.pos >= len(self.d):              # nothing extracted
            self.eofbit = self.failbit = True
            return b""
        j = self.d.find(b"\n", self.pos)
        if j < 0:
            line, self.pos, self.eofbit = self.d[self.pos:], len(self.d), True
        else:
            line, self.pos = self.d[self.pos:j], j + 1
        return line

    def peek(self):
        if self.failbit or self.eofbit:
            self.failbit = self.failbit or False
            return -1
        if self.pos >= len(self.d):
            self.eofbit = True
            return -1
        b = self.d[self.pos]
        return b - 256 if b >= 128 else b        # `char c = in->peek()` is signed: 0xFF == EOF


def frame_records(data, type_, K):
    """`while(!in.eof()) Biogetline(...)` of src/niqki_index.cpp:890-941 and its callers
    (:390-403, :446-453): the (header line offset, header, sequence) of every record
    LONGER THAN K, in file order."""
    st = _Stream(data)
    out = []
    header = result = discard = b""
    while not st.eofbit:
        result = b""
        hdr_at = st.pos
        if type_ == "Q":
            header = st.getline(header)
            result = st.getline(result)
            discard = st.getline(discard)
            discard = st.getline(discard)
        else:
            header = st.getline(header)
            c = st.peek()
            while c != ord(">") and c != -1:
                discard = st.getline(discard)
                result += discard
                c = st.peek()
        if len(result) > K:
            out.append((hdr_at, header, result))
    return out


# ---- the real reference (only where oracle/_ref was built) ------------------

def have_ref():
    return os.path.exists(REF_LIB_PATH)


class Ref:
    """The reference's Index class through oracle/ref_harness.cpp."""

    def __init__(self, K=31, S=15, W=12, H=4, J=0.0, out_path="/tmp/niqki_ref_scratch.gz"):
        L = C.CDLL(REF_LIB_PATH)
        L.ref_create.restype = C.c_void_p
        L.ref_create.argtypes = [C.c_uint32] * 4 + [C.c_char_p, C.c_double]
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_min_score.restype = C.c_uint32
        L.ref_min_score.argtypes = [C.c_void_p]
        for f in (L.ref_rev64, L.ref_unrev64):
            f.restype = C.c_uint64
            f.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_fingerprint.restype = C.c_int32
        L.ref_fingerprint.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_hash_family.restype = C.c_uint64
        L.ref_hash_family.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
        L.ref_compute_sketch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
        L.ref_densify.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.ref_insert_sketch.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        L.ref_query_sketch.restype = C.c_uint32
        L.ref_query_sketch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
        L.ref_bucket_size.restype = C.c_uint64
        L.ref_bucket_size.argtypes = [C.c_void_p, C.c_uint64]
        L.ref_dump.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_select_best_H.restype = C.c_uint32
        L.ref_select_best_H.argtypes = [C.c_void_p, C.c_double]
        if hasattr(L, "ref_query_batch"):
            L.ref_sketch_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int]
            L.ref_insert_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
            L.ref_query_batch.restype = C.c_uint64
            L.ref_query_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int]
        self._L = L
        self.S, self.K, self.W, self.H = S, K, W, H
        self.F = 1 << S
        self.n = 0
        self._h = L.ref_create(S, K, W, H, out_path.encode(), J)

    def close(self):
        if self._h:
            self._L.ref_destroy(self._h)
            self._h = None

    def min_score(self):
        return self._L.ref_min_score(self._h)

    def select_best_H(self, genome_size):
        self.H = self._L.ref_select_best_H(self._h, genome_size)
        return self.H

    def rev64(self, x):
        return self._L.ref_rev64(self._h, x)

    def unrev64(self, x):
        return self._L.ref_unrev64(self._h, x)

    def fingerprint(self, x):
        return self._L.ref_fingerprint(self._h, x)

    def hash_family(self, x, step):
        return self._L.ref_hash_family(self._h, x, step)

    def compute_sketch(self, seq):
        s = _seq(seq)
        out = np.empty(self.F, dtype=np.int32)
        self._L.ref_compute_sketch(self._h, _ptr(s), s.size, _ptr(out))
        return out

    def densify(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.int32).copy()
        self._L.ref_densify(self._h, _ptr(sk), int((sk == -1).sum()))
        return sk

    def insert(self, sk, name="g"):
        sk = np.ascontiguousarray(sk, dtype=np.int32)
        self._L.ref_insert_sketch(self._h, _ptr(sk), name.encode())
        self.n += 1

    def query(self, sk):
        sk = np.ascontiguousarray(sk, dtype=np.int32)
        cap = max(self.n, 1)
        hc = np.empty(cap, dtype=np.uint32)
        hg = np.empty(cap, dtype=np.uint32)
        nh = self._L.ref_query_sketch(self._h, _ptr(sk), _ptr(hc), _ptr(hg), cap)
        return hc[:nh].copy(), hg[:nh].copy()

    def dump(self, path):
        self._L.ref_dump(self._h, path.encode())

    # the reference's threaded driver loops on records in memory (oracle/ref_harness.cpp)
    def sketch_batch(self, seqs, rec_off, threads=0):
        s = _seq(seqs)
        ro = np.ascontiguousarray(rec_off, dtype=np.uint64)
        out = np.empty((ro.size - 1, self.F), dtype=np.int32)
        self._L.ref_sketch_batch(self._h, _ptr(s), _ptr(ro), ro.size - 1, _ptr(out), threads)
        return out

    def insert_batch(self, sketches, threads=0):
        sk = np.ascontiguousarray(sketches, dtype=np.int32)
        self._L.ref_insert_batch(self._h, _ptr(sk), sk.shape[0], threads)
        self.n += sk.shape[0]

    def query_batch(self, sketches, threads=0, cap=None):
        sk = np.ascontiguousarray(sketches, dtype=np.int32)
        n = sk.shape[0]
        cap = cap if cap is not None else max(1, n * 4096)
        off = np.zeros(n + 1, dtype=np.uint64)
        hc, hg = np.empty(cap, dtype=np.uint32), np.empty(cap, dtype=np.uint32)
        tot = self._L.ref_query_batch(self._h, _ptr(sk), n, _ptr(off), _ptr(hc), _ptr(hg), cap, threads)
        return off, hc[:min(tot, cap)], hg[:min(tot, cap)]
