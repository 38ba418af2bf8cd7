"""ctypes binding of include/niqki_hip.h (libniqki_hip.so).

Fails loudly when the shared library is missing or cannot be loaded: there is
no CPU fallback for any entry point.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "lib", "libniqki_hip.so")

MEM_HOST, MEM_DEVICE = 0, 1
SEQ_PAD = 64
KC_SKETCH, KC_DENSIFY, KC_GATHER, KC_HITS, KC_BUILD, KC_INGEST, KC_EXCHANGE, KC_INFLATE = 0, 1, 2, 3, 4, 5, 6, 7
E_GZIP = 7
FILE_GZIP = 0x80
GROUP_ID_BYTES = 128
E_CAPACITY = 4


class NiqkiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("niqki status %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("K", C.c_uint32), ("S", C.c_uint32), ("W", C.c_uint32), ("H", C.c_uint32),
                ("min_score", C.c_uint32), ("slot_begin", C.c_uint32), ("slot_end", C.c_uint32),
                ("device", C.c_int32), ("tile_genomes", C.c_uint32), ("resident_mib", C.c_uint32),
                ("reserved", C.c_uint32 * 2)]


class RawBatch(C.Structure):
    _fields_ = [("raw", C.c_void_p), ("file_ptr", C.c_void_p), ("file_off", C.c_void_p), ("file_type", C.c_void_p),
                ("n_files", C.c_uint32), ("lines", C.c_uint32), ("final", C.c_uint32),
                ("max_entries", C.c_uint32), ("file_status", C.c_void_p)]


class GroupPlan(C.Structure):
    _fields_ = [("sparse", C.c_uint32), ("cand_threshold", C.c_uint32), ("surv_threshold", C.c_uint32),
                ("slice_slots", C.c_uint32), ("slice_bytes", C.c_uint64), ("cand_blob_bytes", C.c_uint64),
                ("sum_words", C.c_uint64), ("row_stride", C.c_uint64)]


class StageInfo(C.Structure):
    _fields_ = [("n_entry", C.c_uint32), ("n_rec", C.c_uint32), ("consumed", C.c_uint64),
                ("seq_bytes", C.c_uint64)]


def lib_path():
    return _LIB


def build_native(force=False):
    """Compile libniqki_hip.so for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    src = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", src, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", src, "-j4"], stdout=subprocess.DEVNULL)
    return _LIB


# every symbol include/niqki_hip.h declares: (name, restype, argtypes)
_vp, _u32, _u64, _i32, _i64, _int, _dbl = (C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32,
                                         C.c_int64, C.c_int, C.c_double)
ABI = [
    ("niqki_abi_version", _int, []),
    ("niqki_status_string", C.c_char_p, [_int]),
    ("niqki_min_score", _u32, [_dbl, _u32]),
    ("niqki_create", _int, [C.POINTER(Params), C.POINTER(_vp)]),
    ("niqki_destroy", None, [_vp]),
    ("niqki_last_error", C.c_char_p, [_vp]),
    ("niqki_get_params", _int, [_vp, C.POINTER(Params)]),
    ("niqki_select_best_H", _int, [_vp, _dbl, C.POINTER(C.c_uint32)]),
    ("niqki_set_stream", _int, [_vp, _vp]),
    ("niqki_get_stream", _vp, [_vp]),
    ("niqki_synchronize", _int, [_vp]),
    ("niqki_set_option", _int, [_vp, C.c_char_p, _i64]),
    ("niqki_reserve", _int, [_vp, _u32]),
    ("niqki_sketch", _int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _int]),
    ("niqki_densify", _int, [_vp, _vp, _u32, _int]),
    ("niqki_insert", _int, [_vp, _vp, _u32, _int]),
    ("niqki_genome_count", _u32, [_vp]),
    ("niqki_build", _int, [_vp]),
    ("niqki_query_counts", _int, [_vp, _vp, _u32, _vp, _u64, _int]),
    ("niqki_query_counts32", _int, [_vp, _vp, _u32, _vp, _u64, _int]),
    ("niqki_hits_from_counts", _int, [_vp, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_candidates_from_counts", _int, [_vp, _vp, _u32, _u64, _u32, _u32, _u32, _vp, _vp, _int]),
    ("niqki_query_counts_candidates", _int, [_vp, _vp, _u32, _vp, _u64, _u32, _u32, _vp, _vp, _int]),
    ("niqki_query", _int, [_vp, _vp, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_sketch_shared", _int, [_vp, _vp, _u64, _vp]),
    ("niqki_insert_shared", _int, [_vp, _vp, C.POINTER(_u32)]),
    ("niqki_query_shared", _int, [_vp, _vp, C.POINTER(_u64), _vp, _vp, _u64]),
    ("niqki_query_sequence_shared", _int, [_vp, _vp, _u64, C.POINTER(_u64), _vp, _vp, _u64]),
    ("niqki_shared_stats", _int, [_vp, C.POINTER(_u64), C.POINTER(_u64), C.POINTER(_u64)]),
    ("niqki_query_survivors", _int, [_vp, _vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _int]),
    ("niqki_survivor_counts", _int, [_vp, _vp, _u32, _vp, _u32, _vp, _vp, _u32, _vp, _int]),
    ("niqki_hits_from_candidates", _int, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_query_sequences", _int, [_vp, _vp, _vp, _u32, _vp, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_sketch_ahead", _int, [_vp, _vp, _vp, _u32, _vp, _u32, _int]),
    ("niqki_query_ahead", _int, [_vp, C.POINTER(_u32), _vp, _vp, _vp, _u64, _vp, _int]),
    ("niqki_stage_raw", _int, [_vp, C.POINTER(RawBatch), _int, C.POINTER(StageInfo), _vp]),
    ("niqki_stage_raw_prefetch", _int, [_vp, C.POINTER(RawBatch)]),
    ("niqki_staged_sketch", _int, [_vp, _vp, _int]),
    ("niqki_staged_insert", _int, [_vp]),
    ("niqki_staged_query", _int, [_vp, _vp, _vp, _vp, _u64, _int]),
    ("niqki_staged_records", _int, [_vp, _vp, _vp, _vp, _vp]),
    ("niqki_pack_bound", C.c_size_t, [C.c_size_t]),
    ("niqki_pack_fasta", C.c_size_t, [_vp, C.c_size_t, _vp, C.c_size_t]),
    ("niqki_unpack_fasta", _int, [_vp, C.c_size_t, _vp, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("niqki_host_alloc", _vp, [C.c_size_t]),
    ("niqki_host_free", None, [_vp]),
    ("niqki_matrix_range", _int, [_vp, _u32, _u32, _vp, _u64, _int]),
    ("niqki_export_dump", _int, [_vp, _vp, _u64, C.POINTER(_u64)]),
    ("niqki_import_dump", _int, [C.POINTER(Params), _vp, _u64, C.POINTER(_u64), C.POINTER(_vp)]),
    ("niqki_export_dump_header", _int, [_vp, _vp]),
    ("niqki_export_dump_layout", _int, [_vp, _vp]),
    ("niqki_export_dump_slots", _int, [_vp, _u32, _u32, _vp, _u64, C.POINTER(_u64)]),
    ("niqki_import_begin", _int, [C.POINTER(Params), _vp, C.POINTER(_vp)]),
    ("niqki_import_slots", _int, [_vp, _u32, _u32, _vp, _u64, C.POINTER(_u64)]),
    ("niqki_get_sketches", _int, [_vp, _u32, _u32, _vp, _int]),
    ("niqki_query_gathered", _int, [_vp, _vp, _u32, _vp, _int]),
    ("niqki_group_slot_range", None, [_u32, _u32, _u32, C.POINTER(_u32), C.POINTER(_u32)]),
    ("niqki_group_new_id", _int, [_vp]),
    ("niqki_group_plan_batch", _int, [_u32, _u32, _u32, _int, _u32, _u32, _u32, C.POINTER(GroupPlan)]),
    ("niqki_group_create", _int, [_vp, _u32, _u32, _u32, _vp, C.POINTER(_vp)]),
    ("niqki_group_destroy", None, [_vp]),
    ("niqki_group_last_error", C.c_char_p, [_vp]),
    ("niqki_group_set_option", _int, [_vp, C.c_char_p, _i64]),
    ("niqki_group_get_stat", _int, [_vp, C.c_char_p, C.POINTER(_u64)]),
    ("niqki_group_insert", _int, [_vp, _vp, _u32, _u32]),
    ("niqki_group_query", _int, [_vp, _vp, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_group_query_begin", _int, [_vp, _vp, _u32, _vp, _vp, _vp, _u64, _int]),
    ("niqki_group_query_end", _int, [_vp]),
    ("niqki_group_staged_insert", _int, [_vp, _u32, _vp]),
    ("niqki_group_staged_query", _int, [_vp, _u32, _vp, _vp, _vp, _vp, _u64, _int]),
    ("niqki_get_stat", _int, [_vp, C.c_char_p, C.POINTER(_u64)]),
    ("niqki_profile_enable", _int, [_vp, _int]),
    ("niqki_profile_reset", _int, [_vp]),
    ("niqki_profile_read", _int, [_vp, _int, C.POINTER(_dbl), C.POINTER(_u64)]),
    ("niqki_synth_genomes", _int, [_vp, _u64, _vp, _vp, _vp, _u32, _u64, _u64, _vp, _int]),
    ("niqki_synth_genome_host", None, [_u64, _u32, _u32, _u32, _u64, _vp]),
    ("niqki_synth_reads", _int, [_vp, _u64, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u64, _vp, _int]),
    ("niqki_measure_alu", _int, [_vp, _int, _dbl, C.POINTER(_dbl)]),
    ("niqki_gunzip", _int, [_vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_u64)]),
    ("niqki_gunzip_stats", _int, [_vp, _vp]),
]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            raise ImportError("%s is missing: run __graft_entry__.build() (make -C niqki_amd/csrc); "
                              "there is no fallback path" % _LIB)
        # libniqki_hip.so binds to the HIP runtime by soname (libamdhip64.so.7).
        # PyTorch wheels ship their own copy under the same soname; when both are
        # used in one process PyTorch's must be loaded first, otherwise the process
        # ends up with two runtimes and the second one to initialise finds no
        # device.  (The C++ host program never loads PyTorch.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB)
        for name, res, args in ABI:
            f = getattr(L, name)  # AttributeError if the library lacks a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def row_stride(n):
    """NIQKI_ROW_STRIDE of the header: counter rows that start on 128-byte lines."""
    return (int(n) + 63) & ~63


def min_score(J, S):
    return lib().niqki_min_score(J, S)


def pack_fasta(data):
    """niqki_pack_fasta: the packed container of a FASTA file's bytes (np.uint8 array), or None when the file is not
    worth packing (hand it over raw).  Host code: no GPU needed."""
    raw = np.frombuffer(bytes(data), dtype=np.uint8)
    out = np.empty(lib().niqki_pack_bound(raw.size), dtype=np.uint8)
    n = lib().niqki_pack_fasta(raw.ctypes.data if raw.size else None, raw.size, out.ctypes.data, out.size)
    return out[:n].copy() if n else None


def unpack_fasta(container):
    """niqki_unpack_fasta: the file's bytes back (the host restatement of the device pass)."""
    c = np.ascontiguousarray(container, dtype=np.uint8)
    n = C.c_size_t(0)
    rc = lib().niqki_unpack_fasta(c.ctypes.data, c.size, None, 0, C.byref(n))
    if rc:
        raise NiqkiError(rc, "niqki_unpack_fasta: not a well-formed container")
    out = np.empty(max(n.value, 1), dtype=np.uint8)
    rc = lib().niqki_unpack_fasta(c.ctypes.data, c.size, out.ctypes.data, out.size, C.byref(n))
    if rc:
        raise NiqkiError(rc, "niqki_unpack_fasta failed")
    return out[:n.value]


def synth_genome_host(seed, family, member, rate14, length):
    out = np.empty(length, dtype=np.uint8)
    lib().niqki_synth_genome_host(seed, family, member, rate14, length, out.ctypes.data)
    return out


def _p(x):
    """numpy array / torch tensor / int / None -> raw address."""
    if x is None:
        return None
    if isinstance(x, np.ndarray):
        return x.ctypes.data
    if isinstance(x, int):
        return x
    return x.data_ptr()  # torch tensor


class Engine:
    """One handle = one GPU (or one slot shard of an index)."""

    def __init__(self, K=31, S=15, W=12, H=4, J=0.0, min_score_value=None, device=-1,
                 slot_begin=0, slot_end=0, tile_genomes=0, resident_mib=0, _handle=None):
        self.L = lib()
        if _handle is not None:
            self.h = _handle
        else:
            ms = self.L.niqki_min_score(J, S) if min_score_value is None else min_score_value
            p = Params(K, S, W, H, ms, slot_begin, slot_end, device, tile_genomes, resident_mib)
            h = _vp()
            rc = self.L.niqki_create(C.byref(p), C.byref(h))
            if rc:
                raise NiqkiError(rc, "%s (%s)" % (self.L.niqki_status_string(rc).decode(),
                                                  self.L.niqki_last_error(None).decode()))
            self.h = h
        q = Params()
        self.L.niqki_get_params(self.h, C.byref(q))
        self.K, self.S, self.W, self.H, self.min_score = q.K, q.S, q.W, q.H, q.min_score
        self.slot_begin, self.slot_end, self.device = q.slot_begin, q.slot_end, q.device
        self.F = 1 << self.S

    def close(self):
        if getattr(self, "h", None):
            self.L.niqki_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, allow=()):
        if rc and rc not in allow:
            raise NiqkiError(rc, "%s (%s)" % (self.L.niqki_status_string(rc).decode(),
                                              self.L.niqki_last_error(self.h).decode()))
        return rc

    # -- plumbing
    def set_stream(self, stream_ptr):
        self._ck(self.L.niqki_set_stream(self.h, stream_ptr))

    # -- many host threads on one handle (niqki_*_shared: callers are combined into batches)
    def sketch_shared(self, seq):
        s = np.ascontiguousarray(seq, dtype=np.uint8)
        out = np.empty(self.F, dtype=np.int32)
        self._ck(self.L.niqki_sketch_shared(self.h, _p(s), s.size, _p(out)))
        return out

    def insert_shared(self, sketch):
        sk = np.ascontiguousarray(sketch, dtype=np.int32)
        gid = _u32(0)
        self._ck(self.L.niqki_insert_shared(self.h, _p(sk), C.byref(gid)))
        return gid.value

    def query_shared(self, sketch, capacity=1024):
        sk = np.ascontiguousarray(sketch, dtype=np.int32)
        n = _u64(0)
        hc, hg = np.empty(capacity, np.uint32), np.empty(capacity, np.uint32)
        self._ck(self.L.niqki_query_shared(self.h, _p(sk), C.byref(n), _p(hc), _p(hg), capacity))
        if n.value > capacity:
            return self.query_shared(sketch, int(n.value))
        return hc[:n.value].copy(), hg[:n.value].copy()

    def query_sequence_shared(self, seq, capacity=1024):
        s = np.ascontiguousarray(seq, dtype=np.uint8)
        n = _u64(0)
        hc, hg = np.empty(capacity, np.uint32), np.empty(capacity, np.uint32)
        self._ck(self.L.niqki_query_sequence_shared(self.h, _p(s), s.size, C.byref(n), _p(hc), _p(hg), capacity))
        if n.value > capacity:
            return self.query_sequence_shared(seq, int(n.value))
        return hc[:n.value].copy(), hg[:n.value].copy()

    def shared_stats(self):
        b, r, m = _u64(0), _u64(0), _u64(0)
        self._ck(self.L.niqki_shared_stats(self.h, C.byref(b), C.byref(r), C.byref(m)))
        return {"batches": b.value, "requests": r.value, "largest_batch": m.value}

    def get_stream(self):
        """The hipStream_t the handle enqueues on (an int, for torch.cuda.ExternalStream)."""
        return int(self.L.niqki_get_stream(self.h) or 0)

    def synchronize(self):
        self._ck(self.L.niqki_synchronize(self.h))

    def set_option(self, key, value):
        self._ck(self.L.niqki_set_option(self.h, key.encode(), int(value)))

    def reserve(self, n):
        self._ck(self.L.niqki_reserve(self.h, n))

    def select_best_H(self, genome_size):
        """Index::select_best_H (-G): src/niqki_index.cpp:126-138."""
        H = C.c_uint32(0)
        self._ck(self.L.niqki_select_best_H(self.h, float(genome_size), C.byref(H)))
        self.H = H.value
        return self.H

    @property
    def n_genomes(self):
        return self.L.niqki_genome_count(self.h)

    def stat(self, key):
        v = _u64(0)
        self._ck(self.L.niqki_get_stat(self.h, key.encode(), C.byref(v)))
        return v.value

    def tile_genomes(self):
        q = Params()
        self.L.niqki_get_params(self.h, C.byref(q))
        return q.tile_genomes

    # -- host-memory convenience (numpy in / numpy out)
    @staticmethod
    def pack_records(records):
        """list of bytes/str/uint8 arrays -> (seqs uint8, rec_off uint64)."""
        arrs = []
        for r in records:
            if isinstance(r, str):
                r = r.encode()
            if isinstance(r, (bytes, bytearray)):
                r = np.frombuffer(bytes(r), dtype=np.uint8)
            arrs.append(np.ascontiguousarray(r, dtype=np.uint8))
        off = np.zeros(len(arrs) + 1, dtype=np.uint64)
        if arrs:
            off[1:] = np.cumsum([a.size for a in arrs], dtype=np.uint64)
        seqs = np.concatenate(arrs) if arrs else np.zeros(0, dtype=np.uint8)
        return np.ascontiguousarray(seqs), off

    def sketch(self, records, entry_rec=None):
        seqs, off = self.pack_records(records)
        n_rec = off.size - 1
        if entry_rec is None:
            n_entry, er = n_rec, None
        else:
            er = np.ascontiguousarray(entry_rec, dtype=np.uint32)
            n_entry = er.size - 1
        out = np.empty((n_entry, self.F), dtype=np.int32)
        self._ck(self.L.niqki_sketch(self.h, _p(seqs) if seqs.size else _p(np.zeros(1, np.uint8)),
                                     _p(off), n_rec, _p(er), n_entry, _p(out), MEM_HOST))
        return out

    def densify(self, sketches):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).copy()
        sk2 = sk.reshape(-1, self.F)
        self._ck(self.L.niqki_densify(self.h, _p(sk2), sk2.shape[0], MEM_HOST))
        return sk

    def insert(self, sketches):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).reshape(-1, self.F)
        self._ck(self.L.niqki_insert(self.h, _p(sk), sk.shape[0], MEM_HOST))

    def build(self):
        self._ck(self.L.niqki_build(self.h))

    def query_counts(self, sketches):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).reshape(-1, self.F)
        n = self.n_genomes
        stride = row_stride(n)
        out = np.zeros((sk.shape[0], max(stride, 2)), dtype=np.uint16)
        self._ck(self.L.niqki_query_counts(self.h, _p(sk), sk.shape[0], _p(out), out.shape[1], MEM_HOST))
        return out[:, :n]

    def query_counts32(self, sketches):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).reshape(-1, self.F)
        n = self.n_genomes
        stride = row_stride(n)
        out = np.zeros((sk.shape[0], max(stride, 2)), dtype=np.uint32)
        self._ck(self.L.niqki_query_counts32(self.h, _p(sk), sk.shape[0], _p(out), out.shape[1], MEM_HOST))
        return out[:, :n]

    def _hits(self, call, nq, capacity):
        off = np.zeros(nq + 1, dtype=np.uint64)
        while True:
            hc = np.empty(max(capacity, 1), dtype=np.uint32)
            hg = np.empty(max(capacity, 1), dtype=np.uint32)
            rc = call(off, hc, hg, capacity)
            self._ck(rc, allow=(E_CAPACITY,))
            if rc == 0:
                tot = int(off[nq])
                return off, hc[:tot], hg[:tot]
            capacity = int(off[nq])

    def query(self, sketches, capacity=None):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).reshape(-1, self.F)
        nq = sk.shape[0]
        cap = capacity if capacity is not None else max(1024, nq * 64)
        return self._hits(lambda off, hc, hg, c: self.L.niqki_query(
            self.h, _p(sk), nq, _p(off), _p(hc), _p(hg), c, MEM_HOST), nq, cap)

    def hits_from_counts(self, counts, gid_begin=0, n_gids=None, capacity=None):
        ct = np.ascontiguousarray(counts, dtype=np.uint16)
        nq, stride = ct.shape
        ng = stride - gid_begin if n_gids is None else n_gids
        cap = capacity if capacity is not None else max(1024, nq * 64)
        return self._hits(lambda off, hc, hg, c: self.L.niqki_hits_from_counts(
            self.h, _p(ct), nq, stride, gid_begin, ng, _p(off), _p(hc), _p(hg), c, MEM_HOST), nq, cap)

    def query_sequences(self, records, capacity=None):
        seqs, off = self.pack_records(records)
        nq = off.size - 1
        cap = capacity if capacity is not None else max(1024, nq * 64)
        return self._hits(lambda ho, hc, hg, c: self.L.niqki_query_sequences(
            self.h, _p(seqs), _p(off), nq, None, nq, _p(ho), _p(hc), _p(hg), c, MEM_HOST), nq, cap)

    # -- raw file bytes, framed on the GPU (niqki_stage_raw and friends)
    def stage_raw(self, files, types=None, lines=False, final=True, max_entries=16384, scattered=False, prefetch=None):
        """files: list of bytes-like (the gunzipped content of each file).  Returns
        (StageInfo, entry_hdr) -- entry_hdr: raw offset of each entry's header line (lines mode).
        scattered: hand the files over as separate buffers (file_ptr) instead of one.
        prefetch (with scattered): "this" = niqki_stage_raw_prefetch of this very batch first,
        "only" = just the prefetch, nothing staged (the buffers stay alive on the object), "take" =
        stage the batch of the last "only" call from those very buffers."""
        blobs = [np.frombuffer(bytes(f), dtype=np.uint8) for f in files]
        off = np.zeros(len(blobs) + 1, dtype=np.uint64)
        off[1:] = np.cumsum([b.size for b in blobs], dtype=np.uint64)
        raw = np.concatenate(blobs + [np.zeros(0, np.uint8)]) if blobs else np.zeros(0, np.uint8)
        ty = np.array([ord(t) if isinstance(t, str) else int(t) for t in (types if types else "A" * len(blobs))], dtype=np.uint8)
        if ty.size == 0:
            ty = np.zeros(1, np.uint8)
        self.file_status = np.zeros(max(len(blobs), 1), dtype=np.uint8)   # (why a gzip file was refused, after E_GZIP)
        ptrs = None
        if prefetch == "take":      # the buffers an earlier prefetch="only" call handed over
            keep, ptrs, off, ty = self._pre_keep
        elif scattered:
            keep = [np.ascontiguousarray(x).copy() for x in blobs]
            ptrs = (C.c_void_p * max(len(keep), 1))(*[k.ctypes.data for k in keep])
        b = RawBatch(None if scattered or not raw.size else _p(raw), C.cast(ptrs, C.c_void_p) if scattered else None,
                     _p(off), _p(ty), len(blobs), int(bool(lines)), int(bool(final)), max_entries, _p(self.file_status))
        info = StageInfo()
        hdr = np.zeros(max(max_entries, 1), dtype=np.uint64)
        if prefetch in ("this", "only"):
            self._ck(self.L.niqki_stage_raw_prefetch(self.h, C.byref(b)))
            if prefetch == "only":
                self._pre_keep = (keep, ptrs, off, ty)
                return None, None
        self._ck(self.L.niqki_stage_raw(self.h, C.byref(b), MEM_HOST, C.byref(info), _p(hdr) if lines else None))
        self._staged = info
        return info, hdr[:info.n_entry] if lines else None

    def gunzip(self, files, sizes, check_outside=True):
        """The device inflate alone (niqki_gunzip): files = gzip files as bytes, sizes = the length each is expected
        to inflate to.  Returns (list of bytes, status, produced, members, outside)."""
        blobs = [np.frombuffer(bytes(f), dtype=np.uint8) for f in files]
        goff = np.zeros(len(blobs) + 1, dtype=np.uint64)
        goff[1:] = np.cumsum([b.size for b in blobs], dtype=np.uint64)
        roff = np.zeros(len(blobs) + 1, dtype=np.uint64)
        roff[1:] = np.cumsum(list(sizes), dtype=np.uint64)
        gz = np.concatenate(blobs + [np.zeros(8, np.uint8)])
        raw = np.zeros(int(roff[-1]) + 8, dtype=np.uint8)
        n = len(blobs)
        status, produced, members = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.uint64), np.zeros(max(n, 1), np.uint32)
        outside = C.c_uint64(0)
        self._ck(self.L.niqki_gunzip(self.h, _p(gz), _p(goff), n, _p(roff), _p(raw), _p(status), _p(produced), _p(members),
                                     C.byref(outside) if check_outside else None))
        out = [raw[int(roff[i]):int(roff[i]) + int(min(produced[i], roff[i + 1] - roff[i]))].tobytes() for i in range(n)]
        return out, status[:n], produced[:n], members[:n], int(outside.value)

    def gunzip_stats(self):
        """{rounds, round_bytes, serial_tokens, blocks} of the last device inflate (summed over its files)"""
        o = np.zeros(4, dtype=np.uint64)
        self._ck(self.L.niqki_gunzip_stats(self.h, _p(o)))
        return dict(zip(("rounds", "round_bytes", "serial_tokens", "blocks"), (int(x) for x in o)))

    def staged_records(self):
        """(records as list of bytes, entry_rec, hdr_pos) of the staged batch."""
        st = self._staged
        rec_off = np.zeros(st.n_rec + 1, dtype=np.uint64)
        seqs = np.zeros(max(st.seq_bytes, 1), dtype=np.uint8)
        entry_rec = np.zeros(st.n_entry + 1, dtype=np.uint32)
        hdr_pos = np.zeros(max(st.n_rec, 1), dtype=np.uint64)
        self._ck(self.L.niqki_staged_records(self.h, _p(rec_off), _p(seqs), _p(entry_rec), _p(hdr_pos)))
        recs = [seqs[int(rec_off[i]):int(rec_off[i + 1])].tobytes() for i in range(st.n_rec)]
        return recs, entry_rec, hdr_pos[:st.n_rec]

    def staged_sketch(self):
        out = np.empty((self._staged.n_entry, self.F), dtype=np.int32)
        self._ck(self.L.niqki_staged_sketch(self.h, _p(out), MEM_HOST))
        return out

    def staged_insert(self):
        self._ck(self.L.niqki_staged_insert(self.h))

    def staged_query(self, capacity=None):
        nq = self._staged.n_entry
        cap = capacity if capacity is not None else max(1024, nq * 64)
        return self._hits(lambda ho, hc, hg, c: self.L.niqki_staged_query(
            self.h, _p(ho), _p(hc), _p(hg), c, MEM_HOST), nq, cap)

    def matrix_range(self, begin, end):
        n = self.n_genomes
        stride = max(row_stride(n), 2)
        out = np.zeros((end - begin, stride), dtype=np.uint16)
        self._ck(self.L.niqki_matrix_range(self.h, begin, end, _p(out), stride, MEM_HOST))
        return out[:, :n]

    def get_sketches(self, begin, n):
        out = np.empty((n, self.F), dtype=np.int32)
        self._ck(self.L.niqki_get_sketches(self.h, begin, n, _p(out), MEM_HOST))
        return out

    def gathered(self, sketches):
        sk = np.ascontiguousarray(sketches, dtype=np.int32).reshape(-1, self.F)
        out = np.zeros(sk.shape[0], dtype=np.uint64)
        self._ck(self.L.niqki_query_gathered(self.h, _p(sk), sk.shape[0], _p(out), MEM_HOST))
        return out

    def export_dump(self):
        size = _u64(0)
        self._ck(self.L.niqki_export_dump(self.h, None, 0, C.byref(size)))
        buf = np.empty(size.value, dtype=np.uint8)
        self._ck(self.L.niqki_export_dump(self.h, _p(buf), buf.size, C.byref(size)))
        return buf.tobytes()

    @classmethod
    def import_dump(cls, data, device=-1, tile_genomes=0, resident_mib=0):
        L = lib()
        buf = np.frombuffer(data, dtype=np.uint8)
        p = Params(31, 15, 12, 4, 0, 0, 0, device, tile_genomes, resident_mib)
        h = _vp()
        consumed = _u64(0)
        rc = L.niqki_import_dump(C.byref(p), _p(buf), buf.size, C.byref(consumed), C.byref(h))
        if rc:
            raise NiqkiError(rc, L.niqki_status_string(rc).decode())
        e = cls(_handle=h)
        e.names_offset = consumed.value
        return e

    # -- profiling
    def profile(self, on=True):
        self._ck(self.L.niqki_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._ck(self.L.niqki_profile_reset(self.h))

    def profile_read(self, kc):
        ms, n = _dbl(0), _u64(0)
        self._ck(self.L.niqki_profile_read(self.h, kc, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    # -- raw device-pointer calls (torch tensors or addresses)
    def synth_dev(self, seed, family, member, rate14, n, length, stride, out):
        self._ck(self.L.niqki_synth_genomes(self.h, seed, _p(family), _p(member), _p(rate14), n,
                                            length, stride, _p(out), MEM_DEVICE))

    def synth_reads_dev(self, seed, family, member, rate14, offset, read_id, read_rate14, n, length, stride, out):
        self._ck(self.L.niqki_synth_reads(self.h, seed, _p(family), _p(member), _p(rate14), _p(offset), _p(read_id),
                                          read_rate14, n, length, stride, _p(out), MEM_DEVICE))

    def synth_reads_host(self, seed, family, member, rate14, offset, read_id, read_rate14, length):
        a = [np.ascontiguousarray(x, dtype=np.uint32) for x in (family, member, rate14, read_id)]
        off = np.ascontiguousarray(offset, dtype=np.uint64)
        out = np.empty((a[0].size, length), dtype=np.uint8)
        self._ck(self.L.niqki_synth_reads(self.h, seed, _p(a[0]), _p(a[1]), _p(a[2]), _p(off), _p(a[3]), read_rate14,
                                          a[0].size, length, length, _p(out), MEM_HOST))
        return out

    def measure_alu(self, what, ms=20.0):
        r = _dbl(0)
        self._ck(self.L.niqki_measure_alu(self.h, what, ms, C.byref(r)))
        return r.value

    def stage_raw_dev(self, raw, file_off, types, lines=False, final=True, max_entries=16384):
        """raw: device bytes (torch tensor / address); file_off, types: host numpy arrays."""
        off = np.ascontiguousarray(file_off, dtype=np.uint64)
        ty = np.ascontiguousarray(types, dtype=np.uint8)
        b = RawBatch(_p(raw), None, _p(off), _p(ty), off.size - 1, int(bool(lines)), int(bool(final)), max_entries)
        info = StageInfo()
        hdr = np.zeros(max(max_entries, 1), dtype=np.uint64)
        self._ck(self.L.niqki_stage_raw(self.h, C.byref(b), MEM_DEVICE, C.byref(info), _p(hdr) if lines else None))
        self._staged = info
        return info, hdr[:info.n_entry] if lines else None

    def sketch_dev(self, seqs, rec_off, n_rec, sketches, entry_rec=None, n_entry=None):
        self._ck(self.L.niqki_sketch(self.h, _p(seqs), _p(rec_off), n_rec, _p(entry_rec),
                                     n_rec if n_entry is None else n_entry, _p(sketches), MEM_DEVICE))

    def insert_dev(self, sketches, n):
        self._ck(self.L.niqki_insert(self.h, _p(sketches), n, MEM_DEVICE))

    def query_counts_dev(self, sketches, nq, counts, stride):
        self._ck(self.L.niqki_query_counts(self.h, _p(sketches), nq, _p(counts), stride, MEM_DEVICE))

    def hits_from_counts_dev(self, counts, nq, stride, gid_begin, n_gids, hit_off, hc, hg, capacity):
        self._ck(self.L.niqki_hits_from_counts(self.h, _p(counts), nq, stride, gid_begin, n_gids,
                                               _p(hit_off), _p(hc), _p(hg), capacity, MEM_DEVICE))

    def candidates_dev(self, counts, nq, stride, n_gids, threshold, cap, cand, n_cand):
        self._ck(self.L.niqki_candidates_from_counts(self.h, _p(counts), nq, stride, n_gids, threshold, cap,
                                                     _p(cand), _p(n_cand), MEM_DEVICE))

    def query_counts_candidates_dev(self, sketches, nq, counts, stride, threshold, cap, cand, n_cand):
        self._ck(self.L.niqki_query_counts_candidates(self.h, _p(sketches), nq, _p(counts), stride, threshold, cap,
                                                      _p(cand), _p(n_cand), MEM_DEVICE))

    def query_survivors_dev(self, sketches, nq, cand_thr, surv_thr, cand_cap, surv_cap, cand, n_cand, surv, n_surv):
        self._ck(self.L.niqki_query_survivors(self.h, _p(sketches), nq, cand_thr, surv_thr, cand_cap, surv_cap, _p(cand),
                                              _p(n_cand), _p(surv), _p(n_surv), MEM_DEVICE))

    def survivor_counts_dev(self, sketches, nq, ids, m, surv, n_surv, surv_cap, counts):
        self._ck(self.L.niqki_survivor_counts(self.h, _p(sketches), nq, _p(ids), m, _p(surv), _p(n_surv), surv_cap,
                                              _p(counts), MEM_DEVICE))

    def hits_from_candidates_dev(self, ids, totals, nq, m, hit_off, hc, hg, capacity):
        self._ck(self.L.niqki_hits_from_candidates(self.h, _p(ids), _p(totals), nq, m, _p(hit_off), _p(hc), _p(hg),
                                                   capacity, MEM_DEVICE))

    def query_dev(self, sketches, nq, hit_off, hc, hg, capacity):
        self._ck(self.L.niqki_query(self.h, _p(sketches), nq, _p(hit_off), _p(hc), _p(hg), capacity,
                                    MEM_DEVICE))

    def query_sequences_dev(self, seqs, rec_off, n, hit_off, hc, hg, capacity):
        self._ck(self.L.niqki_query_sequences(self.h, _p(seqs), _p(rec_off), n, None, n, _p(hit_off),
                                              _p(hc), _p(hg), capacity, MEM_DEVICE))

    def sketch_ahead_dev(self, seqs, rec_off, n_rec, entry_rec=None, n_entry=None):
        """niqki_sketch_ahead: the batch's sketch kernel on the handle's sketch lane, beside what its stream runs."""
        self._ck(self.L.niqki_sketch_ahead(self.h, _p(seqs), _p(rec_off), n_rec, _p(entry_rec),
                                           n_rec if n_entry is None else n_entry, MEM_DEVICE))

    def query_ahead_dev(self, hit_off, hc, hg, capacity, sketches=None):
        """niqki_query_ahead with device outputs: the oldest batch sketched ahead; returns its entry count."""
        n = _u32(0)
        self._ck(self.L.niqki_query_ahead(self.h, C.byref(n), _p(hit_off), _p(hc), _p(hg), capacity, _p(sketches), MEM_DEVICE))
        return n.value

    def query_ahead(self, n_entry, capacity=None, want_sketches=False):
        """niqki_query_ahead with host outputs: (off, counts, gids[, sketches]) of the oldest batch sketched ahead
        (n_entry: its entry count, as given to sketch_ahead_dev)."""
        n = _u32(0)
        cap = capacity if capacity is not None else max(1024, n_entry * 64)
        while True:
            off = np.zeros(n_entry + 1, dtype=np.uint64)
            hc, hg = np.empty(max(cap, 1), dtype=np.uint32), np.empty(max(cap, 1), dtype=np.uint32)
            sk = np.empty((n_entry, self.F), dtype=np.int32) if want_sketches else None
            rc = self.L.niqki_query_ahead(self.h, C.byref(n), _p(off), _p(hc), _p(hg), cap, _p(sk), MEM_HOST)
            self._ck(rc, allow=(E_CAPACITY,))
            if rc == 0:
                tot = int(off[n.value])
                res = (off[:n.value + 1], hc[:tot], hg[:tot])
                return res + (sk[:n.value],) if want_sketches else res
            cap = int(off[n.value])

    def gathered_dev(self, sketches, nq):
        out = np.zeros(nq, dtype=np.uint64)
        self._ck(self.L.niqki_query_gathered(self.h, _p(sketches), nq, _p(out), MEM_DEVICE))
        return out


def group_slot_range(rank, world, S):
    b, e = _u32(0), _u32(0)
    lib().niqki_group_slot_range(rank, world, S, C.byref(b), C.byref(e))
    return b.value, e.value


def group_plan(world, S, min_score, exchange_option, per, n_genomes, cand_cap):
    """niqki_group_plan_batch: the decisions and sizes of one query batch of a group (pure arithmetic: no GPU)."""
    p = GroupPlan()
    rc = lib().niqki_group_plan_batch(world, S, min_score, exchange_option, per, n_genomes, cand_cap, C.byref(p))
    if rc:
        raise NiqkiError(rc, "niqki_group_plan_batch: invalid arguments")
    return p


def group_new_id():
    """128 opaque bytes naming a group about to be formed (rank 0 makes them, the caller carries
    them to the other processes)."""
    buf = np.zeros(GROUP_ID_BYTES, dtype=np.uint8)
    rc = lib().niqki_group_new_id(buf.ctypes.data)
    if rc:
        raise NiqkiError(rc, "niqki_group_new_id: " + lib().niqki_status_string(rc).decode())
    return buf


class Group:
    """The ranks of a slot-sharded index that live in this process (niqki_group_*):
    engines[i] is rank first_rank + i, created with group_slot_range(rank, world, S)."""

    def __init__(self, engines, first_rank=0, world=None, group_id=None):
        self.L = lib()
        self.engines = list(engines)
        self.n_local = len(self.engines)
        self.first = first_rank
        self.world = self.n_local if world is None else world
        hs = (_vp * self.n_local)(*[e.h for e in self.engines])
        idp = None if group_id is None else np.ascontiguousarray(group_id, dtype=np.uint8).ctypes.data
        g = _vp()
        rc = self.L.niqki_group_create(C.cast(hs, _vp), self.n_local, first_rank, self.world, idp, C.byref(g))
        if rc:
            raise NiqkiError(rc, "%s (%s)" % (self.L.niqki_status_string(rc).decode(),
                                              self.L.niqki_last_error(self.engines[0].h).decode()))
        self.g = g

    def close(self):
        if getattr(self, "g", None):
            self.L.niqki_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, allow=()):
        if rc and rc not in allow:
            raise NiqkiError(rc, "%s (%s)" % (self.L.niqki_status_string(rc).decode(),
                                              self.L.niqki_group_last_error(self.g).decode()))
        return rc

    def set_option(self, key, value):
        self._ck(self.L.niqki_group_set_option(self.g, key.encode(), int(value)))

    def stat(self, key):
        v = _u64(0)
        self._ck(self.L.niqki_group_get_stat(self.g, key.encode(), C.byref(v)))
        return v.value

    def _ptrs(self, xs):
        return C.cast((_vp * self.n_local)(*[_p(x) for x in xs]), _vp)

    def insert_dev(self, local_sketches, per, n_total):
        """local_sketches: per local rank a device tensor / address of per x F int32."""
        self._ck(self.L.niqki_group_insert(self.g, self._ptrs(local_sketches), per, n_total))

    def query_dev(self, local_sketches, per, hit_off, hit_counts, hit_gids, capacity):
        """Per-rank device outputs (torch tensors / addresses): hit_off[i] int64 [per+1]."""
        self._ck(self.L.niqki_group_query(self.g, self._ptrs(local_sketches), per, self._ptrs(hit_off),
                                          self._ptrs(hit_counts), self._ptrs(hit_gids), capacity, MEM_DEVICE))

    def query_begin_dev(self, local_sketches, per, hit_off, hit_counts, hit_gids, capacity):
        """niqki_group_query_begin with device outputs: returns without waiting for the device."""
        self._ck(self.L.niqki_group_query_begin(self.g, self._ptrs(local_sketches), per, self._ptrs(hit_off),
                                                self._ptrs(hit_counts), self._ptrs(hit_gids), capacity, MEM_DEVICE))

    def query_end(self):
        self._ck(self.L.niqki_group_query_end(self.g))

    def query(self, local_sketches, per, capacity=None):
        """Device sketches in, numpy hits out: list of (off, counts, gids) per local rank."""
        cap = capacity if capacity is not None else max(1024, per * 64)
        while True:
            off = [np.zeros(per + 1, dtype=np.uint64) for _ in range(self.n_local)]
            hc = [np.empty(max(cap, 1), dtype=np.uint32) for _ in range(self.n_local)]
            hg = [np.empty(max(cap, 1), dtype=np.uint32) for _ in range(self.n_local)]
            rc = self.L.niqki_group_query(self.g, self._ptrs(local_sketches), per, self._ptrs(off), self._ptrs(hc),
                                          self._ptrs(hg), cap, MEM_HOST)
            self._ck(rc, allow=(E_CAPACITY,))
            if rc == 0:
                return [(o, c[:int(o[per])], g_[:int(o[per])]) for o, c, g_ in zip(off, hc, hg)]
            cap = max(int(o[per]) for o in off)
