"""All ranks of a slot-sharded group in ONE process over the RCCL transport, with tests/fake_rccl/librccl.so.1
standing in for librccl (tests/test_gpu_fake_rccl.py starts this with that directory first in LD_LIBRARY_PATH and
NIQKI_GROUP_TRANSPORT=rccl).  No torch here: torch brings its own librccl.so.1 into the process, and the loader
would hand THAT to the product's dlopen("librccl.so.1").  Device buffers come from the HIP runtime through ctypes.

    python tests/fake_rccl_worker.py <world> <exchange: sparse|dense|overflow|fail> <S>

Checks, at world ranks on the one device: transport == rccl, the communicator has seen `world` ranks, hit lists of
the group == a whole-range handle == the oracle; the stand-in served the calls; no group call is left open --
also after an injected ncclSend failure, after which the group still works."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


class Hip:
    def __init__(self):
        self.L = C.CDLL("libamdhip64.so")
        self.L.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.L.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.L.hipFree.argtypes = [C.c_void_p]
        self.bufs = []

    def to_dev(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.L.hipMalloc(C.byref(p), max(a.nbytes, 4)) == 0
        assert self.L.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0      # hipMemcpyHostToDevice
        self.bufs.append(p)
        return p.value

    def free_all(self):
        for p in self.bufs:
            self.L.hipFree(p)
        self.bufs = []


def main():
    world, exchange, S = int(sys.argv[1]), sys.argv[2], int(sys.argv[3])
    fake_path = os.path.join(ROOT, "tests", "fake_rccl", "librccl.so.1")
    assert os.environ.get("NIQKI_GROUP_TRANSPORT") == "rccl"
    assert os.path.dirname(fake_path) in os.environ.get("LD_LIBRARY_PATH", "").split(":")[0]
    fake = C.CDLL(fake_path)
    fake.fake_rccl_stat.restype = C.c_uint64
    fake.fake_rccl_stat.argtypes = [C.c_int]
    fake.fake_rccl_fail_send.argtypes = [C.c_int64]
    import niqki_amd as native
    from oracle import pyoracle as po
    from test_gpu_group import make_data
    hip = Hip()
    wide = S == 16
    W, N, NQ, MS = (8, 1234, 21, 40) if not wide else (8, 150, 9, 2000)
    sk, q = make_data(S, W, N, NQ, 5 + world)
    if wide:
        q[0] = sk[3]            # counts of 2^16 (genomes 3 and 5 are equal): the u32 sums of the S = 16 exchange
        valid = (q[0] >= 0) & (q[0] < (1 << W))
        q[0][~valid] = 7
        sk[3] = q[0]
        sk[5] = q[0]
    F = 1 << S
    whole = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    whole.insert(sk)
    w_off, w_hc, w_hg = whole.query(q)
    engines = []
    for r in range(world):
        b, e = native.group_slot_range(r, world, S)
        engines.append(native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS, slot_begin=b, slot_end=e))
    grp = native.Group(engines)
    assert grp.stat("transport") == 1 and grp.stat("rccl") == 1, "the RCCL branch must be the one that runs"
    assert grp.stat("ranks_seen") == world
    assert fake.fake_rccl_stat(0) == world and fake.fake_rccl_stat(8) == world, "the stand-in made the communicators"
    grp.set_option("exchange", 2 if exchange == "dense" else 1)
    if exchange == "overflow":
        grp.set_option("cand_cap", 2)
    # insert through the group: batches of world * ins_per rows, the last one ragged
    ins_per = 37
    for a in range(0, N, world * ins_per):
        blk = sk[a:a + world * ins_per]
        pad = np.full((world * ins_per, F), -1, np.int32)
        pad[:blk.shape[0]] = blk
        grp.insert_dev([hip.to_dev(pad[r * ins_per:(r + 1) * ins_per]) for r in range(world)], ins_per, blk.shape[0])
    assert all(e.n_genomes == N for e in engines)
    per = -(-NQ // world)
    pad = np.full((world * per, F), -1, np.int32)
    pad[:NQ] = q
    loc = [hip.to_dev(pad[r * per:(r + 1) * per]) for r in range(world)]

    def query():
        res = grp.query(loc, per, capacity=8)          # host results; forces the capacity retry
        out = []
        for r in range(world):
            off, hc, hg = res[r]
            for i in range(per):
                out.append((hc[int(off[i]):int(off[i + 1])], hg[int(off[i]):int(off[i + 1])]))
        return out[:NQ]

    if exchange == "fail":
        # the 4th ncclSend of the next all-to-all fails: the call reports it, the group call is closed all the same
        g0 = fake.fake_rccl_stat(5)
        fake.fake_rccl_fail_send(3)
        try:
            query()
            raise AssertionError("the injected ncclSend failure was not reported")
        except native.NiqkiError as err:
            assert err.code == 3 and "Send" in str(err), str(err)
        assert fake.fake_rccl_stat(7) == 0, "a failed send left the group call open"
        assert fake.fake_rccl_stat(5) == g0 + 1
    got = query()
    assert fake.fake_rccl_stat(7) == 0
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    n_hits = 0
    for i in range(NQ):
        lo, hi = int(w_off[i]), int(w_off[i + 1])
        assert np.array_equal(got[i][0], w_hc[lo:hi]) and np.array_equal(got[i][1], w_hg[lo:hi]), (world, exchange, i)
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(got[i][0], ehc) and np.array_equal(got[i][1], ehg), i
        n_hits += hi - lo
    assert n_hits > (50 if not wide else 2)
    if wide:
        assert int(got[0][0][0]) == 1 << 16
    assert (grp.stat("overflows") >= 1) == (exchange == "overflow")
    sends, recvs, ag, rs = (fake.fake_rccl_stat(k) for k in (1, 2, 3, 4))
    assert sends >= world * world and recvs >= world * world and rs >= world, (sends, recvs, ag, rs)
    if exchange in ("sparse", "overflow", "fail"):
        assert ag >= world
    grp.close()
    assert fake.fake_rccl_stat(8) == 0, "niqki_group_destroy must destroy its communicators"
    for e in engines + [whole]:
        e.close()
    hip.free_all()
    print("fake-rccl ok: world %d %s S=%d hits %d sends %d recvs %d allgathers %d reduce-scatters %d bytes %d"
          % (world, exchange, S, n_hits, sends, recvs, ag, rs, fake.fake_rccl_stat(6)))


if __name__ == "__main__":
    main()
