"""GPU: a slot-sharded index behind the C ABI (niqki_group_*, niqki_amd/csrc/nq_group.hip).

World-size G groups are run on ONE GPU: all ranks in this process, the shards sharing the device,
so device-to-device copies stand in for the RCCL collectives while every exchange kernel, the
slot-shard gather, the candidate logic and the per-rank threshold run as they do on G GPUs.  The
answers must be those of one whole-range handle (itself checked against the oracle elsewhere) and
of the oracle.  A world-1 group goes through librccl itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_data(S, W, N, nq, seed):
    rng = np.random.default_rng(seed)
    F = 1 << S
    fam = rng.integers(0, 1 << W, (12, F)).astype(np.int32)
    sk = fam[rng.integers(0, 12, N)].copy()
    noise = rng.random((N, F)) < 0.4
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((N, F)) < 0.02] = -1
    sk[5] = sk[3]
    q = fam[rng.integers(0, 12, nq)].copy()
    m = rng.random((nq, F)) < 0.25
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[1] = -1                                   # no hit at all
    q[2, ::2] = 1 << W                          # cells outside [0, 2^W) are never queried
    return sk, q


def shards_and_group(native, world, S, W, min_score, **kw):
    engines = []
    for r in range(world):
        b, e = native.group_slot_range(r, world, S)
        engines.append(native.Engine(K=31, S=S, W=W, H=3, min_score_value=min_score, slot_begin=b, slot_end=e))
    return engines, native.Group(engines, **kw)


def run_group(native, torch, engines, grp, sk, q, per):
    """Inserts sk through the group in ragged batches, queries q; returns per-query (counts, gids)."""
    dev = torch.device("cuda")
    world, F = len(engines), sk.shape[1]
    for e in engines:
        e.set_stream(torch.cuda.current_stream().cuda_stream)
    # insert: batches of world * ins_per rows, the last one ragged
    ins_per = 37
    for a in range(0, sk.shape[0], world * ins_per):
        blk = sk[a:a + world * ins_per]
        pad = np.full((world * ins_per, F), -1, np.int32)
        pad[:blk.shape[0]] = blk
        loc = [torch.from_numpy(pad[r * ins_per:(r + 1) * ins_per].copy()).to(dev) for r in range(world)]
        grp.insert_dev(loc, ins_per, blk.shape[0])
    assert all(e.n_genomes == sk.shape[0] for e in engines)
    nq = q.shape[0]
    pad = np.full((world * per, F), -1, np.int32)
    pad[:nq] = q
    loc = [torch.from_numpy(pad[r * per:(r + 1) * per].copy()).to(dev) for r in range(world)]
    res = grp.query(loc, per, capacity=8)         # forces the capacity retry
    out = []
    for r in range(world):
        off, hc, hg = res[r]
        for i in range(per):
            out.append((hc[int(off[i]):int(off[i + 1])], hg[int(off[i]):int(off[i + 1])]))
    return out[:nq]


@pytest.mark.parametrize("world,exchange", [(8, "sparse"), (8, "dense"), (8, "overflow"), (3, "sparse"), (2, "dense"), (1, "sparse")])
def test_emulated_slot_shards_equal_the_whole_index(native, po, world, exchange):
    import torch
    S, W, N, NQ, MS = 9, 8, 1234, 21, 40
    sk, q = make_data(S, W, N, NQ, 5 + world)
    whole = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    whole.insert(sk)
    w_off, w_hc, w_hg = whole.query(q)
    engines, grp = shards_and_group(native, world, S, W, MS)
    assert grp.stat("rccl") == (1 if world == 1 else 0)     # shards on one device: local transport
    grp.set_option("exchange", 2 if exchange == "dense" else 1)
    if exchange == "overflow":
        grp.set_option("cand_cap", 2)                        # lists overflow -> the step is redone densely
    per = -(-NQ // world)
    got = run_group(native, torch, engines, grp, sk, q, per)
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    for i in range(NQ):
        lo, hi = int(w_off[i]), int(w_off[i + 1])
        assert np.array_equal(got[i][0], w_hc[lo:hi]) and np.array_equal(got[i][1], w_hg[lo:hi]), (world, exchange, i)
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(got[i][0], ehc) and np.array_equal(got[i][1], ehg), i
    assert sum(len(g[0]) for g in got) > 50
    assert (grp.stat("overflows") >= 1) == (exchange == "overflow")   # every (retried) step of the overflowing case
    # the shards' partial hit vectors add up to the whole index's
    tot = sum(e.query_counts(q).astype(np.uint32) for e in engines)
    assert np.array_equal(tot, whole.query_counts(q).astype(np.uint32))
    grp.close()
    for e in engines + [whole]:
        e.close()


def test_group_rejects_wrong_shards(native):
    S = 8
    a = native.Engine(K=31, S=S, W=8, H=3, slot_begin=0, slot_end=128)
    b = native.Engine(K=31, S=S, W=8, H=3, slot_begin=100, slot_end=256)      # not rank 1's range
    with pytest.raises(native.NiqkiError) as ei:
        native.Group([a, b])
    assert ei.value.code == 1 and "slots [128, 256)" in str(ei.value)
    a.close()
    b.close()


def test_rccl_world1_group(native, po):
    """One rank, transport = librccl (all-to-all, all-gather and reduce-scatter on a 1-rank
    communicator): the C ABI's RCCL path end to end on the one GPU a test box has."""
    import torch
    S, W, N, NQ, MS = 10, 8, 900, 16, 50
    sk, q = make_data(S, W, N, NQ, 77)
    e = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    grp = native.Group([e], first_rank=0, world=1, group_id=native.group_new_id())
    assert grp.stat("rccl") == 1 and grp.stat("sparse") == 1
    got = run_group(native, torch, [e], grp, sk, q, NQ)
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    for i in range(NQ):
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(got[i][0], ehc) and np.array_equal(got[i][1], ehg), i
    # dense exchange through RCCL on the same index
    grp.set_option("exchange", 2)
    dev = torch.device("cuda")
    res = grp.query([torch.from_numpy(q.copy()).to(dev)], NQ)
    off, hc, hg = res[0]
    for i in range(NQ):
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], got[i][0]) and np.array_equal(hg[int(off[i]):int(off[i + 1])], got[i][1])
    grp.close()
    e.close()
