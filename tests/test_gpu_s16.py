"""GPU: S = 16 (2^16 sketch slots), the reference's lF > 15 branch with uint32 counters
(src/niqki_index.cpp:668-682).  A count can reach 2^16, one more than a u16 counter holds: the
gather kernel walks the slots in two passes into two counter planes that the hit kernels sum in 32
bits; the sketch kernel keeps one half of the 256 KB of cells per workgroup and densifies in global
memory.  Golden D5 comes from the real reference (oracle/make_goldens_s16.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def d5():
    vec = np.load(os.path.join(GOLD, "reference_s16.npz"))
    meta = json.load(open(os.path.join(GOLD, "reference_s16.json")))
    return vec, meta["D5"], meta["seed"]


def test_s16_sketch_insert_query_dump_vs_reference(native, po, d5):
    vec, m, seed = d5
    e = native.Engine(K=m["K"], S=m["S"], W=m["W"], H=m["H"], J=m["J"])
    assert e.min_score == int(vec["D5_min_score"][0])
    genomes = [native.synth_genome_host(seed, a, b, c, m["len"]) for a, b, c in zip(m["fam"], m["mem"], m["rate"])]
    sk = e.sketch(genomes)
    assert np.array_equal(sk, vec["D5_sketches"].astype(np.int32))
    # a 500-base record: 470 k-mers, 65 066 cells filled by densification (in global memory at S = 16)
    assert np.array_equal(e.sketch([vec["D5_short_seq"]])[0], vec["D5_short_sketch"].astype(np.int32))
    assert np.array_equal(e.densify(np.where(np.arange(1 << 16) % 7 == 0, sk[0], -1).astype(np.int32).reshape(1, -1))[0],
                          po.densify(po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"]),
                                     np.where(np.arange(1 << 16) % 7 == 0, sk[0], -1).astype(np.int32))[0])
    e.insert(sk)
    qsk = vec["D5_qsketches"].astype(np.int32)
    off, hc, hg = e.query(qsk)
    assert np.array_equal(off, vec["D5_hit_off"])
    assert np.array_equal(hc, vec["D5_hit_counts"]) and np.array_equal(hg, vec["D5_hit_gids"])
    assert int(hc.max()) == 1 << 16                     # the self hits: F matching slots
    # sequences in, hits out
    queries = [native.synth_genome_host(seed, a, b, c, m["len"]) for a, b, c in zip(m["qfam"], m["qmem"], m["qrate"])]
    off2, hc2, hg2 = e.query_sequences(queries)
    assert np.array_equal(off2, off) and np.array_equal(hc2, hc) and np.array_equal(hg2, hg)
    # exact dense counters (uint32) against the oracle
    p = po.make_params(m["K"], m["S"], m["W"], m["H"], m["J"])
    ix = po.Index(p, sk)
    c32 = e.query_counts32(qsk)
    for q in range(qsk.shape[0]):
        assert np.array_equal(c32[q], ix.counts(qsk[q])), q
    # the u16 counter calls cannot hold 2^16: refused, not wrapped
    with pytest.raises(native.NiqkiError) as ei:
        e.query_counts(qsk[:1])
    assert ei.value.code == 1
    # the matrix has uint16 counters in the reference whatever S (:572): identical genomes read 0
    mat = e.matrix_range(0, len(genomes))
    exp = ix.matrix_range(0, len(genomes))
    assert np.array_equal(mat, exp.T) and int(mat[0, 0]) == 0
    # dump bytes = the reference's, and back
    raw = e.export_dump() + "".join("g%d\n" % i for i in range(len(genomes))).encode()
    assert len(raw) == m["dump_len"] and hashlib.md5(raw).hexdigest() == m["dump_md5"]
    e2 = native.Engine.import_dump(raw)
    assert (e2.S, e2.W, e2.n_genomes) == (16, m["W"], len(genomes))
    assert all(np.array_equal(a, b) for a, b in zip(e2.query(qsk), (off, hc, hg)))
    e2.close()
    e.close()


def test_s16_many_genomes_two_tiles_and_device_path(native, po):
    """Random sketches: 70 000 genomes (two counter tiles) at S = 16 would be 9 GB of int32 -- 3000 genomes with a
    small tile instead (several tiles, ragged), device-memory query path, locality order off/on irrelevant."""
    import torch
    rng = np.random.default_rng(16)
    S, W, N = 16, 6, 700
    F = 1 << S
    fam = rng.integers(0, 1 << W, (5, F)).astype(np.int32)
    sk = fam[rng.integers(0, 5, N)].copy()
    noise = rng.random((N, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[0] = fam[0]
    e = native.Engine(K=31, S=S, W=W, H=3, J=0.7, tile_genomes=256)
    e.insert(sk)
    q = np.concatenate([fam, sk[:3]])
    p = po.make_params(31, S, W, 3, 0.7)
    ix = po.Index(p, sk)
    off, hc, hg = e.query(q)
    for i in range(q.shape[0]):
        ehc, ehg = ix.query(q[i])
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg), i
    assert int(hc.max()) == F
    dev = torch.device("cuda")
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    dq = torch.from_numpy(q).to(dev)
    cap = q.shape[0] * N
    ho = torch.zeros(q.shape[0] + 1, dtype=torch.int64, device=dev)
    dc = torch.zeros(cap, dtype=torch.int32, device=dev)
    dg = torch.zeros(cap, dtype=torch.int32, device=dev)
    e.query_dev(dq, q.shape[0], ho, dc, dg, cap)
    e.synchronize()
    assert np.array_equal(ho.cpu().numpy().astype(np.uint64), off)
    assert np.array_equal(dc.cpu().numpy()[:int(off[-1])].astype(np.uint32), hc)
    # the same index paged (pages never straddle the two halves of the slots: each half's counters
    # add up in a plane of its own): same hits, same exact counters
    c32 = e.query_counts32(q)
    pg = native.Engine(K=31, S=S, W=W, H=3, J=0.7, tile_genomes=256, resident_mib=8)
    pg.insert(sk)
    assert pg.stat("pages") >= 4 and pg.stat("pages") % 2 == 0
    poff, phc, phg = pg.query(q)
    assert np.array_equal(poff, off) and np.array_equal(phc, hc) and np.array_equal(phg, hg)
    assert np.array_equal(pg.query_counts32(q), c32)
    assert np.array_equal(c32[0].astype(np.int64), ix.counts(q[0]))
    pg.close()
    e.close()


@pytest.mark.parametrize("world,exchange", [(2, "sparse"), (2, "dense"), (4, "sparse"), (4, "overflow")])
def test_s16_slot_shards_equal_the_whole_index(native, d5, world, exchange):
    """S = 16 over slot shards (niqki_group_*): a shard counts at most 2^15 slots in u16, the cross-shard sums
    reach 2^16 and travel as u32 -- candidates' sums in the sparse exchange, widened counter rows in the dense
    one (two planes for the hit kernels).  Golden D5's hit lists, counts of 2^16 included."""
    import torch
    vec, m, seed = d5
    dev = torch.device("cuda")
    S, F = m["S"], 1 << m["S"]
    sk = vec["D5_sketches"].astype(np.int32)
    qsk = vec["D5_qsketches"].astype(np.int32)
    engines = []
    for r in range(world):
        b, e_ = native.group_slot_range(r, world, S)
        sh = native.Engine(K=m["K"], S=S, W=m["W"], H=m["H"], J=m["J"], slot_begin=b, slot_end=e_)
        sh.set_stream(torch.cuda.current_stream().cuda_stream)
        engines.append(sh)
    grp = native.Group(engines)
    grp.set_option("exchange", 2 if exchange == "dense" else 1)
    if exchange == "overflow":
        grp.set_option("cand_cap", 2)
    n = sk.shape[0]
    per_i = -(-n // world)
    pad = np.full((world * per_i, F), -1, np.int32)
    pad[:n] = sk
    grp.insert_dev([torch.from_numpy(pad[r * per_i:(r + 1) * per_i].copy()).to(dev) for r in range(world)], per_i, n)
    nq = qsk.shape[0]
    per = -(-nq // world)
    padq = np.full((world * per, F), -1, np.int32)
    padq[:nq] = qsk
    res = grp.query([torch.from_numpy(padq[r * per:(r + 1) * per].copy()).to(dev) for r in range(world)], per)
    off, hc, hg = vec["D5_hit_off"], vec["D5_hit_counts"], vec["D5_hit_gids"]
    seen_full = 0
    for q in range(nq):
        r, j = divmod(q, per)
        o, c, g_ = res[r]
        lo, hi = int(off[q]), int(off[q + 1])
        assert np.array_equal(c[int(o[j]):int(o[j + 1])], hc[lo:hi]) and np.array_equal(g_[int(o[j]):int(o[j + 1])], hg[lo:hi]), (world, exchange, q)
        seen_full += int((c[int(o[j]):int(o[j + 1])] == 1 << 16).sum())
    assert seen_full >= 1
    assert (grp.stat("overflows") >= 1) == (exchange == "overflow")
    grp.close()
    for sh in engines:
        sh.close()
    # one shard cannot hold 2^16 slots in u16 counters
    whole = native.Engine(K=m["K"], S=S, W=m["W"], H=m["H"], J=m["J"])
    with pytest.raises(native.NiqkiError):
        native.Group([whole])
    whole.close()
