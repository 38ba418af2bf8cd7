"""GPU: randomized cross-check of the query path's launch forms on mid-sized random indexes: counter
tiles (size, ranges / stripes of 1 .. 64 genomes), table look-ups inside the gather kernel or by the
pre-pass (random look-ups / streamed rows), locality order on or off, genomes added after a build
(rebuild, or -- main index of >= 4096 genomes -- the delta segment, asserted), candidates picked inside
the gather kernel -- all against one plain engine and
against the oracle.  NIQKI_FUZZ_SCALE=k runs k times as many seeds."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
SCALE = int(os.environ.get("NIQKI_FUZZ_SCALE", "1"))


@pytest.mark.parametrize("seed", range(8 * SCALE))
def test_launch_forms_agree(native, po, seed, monkeypatch):
    import torch
    rng = np.random.default_rng(5000 + seed)
    S = int(rng.choice([5, 6, 7]))
    W = int(rng.choice([6, 11, 12]))
    F = 1 << S
    n = int(rng.integers(100, 3000))
    n_late = int(rng.choice([0, 0, 37]))
    if n_late and seed % 2:
        n = int(rng.integers(4096 + n_late, 5200))      # a main index of >= 4096 genomes: the late ones get a DELTA segment
    tile = int(rng.choice([0, 64, 192, 256, 1024]))
    if n >= 4096 and tile == 64:
        tile = 192                                       # (the dump merges at most 64 tiles)
    stripe = int(rng.choice([0, 1, 2, 8, 32, 64]))
    nq = int(rng.choice([7, 70, 600, 1100, 2100]))
    fam = rng.integers(0, 1 << W, (6, F)).astype(np.int32)
    sk = fam[(np.arange(n) // int(rng.integers(5, 90))) % 6].copy()
    noise = rng.random((n, F)) < 0.3
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((n, F)) < 0.01] = -1
    q = sk[rng.integers(0, n, nq)].copy()
    m = rng.random((nq, F)) < 0.15
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[0] = -1
    J = float(rng.choice([0.1, 0.4]))
    p = po.make_params(31, S, W, 4, J)
    ix = po.Index(p, sk)

    def engine(tile_, stripe_, prepass, order):
        monkeypatch.setenv("NIQKI_TILE_STRIPE", str(stripe_))
        e = native.Engine(K=31, S=S, W=W, H=4, J=J, tile_genomes=tile_)
        e.set_option("lookup_prepass", prepass)
        e.set_option("query_order", order)
        e.insert(sk[:n - n_late])
        if n_late:
            e.query(q[:1])
            e.insert(sk[n - n_late:])
        return e

    plain = engine(0, 0, 0, 0)
    ref_cnt, ref_hits = plain.query_counts(q), plain.query(q)
    for i in (0, 1, nq // 2, nq - 1):
        assert np.array_equal(ref_cnt[i].astype(np.uint32), ix.counts(q[i])), (seed, i)
    plain.close()
    e = engine(tile, stripe, int(rng.choice([-1, 0, 1])), int(rng.choice([0, 2])))
    cnt, hits = e.query_counts(q), e.query(q)
    # genomes that arrived after a query on a main index of >= 4096: indexed by the delta segment, not a rebuild
    assert e.stat("delta_genomes") == (n_late if (n_late and n - n_late >= 4096) else 0), (seed, n, n_late)
    assert np.array_equal(cnt, ref_cnt), (seed, S, W, n, tile, stripe, nq, n_late)
    assert all(np.array_equal(a, b) for a, b in zip(hits, ref_hits)), seed
    # candidates from inside the gather kernel = the counters' threshold
    dev = torch.device("cuda")
    e.set_stream(torch.cuda.current_stream().cuda_stream)
    stride = native.row_stride(n)
    thr, cap = max(1, int(F * J / 2)), 64
    dq = torch.from_numpy(q).to(dev)
    c1 = torch.zeros((nq, stride), dtype=torch.int16, device=dev)
    cand = torch.full((nq, cap), 7, dtype=torch.int32, device=dev)
    nc = torch.full((nq,), 7, dtype=torch.int32, device=dev)
    e.query_counts_candidates_dev(dq, nq, c1, stride, thr, cap, cand, nc)
    e.synchronize()
    assert np.array_equal(c1[:, :n].cpu().numpy().view(np.uint16), ref_cnt), seed
    cand, nc = cand.cpu().numpy(), nc.cpu().numpy()
    for i in (0, 1, nq // 2, nq - 1):
        want = np.nonzero(ref_cnt[i] >= thr)[0]
        assert nc[i] == len(want), (seed, i)
        k = min(len(want), cap)
        assert set(cand[i, :k].tolist()) <= set(want.tolist()) and len(set(cand[i, :k].tolist())) == k and (cand[i, k:] == -1).all(), (seed, i)
    assert e.export_dump() == ix.dump_bytes(), seed
    e.close()
