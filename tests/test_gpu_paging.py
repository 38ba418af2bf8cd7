"""GPU: indexes larger than the memory they are given (option "resident_bytes" / niqki_params.resident_mib,
SURVEY.md 8f row 4): the sketch store in page-locked host memory, the inverted index built one page of
slots at a time, the gather kernel accumulating the hit counters over the pages.  A budget that forces
several pages must give exactly the answers of the resident index and of the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def data(S, W, N, nq, seed):
    rng = np.random.default_rng(seed)
    F = 1 << S
    fam = rng.integers(0, 1 << W, (20, F)).astype(np.int32)
    sk = fam[rng.integers(0, 20, N)].copy()
    noise = rng.random((N, F)) < 0.35
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    sk[rng.random((N, F)) < 0.01] = -1
    q = fam[rng.integers(0, 20, nq)].copy()
    m = rng.random((nq, F)) < 0.2
    q[m] = rng.integers(0, 1 << W, int(m.sum()))
    q[3] = -1
    return sk, q


@pytest.mark.parametrize("how", ["option", "param"])
def test_paged_index_equals_resident_index_and_oracle(native, po, how):
    S, W, N, NQ, MS = 10, 8, 3000, 70, 200
    sk, q = data(S, W, N, NQ, 3)
    res = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    res.insert(sk)
    if how == "option":
        pg = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
        pg.set_option("resident_bytes", 4 << 20)
    else:
        pg = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS, resident_mib=4)
    # ~30 KB per slot (store row + table row + id lists) -> pages of 128 of the 1024 slots: 8 pages
    for a in range(0, N, 700):               # several inserts: the host store grows
        pg.insert(sk[a:a + 700])
    assert pg.n_genomes == N
    assert pg.stat("pages") >= 4 and pg.stat("page_slots") * pg.stat("pages") >= 1 << S and res.stat("pages") == 1
    c_res, c_pg = res.query_counts(q), pg.query_counts(q)
    assert np.array_equal(c_res, c_pg)
    h_res, h_pg = res.query(q), pg.query(q)
    assert all(np.array_equal(a, b) for a, b in zip(h_res, h_pg))
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    off, hc, hg = h_pg
    for i in range(NQ):
        assert np.array_equal(c_pg[i].astype(np.uint32), ix.counts(q[i])), i
        ehc, ehg = ix.query(q[i], min_score=MS)
        assert np.array_equal(hc[int(off[i]):int(off[i + 1])], ehc) and np.array_equal(hg[int(off[i]):int(off[i + 1])], ehg), i
    assert int(off[NQ]) > 100
    # stored sketches come back from the host store
    assert np.array_equal(pg.get_sketches(100, 37), np.where(sk[100:137] < (1 << W), sk[100:137], -1))
    # more genomes after queries: the pages are rebuilt
    pg.insert(sk[:50])
    res.insert(sk[:50])
    assert np.array_equal(pg.query_counts(q[:5]), res.query_counts(q[:5]))
    # matrix rows and the dump of a paged index: the stored sketches are read from the host store, the dump is
    # exported page after page (src/niqki_index.cpp:570-628, :42-59); both equal the resident index's
    assert np.array_equal(pg.matrix_range(0, 300), res.matrix_range(0, 300))
    assert np.array_equal(pg.matrix_range(N - 7, N + 50), res.matrix_range(N - 7, N + 50))
    assert pg.export_dump() == res.export_dump()
    # what a paged handle does not offer is refused, not answered wrongly
    with pytest.raises(native.NiqkiError) as ei:
        pg.gathered(q[:2])
    assert ei.value.code == 5
    with pytest.raises(native.NiqkiError):
        res.set_option("resident_bytes", 1 << 20)     # only before the first insert
    pg.close()
    res.close()


def test_dump_loaded_into_a_paged_handle(native, po):
    S, W, N, MS = 9, 8, 1500, 100
    sk, q = data(S, W, N, 12, 9)
    res = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    res.insert(sk)
    raw = res.export_dump()
    pg = native.Engine.import_dump(raw, resident_mib=2)      # ~16 KB per slot -> 4 pages of 128 slots
    assert pg.n_genomes == N and pg.min_score == MS and pg.stat("pages") >= 4
    assert np.array_equal(pg.query_counts(q), res.query_counts(q))
    assert all(np.array_equal(a, b) for a, b in zip(pg.query(q), res.query(q)))
    assert np.array_equal(pg.get_sketches(0, N), res.get_sketches(0, N))
    pg.close()
    res.close()


def test_paged_query_sequences_and_big_tiles(native, po):
    """Sequences in, several pages, two counter tiles (70 000 genomes), the locality order on."""
    rng = np.random.default_rng(4)
    S, W, N = 6, 8, 70000
    F = 1 << S
    fam = rng.integers(0, 1 << W, (300, F)).astype(np.int32)
    sk = fam[np.arange(N) // 234].copy()
    noise = rng.random((N, F)) < 0.35
    sk[noise] = rng.integers(0, 1 << W, int(noise.sum()))
    res = native.Engine(K=31, S=S, W=W, H=3, J=0.5)
    pg = native.Engine(K=31, S=S, W=W, H=3, J=0.5, resident_mib=16)   # ~0.45 MB per slot -> two pages of 32 slots
    for a in range(0, N, 10000):
        res.insert(sk[a:a + 10000])
        pg.insert(sk[a:a + 10000])
    q = np.concatenate([fam[[0, 7, 299]], sk[[0, 65535, 65536, N - 1]]])
    q = np.concatenate([q] * 10)              # 70 queries: launches big enough for the locality order
    assert pg.stat("pages") == 2
    assert np.array_equal(pg.query_counts(q), res.query_counts(q))
    assert all(np.array_equal(a, b) for a, b in zip(pg.query(q), res.query(q)))
    pg.close()
    res.close()
