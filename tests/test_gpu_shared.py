"""GPU: the reference's threading contract at the boundary.  Its drivers call compute_sketch / insert_sketch /
query_sketch from every thread of an `omp parallel` region on one Index (src/niqki_index.cpp:391-401, :415-428,
:479-490, :525-538).  niqki_*_shared take such concurrent callers on ONE handle and combine them into batches;
answers = the single-caller calls = the oracle."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run_threads(n, fn):
    errs, out = [], [None] * n

    def body(i):
        try:
            out[i] = fn(i)
        except Exception as e:      # noqa: BLE001
            errs.append((i, e))
    ts = [threading.Thread(target=body, args=(i,)) for i in range(n)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs[:3]
    return out


def test_concurrent_sketch_insert_query_on_one_handle(native, po):
    K, S, W, H, J = 31, 10, 12, 4, 0.1
    p = po.make_params(K, S, W, H, J)
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    n = 48
    genomes = [native.synth_genome_host(17, g // 6, g % 6, 120 * (g % 6), 30_000 + 7 * g) for g in range(n)]
    exp_sk = [po.compute_sketch(p, g) for g in genomes]
    # every record sketched by a thread of its own (ctypes drops the GIL in the call: the calls really overlap)
    sk = run_threads(n, lambda i: e.sketch_shared(genomes[i]))
    for i in range(n):
        assert np.array_equal(sk[i], exp_sk[i]), i
    st = e.shared_stats()
    assert st["requests"] == n and st["batches"] <= n
    # inserts from all threads at once: ids in arrival order, each used once (src/niqki_index.cpp:396-401)
    gids = run_threads(n, lambda i: e.insert_shared(sk[i]))
    assert sorted(gids) == list(range(n)) and e.n_genomes == n
    order = np.argsort(gids)                     # genome id -> the record that got it
    ix = po.Index(p, np.stack([exp_sk[i] for i in order]))
    # queries: sketches and raw sequences, from all threads at once, against the oracle's index in the same id order
    queries = [native.synth_genome_host(17, g // 6, 40 + g, 90, 25_000) for g in range(n)]
    res = run_threads(n, lambda i: e.query_shared(exp_sk[i], capacity=4) if i % 2 else e.query_sequence_shared(queries[i], capacity=4))
    n_hits = 0
    for i in range(n):
        ehc, ehg = ix.query(exp_sk[i] if i % 2 else po.compute_sketch(p, queries[i]))
        assert np.array_equal(res[i][0], ehc) and np.array_equal(res[i][1], ehg), i
        n_hits += len(ehc)
    assert n_hits > n          # (the capacity of 4 was exceeded somewhere: the retry path ran)
    st = e.shared_stats()
    assert st["requests"] == 3 * n + sum(1 for i in range(n) if len(res[i][0]) > 4)
    assert st["largest_batch"] >= 2, st      # callers really were combined
    # the single-caller calls still work afterwards, and agree
    off, hc, hg = e.query(np.stack(exp_sk[:5]))
    for i in range(5):
        ehc, ehg = ix.query(exp_sk[i])
        assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg)
    e.close()


def test_stream_priority_option_changes_no_result(native, po):
    """option "stream_priority": the handle moves to a stream of its own at the top / bottom of the device's priority
    range (also after niqki_set_stream); answers and the stream contract stay what they were."""
    import torch
    K, S, W, H, J = 31, 9, 10, 4, 0.05
    p = po.make_params(K, S, W, H, J)
    genomes = [native.synth_genome_host(23, g // 4, g % 4, 200 * (g % 4), 20_000) for g in range(16)]
    exp = np.stack([po.compute_sketch(p, g) for g in genomes])
    ix = po.Index(p, exp)
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    e.set_stream(torch.cuda.current_stream().cuda_stream)      # a caller's stream first ...
    for prio in (1, -1, 0):
        e.set_option("stream_priority", prio)                   # ... then a stream of the handle's own
        assert e.get_stream() not in (0, torch.cuda.current_stream().cuda_stream)
    with pytest.raises(native.NiqkiError):
        e.set_option("stream_priority", 2)
    sk = e.sketch(genomes)
    assert np.array_equal(sk, exp)
    e.insert(sk)
    off, hc, hg = e.query(exp)
    for i in range(len(genomes)):
        ehc, ehg = ix.query(exp[i])
        assert np.array_equal(hc[off[i]:off[i + 1]], ehc) and np.array_equal(hg[off[i]:off[i + 1]], ehg)
    e.set_stream(torch.cuda.current_stream().cuda_stream)      # and back onto the caller's
    off2, hc2, hg2 = e.query(exp)
    assert np.array_equal(off2, off) and np.array_equal(hc2, hc) and np.array_equal(hg2, hg)
    e.close()


@pytest.mark.parametrize("mode", [["--no-legs"], ["--no-overlap", "--no-legs"], ["--priority-streams", "--no-legs"],
                                  ["--no-cpu", "--no-extra", "--no-pmc"]])
def test_bench_modes_on_a_small_index(tmp_path, mode):
    """bench.py's other modes at a small size (the timed-steps-only run that tools/profile_round.sh traces -- the default
    step with the next batch's sketch kernel beside the query, the same one after the other, the default with priority
    streams -- and a run with the legs): one JSON line with the record's fields; the roofline fraction is a fraction."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--genomes", "3000", "--batch", "256", "--steps", "3", "--warmup", "1"] + mode
    r = subprocess.run(cmd, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1
    j = json.loads(lines[0])
    rf = j["roofline"]
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "genomes/s" and j["scaling"] == "weak"
    assert rf["bound"] == "hbm" and 0 < rf["frac"] <= 1.0 and rf["frac_basis"] and rf["frac_algorithmic"] > 0 and rf["frac_layout_min"] > 0
    assert rf["launches"] == 3 and j["kernels"]["sketch"]["launches"] == 3
    assert j["config"]["sketch_beside_query"] == ("--no-overlap" not in mode)
    if "--no-overlap" in mode:
        assert "serial_step" not in j and rf["measured_in"] == "the timed steps"
    else:
        assert j["serial_step"]["value"] > 0 and j["kernels_beside_each_other"]["gather"]["launches"] == 3 and "more steps" in rf["measured_in"]
    assert j["budget"]["used_s"] > 0 and j["budget"]["dropped"] == []
    if "--no-legs" in mode:
        assert j["cpu_baseline"] is None and j["end_to_end_d2h"] is None and "extra_workloads" not in j
    else:
        assert j["end_to_end_d2h"]["value"] > 0


def test_shared_entry_points_under_mixed_load(native, po):
    """64 threads, each a random sequence of sketch / query / query-by-sequence calls on ONE handle while the others do
    the same (an index that does not change during the phase: every answer can be checked); then a phase of
    concurrent inserts; then queries again against the grown index."""
    import random
    K, S, W, H, J = 31, 8, 10, 4, 0.05
    p = po.make_params(K, S, W, H, J)
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    genomes = [native.synth_genome_host(29, g // 5, g % 5, 150 * (g % 5), 6_000 + 11 * g) for g in range(40)]
    exp_sk = [po.compute_sketch(p, g) for g in genomes]
    e.insert(np.stack(exp_sk[:20]))
    ix = po.Index(p, np.stack(exp_sk[:20]))

    def worker(t, index, n_ops):
        rnd = random.Random(1000 + t)
        for _ in range(n_ops):
            g = rnd.randrange(len(genomes))
            op = rnd.randrange(3)
            if op == 0:
                assert np.array_equal(e.sketch_shared(genomes[g]), exp_sk[g])
            else:
                hc, hg = e.query_shared(exp_sk[g], capacity=rnd.choice([1, 8, 64])) if op == 1 else e.query_sequence_shared(genomes[g], capacity=8)
                ehc, ehg = index.query(exp_sk[g])
                assert np.array_equal(hc, ehc) and np.array_equal(hg, ehg), (t, g, op)
        return True
    assert all(run_threads(64, lambda t: worker(t, ix, 12)))
    gids = run_threads(20, lambda i: e.insert_shared(exp_sk[20 + i]))
    assert sorted(gids) == list(range(20, 40))
    order = [20 + int(i) for i in np.argsort(gids)]
    ix2 = po.Index(p, np.stack(exp_sk[:20] + [exp_sk[i] for i in order]))
    assert all(run_threads(64, lambda t: worker(t, ix2, 6)))
    st = e.shared_stats()
    assert st["largest_batch"] >= 4 and st["batches"] < st["requests"]
    e.close()


def test_shared_queries_with_more_hits_than_the_first_internal_capacity(native, po):
    """A combined batch whose queries have far more than 64 hits each: the batch's first capacity (64 per query) is too
    small, niqki_query says NIQKI_E_CAPACITY with exact offsets, and the combiner must run the batch again with room
    for all -- inside the library, whatever capacity each caller passed (the header's contract: *n_hits may exceed
    capacity, the first `capacity` hits are written).  300 near-identical genomes, 24 threads."""
    K, S, W, H, J = 31, 8, 10, 4, 0.3
    p = po.make_params(K, S, W, H, J)
    rng = np.random.default_rng(11)
    base = rng.integers(0, 1 << W, 1 << S).astype(np.int32)
    sk = np.tile(base, (300, 1))
    flip = rng.random(sk.shape) < 0.1                       # every genome differs from the base in ~10 % of its slots
    sk[flip] = rng.integers(0, 1 << W, int(flip.sum()))
    e = native.Engine(K=K, S=S, W=W, H=H, J=J)
    e.insert(sk)
    ix = po.Index(p, sk)
    L = native.lib()
    import ctypes as C

    def one(i):
        q = np.ascontiguousarray(sk[(7 * i) % 300])
        cap = [0, 3, 50, 400][i % 4]
        n = C.c_uint64(0)
        hc, hg = np.zeros(max(cap, 1), np.uint32), np.zeros(max(cap, 1), np.uint32)
        rc = L.niqki_query_shared(e.h, q.ctypes.data, C.byref(n), hc.ctypes.data if cap else None, hg.ctypes.data if cap else None, cap)
        assert rc == 0, (i, rc)                             # never NIQKI_E_CAPACITY: that is the library's to handle
        ehc, ehg = ix.query(q)
        assert n.value == len(ehc) and len(ehc) > 200, (i, n.value, len(ehc))
        w = min(cap, len(ehc))
        assert np.array_equal(hc[:w], ehc[:w]) and np.array_equal(hg[:w], ehg[:w]), i
        return int(n.value)
    hits = run_threads(24, one)
    assert min(hits) > 200
    st = e.shared_stats()
    assert st["requests"] == 24
    e.close()
