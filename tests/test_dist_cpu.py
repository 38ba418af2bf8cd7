"""CPU, world_size 2 over gloo: the slot-shard exchange protocol (tests/torch_exchange.py, the torch
restatement of niqki_amd/csrc/nq_group.hip)
(all_gather of sketches, reduce of packed u16 hit vectors, per-rank threshold)
gives exactly the single-index answer.  The local compute is played by an
oracle-backed stand-in engine with the *_dev method names of niqki_amd.Engine."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from torch_exchange import TorchExchange as ShardedQuery, padded_batch, slot_range

S, W, N, NQ = 9, 8, 333, 6


class OracleShard:
    def __init__(self, po, p, sketches, sb, se):
        sk = sketches.copy()
        sk[:, :sb] = -1
        sk[:, se:] = -1          # only this shard's slots are inserted
        self.ix = po.Index(p, sk)
        self.sb, self.se, self.po, self.p = sb, se, po, p

    def query_counts_dev(self, sk, nq, counts, stride):
        a = sk.numpy()
        out = counts.numpy().view(np.uint16)
        for q in range(nq):
            s = a[q].copy()
            s[:self.sb] = -1
            s[self.se:] = -1
            out[q, :self.ix.n] = self.ix.counts(s).astype(np.uint16)

    def candidates_dev(self, counts, nq, stride, n_gids, thr, cap, cand, ncand):
        c = counts.numpy().view(np.uint16)
        cand.fill_(-1)
        for q in range(nq):
            ids = np.nonzero(c[q, :n_gids] >= thr)[0][::-1]  # any order is allowed
            ncand[q] = len(ids)
            k = min(len(ids), cap)
            cand[q, :k] = torch.from_numpy(ids[:k].astype(np.int32).copy())

    min_score = 5

    def hits_from_counts_dev(self, red, per, stride, g0, n, hit_off, hc, hg, cap):
        c = red.numpy().view(np.uint16)
        L = self.po.lib()
        tot = 0
        hit_off[0] = 0
        for q in range(per):
            row = np.ascontiguousarray(c[q, g0:g0 + n].astype(np.uint32))
            oc = np.empty(n, np.uint32)
            og = np.empty(n, np.uint32)
            k = L.nqo_hits_from_counts(row.ctypes.data, n, self.p.min_score, oc.ctypes.data, og.ctypes.data, n)
            hc[tot:tot + k] = torch.from_numpy(oc[:k].astype(np.int32))
            hg[tot:tot + k] = torch.from_numpy(og[:k].astype(np.int32))
            tot += k
            hit_off[q + 1] = tot


def _data():
    rng = np.random.default_rng(77)
    sk = rng.integers(0, 1 << W, (N, 1 << S)).astype(np.int32)
    sk[7, 5:40] = -1
    sk[100] = sk[8]
    q = rng.integers(0, 1 << W, (NQ, 1 << S)).astype(np.int32)
    q[0] = sk[8]
    q[1] = sk[200]
    q[1, ::2] = sk[201, ::2]
    return sk, q


def _worker(rank, world, port, exchange, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import pyoracle as po
    p = po.make_params(31, S, W, 4, 0.0)
    p.min_score = 5
    sk, q = _data()
    F = 1 << S
    sb, se = slot_range(rank, world, F)
    eng = OracleShard(po, p, sk, sb, se)
    sq = ShardedQuery(eng, N, F, torch.device("cpu"), exchange=exchange.split("+")[0],
                      cand_cap=64 if exchange.startswith("sparse") else 8, compact_sketches=exchange.endswith("+i16"))
    per = padded_batch(NQ, world)
    mine = torch.from_numpy(q[rank * per:(rank + 1) * per].copy())
    hit_off = torch.zeros(per + 1, dtype=torch.int64)
    hc = torch.zeros(per * N, dtype=torch.int32)
    hg = torch.zeros(per * N, dtype=torch.int32)
    whole = po.Index(p, sk)

    def answers_match():
        good = True
        for i in range(per):
            ehc, ehg = whole.query(q[rank * per + i], min_score=5)
            lo, hi = int(hit_off[i]), int(hit_off[i + 1])
            good &= np.array_equal(hc[lo:hi].numpy().astype(np.uint32), ehc)
            good &= np.array_equal(hg[lo:hi].numpy().astype(np.uint32), ehg)
        return good

    sq.step(mine, hit_off, hc, hg, per * N)
    ok = answers_match()
    if exchange.startswith("sparse"):
        ok &= int(sq.overflow.item()) == 0
        # a capacity that is too small is reported and the step redone densely: never silently wrong
        sq2 = ShardedQuery(eng, N, F, torch.device("cpu"), exchange="sparse", cand_cap=1)
        hit_off.zero_()
        sq2.step(mine, hit_off, hc, hg, per * N)
        ok &= int(sq2.overflow.item()) == 1
        ok &= answers_match()
    ret[rank] = bool(ok)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("exchange", ["reduce_scatter", "all_to_all", "sparse", "sparse+i16"])
def test_slot_sharded_query_world2(exchange):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), exchange, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_group_plan_is_the_products():
    """niqki_group_plan_batch (pure arithmetic inside libniqki_hip.so, the function nq_group.hip calls for every
    batch): thresholds, the sparse / dense decision and the sizes of what travels, for the north-star group."""
    import niqki_amd
    p = niqki_amd.group_plan(8, 15, 3276, 0, 512, 100_000, 256)
    assert (p.sparse, p.cand_threshold, p.surv_threshold, p.slice_slots) == (1, 410, 205, 4096)
    assert p.slice_bytes == 512 * 4096 * 2 and p.row_stride == niqki_amd.row_stride(100_000)
    assert p.cand_blob_bytes == (4096 * 256 + 2 * 4096) * 4 and p.sum_words == 512 * 8 * 256 // 2
    assert niqki_amd.group_plan(8, 15, 3276, 2, 512, 100_000, 256).sparse == 0            # option: dense
    assert niqki_amd.group_plan(8, 15, 3276, 2, 512, 100_000, 256).sum_words == 512 * (p.row_stride // 2)
    assert niqki_amd.group_plan(8, 15, 20, 0, 512, 100_000, 256).sparse == 0              # min_score < 4 * world: dense
    assert niqki_amd.group_plan(8, 15, 5, 1, 512, 100_000, 256).sparse == 0               # min_score < world: never sparse
    assert niqki_amd.group_plan(2, 16, 6553, 0, 64, 1000, 256).sum_words == 64 * 2 * 256  # S = 16: sums travel as u32
    with pytest.raises(niqki_amd.NiqkiError):
        niqki_amd.group_plan(0, 15, 1, 0, 1, 1, 2)


def test_slot_ranges_partition_the_sketch():
    for F in (32768, 4096, 64):
        for world in (1, 2, 3, 4, 8):
            cuts = [slot_range(r, world, F) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == F
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
    assert padded_batch(10, 4) == 3
    # the pure-Python cut of niqki_amd/dist.py (usable without the built library) is the library's own
    import niqki_amd
    from niqki_amd import dist as nd
    for S in (6, 12, 15, 16):
        for world in (1, 2, 3, 5, 8, 64):
            for r in range(world):
                assert nd.slot_range(r, world, 1 << S) == niqki_amd.group_slot_range(r, world, S)


def test_bench_launcher_starts_ranks_and_relays_their_exit_code(tmp_path):
    """`python bench.py --gpus 2` without WORLD_SIZE: the process becomes a launcher BEFORE it imports torch or touches
    a GPU -- it starts two ranks of itself under torch.distributed.run as a child and hands on their exit code.  Here
    (no GPU) the ranks fail at their first device call: non-zero exit, no JSON line, and the launcher's own message."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NIQKI_BENCH_LAUNCH_TIMEOUT"] = "240"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--genomes", "500", "--steps", "1", "--warmup", "1",
                        "--batch", "64", "--no-legs"], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    err = r.stderr.decode(errors="replace")
    assert "--gpus 2 without a launcher" in err and "torch.distributed.run" in err and "--nproc-per-node 2" in err
    import torch
    if not torch.cuda.is_available():
        assert r.returncode not in (0, 124) and not r.stdout.strip()
    else:
        assert r.returncode == 0 and r.stdout.decode().count("\n") == 1
