"""GPU: the N > 1 path with one PROCESS per rank, as bench.py runs it under torch.distributed.run
(niqki_group_create(..., n_local = 1, ..., id)).  RCCL refuses ranks that share a device, so on a one-GPU
box the processes use the library's ipc transport (NIQKI_GROUP_TRANSPORT=ipc: HIP IPC handles through a
shared-memory block, peers' buffers pulled with copy / summing kernels, sequence words on the device) --
the transport a multi-GPU node can use as well.  Answers = the whole-range handle's = the oracle's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_group import make_data

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_world(tmp_path, native, world, sk, q, S, W, MS, exchange, cand_cap, transport="ipc", devices=1, words=None):
    """`world` fresh processes, one rank each, dealt over `devices` devices (tests/group_ipc_worker.py)"""
    env = dict(os.environ, NIQKI_TEST_TRANSPORT=transport, NIQKI_TEST_DEVICES=str(devices))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # (exported on this pool already: dmabuf IPC is what its driver supports)
    if transport == "ipc":
        env["NIQKI_GROUP_TRANSPORT"] = "ipc"
    else:
        env.pop("NIQKI_GROUP_TRANSPORT", None)
    if words:
        env["NIQKI_IPC_WORDS"] = words
    if transport == "ipc":
        gid = native.group_new_id()
    else:   # an ncclUniqueId: made without NIQKI_GROUP_TRANSPORT=ipc in the environment
        old = os.environ.pop("NIQKI_GROUP_TRANSPORT", None)
        try:
            gid = native.group_new_id()
        finally:
            if old is not None:
                os.environ["NIQKI_GROUP_TRANSPORT"] = old
    inp = tmp_path / "in.npz"
    np.savez(inp, sk=sk, q=q, S=S, W=W, min_score=MS)
    procs = []
    for r in range(world):
        out = tmp_path / ("out%d.npz" % r)
        procs.append((out, subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "group_ipc_worker.py"), str(r), str(world),
                                             bytes(gid).hex(), str(inp), str(out), exchange, str(cand_cap)],
                                            env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    res = []
    for out, p in procs:
        try:
            log, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for _, pp in procs:
                pp.kill()
            raise
        assert p.returncode == 0, log.decode(errors="replace")[-3000:]
        res.append(np.load(out))
    return res


def check_world(res, native, po, sk, q, S, W, MS, world, exchange, cand_cap):
    """every rank's hit lists = the whole-range handle's = the oracle's"""
    NQ = q.shape[0]
    whole = native.Engine(K=31, S=S, W=W, H=3, min_score_value=MS)
    whole.insert(sk)
    w_off, w_hc, w_hg = whole.query(q)
    p = po.make_params(31, S, W, 3, 0.0)
    p.min_score = MS
    ix = po.Index(p, sk)
    per = int(res[0]["per"])
    n_hits = 0
    for i in range(NQ):
        r, j = divmod(i, per)
        d = res[r]
        lo, hi = int(d["off"][j]), int(d["off"][j + 1])
        ehc, ehg = ix.query(q[i], min_score=MS)
        wl, wh = int(w_off[i]), int(w_off[i + 1])
        assert np.array_equal(d["hc"][lo:hi], ehc) and np.array_equal(d["hg"][lo:hi], ehg), (world, exchange, i)
        assert np.array_equal(d["hc"][lo:hi], w_hc[wl:wh]) and np.array_equal(d["hg"][lo:hi], w_hg[wl:wh])
        # the begin / end halves with device results, after the buffers were remapped
        assert np.array_equal(d["off2"], d["off"].astype(np.int64))
        assert np.array_equal(d["hc2"][lo:hi].astype(np.uint32), ehc) and np.array_equal(d["hg2"][lo:hi].astype(np.uint32), ehg)
        n_hits += hi - lo
    assert n_hits > 50
    # a capacity of 2 candidates overflows: those batches were redone densely, on every rank alike
    assert all((int(d["overflows"]) >= 1) == (cand_cap == 2) for d in res)
    whole.close()


# (words: where the sequence words a peer's running kernel polls live -- fine-grained device memory by default,
# the processes' shared block page-locked into every device where that cannot be made)
@pytest.mark.parametrize("world,exchange,cand_cap,words", [(2, "sparse", 256, None), (2, "dense", 256, "host"), (3, "sparse", 2, None),
                                                           (2, "sparse", 256, "coarse"), (5, "sparse", 256, None), (4, "dense", 256, None)])
def test_one_process_per_rank_on_one_gpu(tmp_path, native, po, world, exchange, cand_cap, words):
    S, W, N, NQ, MS = 9, 8, 1500, 23, 40
    sk, q = make_data(S, W, N, NQ, 31 + world)
    res = run_world(tmp_path, native, world, sk, q, S, W, MS, exchange, cand_cap, words=words)
    check_world(res, native, po, sk, q, S, W, MS, world, exchange, cand_cap)
    if words is None:    # the default: coherent by contract -- fine-grained device memory, or the host block where that failed
        assert all(int(d["words_kind"]) in (1, 2) for d in res)


def test_bench_two_ranks_on_one_gpu(tmp_path):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run), both ranks on the one GPU:
    gloo carries bench.py's own barrier, the library's ipc transport the exchange; the run checks its
    answers against a whole-range handle itself (--verify)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "2000", "--steps", "2", "--warmup", "1",
           "--batch", "256", "--no-cpu", "--no-extra", "--verify", "--devices", "1"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["value"] > 0
    assert j["config"]["transport"] == "ipc" and j["verify"]["hit_lists_equal_whole_range_handle"] is True
    assert j["config"]["transport_ranks_seen"] == j["n_gpus"] and j["config"]["exchange_gbs_per_link"] > 0


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher in the command and no WORLD_SIZE in the environment: the
    process starts two ranks of itself under torch.distributed.run as a child (before it touches the GPU),
    relays rank 0's one JSON line and exits with the child's code."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--genomes", "2000", "--steps", "2", "--warmup", "1",
           "--batch", "256", "--no-cpu", "--no-extra", "--verify", "--devices", "1"]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # ONE JSON line on stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["ranks_share_devices"] is True
    assert j["config"]["transport"] == "ipc" and j["verify"]["hit_lists_equal_whole_range_handle"] is True
    assert j["config"]["transport_ranks_seen"] == j["n_gpus"] and j["config"]["exchange_gbs_per_link"] > 0
