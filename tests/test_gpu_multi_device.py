"""GPU, gated on the number of devices: the slot-sharded group with every rank on a device of ITS OWN --
north_star's "the S x 2^W inverted index shards by sketch-slot range across the 8 GPUs of one node"
(src/niqki_index.cpp:523-540 is the loop that is parallelised, :633-687 the per-query work that is cut by slots).

On the one-GPU boxes of this pool every test here is collected and skipped; on a node with >= 2 (>= 8) devices
they run world 2 (8) as one process per rank over RCCL and over the library's ipc transport (peers' buffers
mapped over xGMI, sequence words in fine-grained memory), the `niqki` program with --gpus N on distinct devices,
and bench.py --gpus N started without a launcher -- each against a whole-range handle and the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_group import make_data
from test_gpu_group_ipc import check_world, run_world

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def n_devices():
    import torch
    return torch.cuda.device_count()   # (does not initialise the GPU in this process)


def need(n):
    have = n_devices()
    if have < n:
        pytest.skip("needs %d devices, this box has %d" % (n, have))


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("transport", ["rccl", "ipc"])
@pytest.mark.parametrize("exchange,cand_cap", [("sparse", 256), ("dense", 256), ("sparse", 2)])
def test_ranks_on_distinct_devices(tmp_path, native, po, world, transport, exchange, cand_cap):
    """world ranks, rank r on device r: hit lists = whole-range handle = oracle; a candidate capacity of 2 overflows
    and the batch is redone densely on every rank alike."""
    need(world)
    S, W, N, NQ, MS = 9, 8, 1500, 4 * world + 3, 40
    sk, q = make_data(S, W, N, NQ, 77 + world)
    res = run_world(tmp_path, native, world, sk, q, S, W, MS, exchange, cand_cap, transport=transport, devices=world)
    assert sorted(int(d["device"]) for d in res) == list(range(world))
    check_world(res, native, po, sk, q, S, W, MS, world, exchange, cand_cap)
    if transport == "ipc":   # the words a peer GPU's running kernel polls: coherent by contract
        assert all(int(d["words_kind"]) in (1, 2) for d in res)


@pytest.mark.parametrize("words", ["host", "coarse"])
def test_ipc_word_kinds_across_devices(tmp_path, native, po, words):
    """the other homes of the sequence words (the shared block page-locked into every device; plain device memory)"""
    need(2)
    S, W, N, NQ, MS = 9, 8, 1500, 11, 40
    sk, q = make_data(S, W, N, NQ, 91)
    res = run_world(tmp_path, native, 2, sk, q, S, W, MS, "sparse", 256, transport="ipc", devices=2, words=words)
    check_world(res, native, po, sk, q, S, W, MS, 2, "sparse", 256)


@pytest.mark.parametrize("gpus", [2, 8])
@pytest.mark.parametrize("transport", [None, "ipc"])
def test_bench_on_distinct_devices(tmp_path, gpus, transport):
    """`python bench.py --gpus N`, no launcher: N ranks on N devices (RCCL unless --transport ipc), --verify against a
    whole-range handle on every rank"""
    need(gpus)
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--genomes", "4000", "--steps", "3", "--warmup", "1",
           "--batch", "512", "--no-cpu", "--no-extra", "--verify"]
    if transport:
        cmd += ["--transport", transport]
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="replace")[-3000:]
    j = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == gpus and j["config"]["ranks_share_devices"] is False
    assert j["config"]["transport"] == (transport or "rccl"), j["config"].get("transport_note")
    assert j["verify"]["hit_lists_equal_whole_range_handle"] is True


@pytest.mark.parametrize("gpus", [2, 8])
def test_host_program_on_distinct_devices(tmp_path_factory, native, gold, gpus):
    """`niqki --gpus N` without NIQKI_SHARDS_ON_ONE_DEVICE: shard r on device r, RCCL inside one process; the
    reference CLI's hits, matrix and dump bytes (tests/golden/reference_meta.json)."""
    need(gpus)
    import hashlib
    from conftest import make_cli_workdir
    from test_cli_gpu import BIN, assert_same_text, gunzip
    _, meta = gold
    wd = make_cli_workdir(tmp_path_factory.mktemp("cli_md"), native, meta)
    env = {k: v for k, v in os.environ.items() if k != "NIQKI_SHARDS_ON_ONE_DEVICE"}

    def run(args):
        r = subprocess.run([BIN, "--gpus", str(gpus)] + args, cwd=wd, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
    run(["-I", "fof.txt", "-Q", "fof.txt", "-S", "10", "-J", "0.1", "-O", "md_hits.gz", "-D", "md.dump"])
    assert_same_text(gunzip(wd / "md_hits.gz").decode(), meta["cli"]["hits"])
    raw = gunzip(wd / "md.dump")
    assert len(raw) == meta["cli"]["dump_len"] and hashlib.md5(raw).hexdigest() == meta["cli"]["dump_md5"]
    run(["-L", "md.dump", "-Q", "fof.txt", "-O", "md_loaded.gz"])
    assert_same_text(gunzip(wd / "md_loaded.gz").decode(), meta["cli"]["hits_loaded"])
    run(["-M", "fof.txt", "-S", "10", "-O", "md_matrix.gz"])
    assert_same_text(gunzip(wd / "md_matrix.gz").decode(), meta["cli"]["matrix"])


def test_worker_over_rccl_at_world_1(tmp_path, native, po):
    """the rccl form of the multi-process harness above (id made without NIQKI_GROUP_TRANSPORT, worker's transport
    switch, RCCL communicator, collectives, begin / end halves) with the one rank a one-GPU box can give it"""
    S, W, N, NQ, MS = 9, 8, 700, 9, 40
    sk, q = make_data(S, W, N, NQ, 5)
    res = run_world(tmp_path, native, 1, sk, q, S, W, MS, "sparse", 256, transport="rccl", devices=1)
    check_world(res, native, po, sk, q, S, W, MS, 1, "sparse", 256)


def test_gated_tests_are_collected_here():
    """the box this runs on says how many of the above really ran"""
    n = n_devices()
    assert n >= 1
    print("devices visible: %d (world-2 tests %s, world-8 tests %s)" % (n, "run" if n >= 2 else "skipped", "run" if n >= 8 else "skipped"))
