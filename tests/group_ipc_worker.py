"""One rank of a multi-process slot-sharded group (tests/test_gpu_group_ipc.py and tests/test_gpu_multi_device.py
start `world` of these as fresh processes).  NIQKI_TEST_TRANSPORT = ipc (default) | rccl; NIQKI_TEST_DEVICES = number
of devices the ranks are dealt over (default 1: all ranks on device 0, which only the ipc transport allows).

    python tests/group_ipc_worker.py <rank> <world> <group id hex> <npz in> <npz out> <exchange> <cand_cap>

npz in: sk [N, F] int32, q [NQ, F] int32, S, W, min_score.  The rank inserts its rows of every batch through
niqki_group_insert, answers its share of the queries twice (the second batch through the begin / end halves)
and writes its hit lists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    gid = np.frombuffer(bytes.fromhex(sys.argv[3]), dtype=np.uint8).copy()
    d = np.load(sys.argv[4])
    out_path, exchange, cand_cap = sys.argv[5], sys.argv[6], int(sys.argv[7])
    transport = os.environ.get("NIQKI_TEST_TRANSPORT", "ipc")
    if transport == "ipc":
        os.environ["NIQKI_GROUP_TRANSPORT"] = "ipc"
    else:
        os.environ.pop("NIQKI_GROUP_TRANSPORT", None)
    import torch
    import niqki_amd
    di = rank % max(1, int(os.environ.get("NIQKI_TEST_DEVICES", "1")))
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    sk, q = d["sk"], d["q"]
    S, W, MS = int(d["S"]), int(d["W"]), int(d["min_score"])
    F = 1 << S
    b, e = niqki_amd.group_slot_range(rank, world, S)
    eng = niqki_amd.Engine(K=31, S=S, W=W, H=3, min_score_value=MS, slot_begin=b, slot_end=e, device=di)
    eng.set_stream(torch.cuda.current_stream().cuda_stream)
    grp = niqki_amd.Group([eng], first_rank=rank, world=world, group_id=gid)
    assert grp.stat("transport") == {"ipc": 2, "rccl": 1}[transport]
    assert grp.stat("ranks_seen") == world
    words_kind = grp.stat("ipc_words_kind") if transport == "ipc" else -1
    want = os.environ.get("NIQKI_IPC_WORDS")
    if transport == "ipc" and want in ("host", "coarse"):
        assert words_kind == {"host": 2, "coarse": 0}[want]
    grp.set_option("exchange", {"sparse": 1, "dense": 2}[exchange])
    grp.set_option("cand_cap", cand_cap)
    ins_per = 41
    for a in range(0, sk.shape[0], world * ins_per):
        blk = sk[a:a + world * ins_per]
        pad = np.full((world * ins_per, F), -1, np.int32)
        pad[:blk.shape[0]] = blk
        grp.insert_dev([torch.from_numpy(pad[rank * ins_per:(rank + 1) * ins_per].copy()).to(dev)], ins_per, blk.shape[0])
    assert eng.n_genomes == sk.shape[0]
    nq = q.shape[0]
    per = -(-nq // world)
    pad = np.full((world * per, F), -1, np.int32)
    pad[:nq] = q
    mine = torch.from_numpy(pad[rank * per:(rank + 1) * per].copy()).to(dev)
    off, hc, hg = grp.query([mine], per, capacity=8)[0]       # host results, forces the capacity retry
    # the same batch again, through the two halves with device results (and a bigger batch in between
    # so that the exchange buffers are reallocated and remapped)
    big = torch.from_numpy(np.tile(pad[rank * per:(rank + 1) * per], (3, 1)).copy()).to(dev)
    grp.query([big], 3 * per)
    cap = int(off[per]) + 16
    d_off = torch.zeros(per + 1, dtype=torch.int64, device=dev)
    d_hc = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_hg = torch.zeros(cap, dtype=torch.int32, device=dev)
    grp.query_begin_dev([mine], per, [d_off], [d_hc], [d_hg], cap)
    grp.query_end()
    eng.synchronize()
    np.savez(out_path, off=off, hc=hc, hg=hg, off2=d_off.cpu().numpy(), hc2=d_hc.cpu().numpy()[:int(off[per])],
             hg2=d_hg.cpu().numpy()[:int(off[per])], overflows=grp.stat("overflows"), per=per, words_kind=words_kind,
             arena_fine=grp.stat("ipc_arena_fine") if transport == "ipc" else -1, device=di)
    grp.close()
    eng.close()


if __name__ == "__main__":
    main()
