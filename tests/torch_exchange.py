"""The slot-shard exchange protocol of niqki_amd/csrc/nq_group.hip written with torch collectives
(gloo on CPU): what tests/test_dist_cpu.py runs with an oracle-backed stand-in engine, so that the
N > 1 logic -- slot slices, packed u16 reduce-scatter, sparse candidate exchange with its overflow
fallback, per-rank threshold -- is checked without GPUs.  Test infrastructure, not product code.

What the PRODUCT decides about a batch is not restated here: slot ranges come from niqki_group_slot_range and
the sparse / dense decision, the candidate threshold and the counter row stride from niqki_group_plan_batch --
the two pure functions of libniqki_hip.so that nq_group.hip itself calls (the library loads without a GPU)."""
import torch
import torch.distributed as dist

from niqki_amd.capi import group_plan, group_slot_range, row_stride


def slot_range(rank, world, F):
    """Slots [begin, end) owned by `rank` (niqki_group_slot_range); F = 2^S and world need not divide it."""
    S = F.bit_length() - 1
    assert 1 << S == F
    return group_slot_range(rank, world, S)


def padded_batch(nq, world):
    """Queries per rank for a batch of nq (the batch is padded to world * this)."""
    return (nq + world - 1) // world


class TorchExchange:
    """The protocol with torch collectives (gloo on CPU, or nccl): reference for the tests."""

    def __init__(self, engine, n_genomes, F, device, group=None, exchange="auto", min_score=None,
                 cand_cap=1024, compact_sketches=False):
        self.e = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.N = n_genomes
        self.F = F
        self.stride = row_stride(n_genomes)
        self.device = torch.device(device)
        self.min_score = engine.min_score if min_score is None else min_score
        self.cand_cap = cand_cap
        # sketch cells are -1 or a fingerprint below 2^W <= 2^15: they travel as int16 when the
        # caller says so (not after niqki_select_best_H, whose cells may exceed 2^W)
        self.compact = compact_sketches
        # the product's own decision for this group shape (option "exchange": 0 = choose, 1 = sparse, 2 = dense)
        S = F.bit_length() - 1
        opt = {"auto": 0, "sparse": 1}.get(exchange, 2)
        self.plan = group_plan(self.world, S, self.min_score, opt, 1, max(n_genomes, 1), cand_cap)
        assert self.plan.row_stride == self.stride
        if exchange == "sparse" and not self.plan.sparse:
            raise ValueError("the sparse exchange needs min_score >= number of shards")
        if exchange == "auto":
            exchange = "sparse" if self.plan.sparse else "reduce_scatter"
        self.exchange = exchange
        self.overflow = torch.zeros(1, dtype=torch.int32, device=device)  # sticky: a sparse step overflowed
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b

    # -- step 2
    def exchange_sketches(self, local_sketches):
        """[per, F] int32 on every rank -> [world*per, F] int32 (query order: rank
        major) whose columns of THIS rank's slot range are filled."""
        per = local_sketches.shape[0]
        G = self.world
        allsk = self._buf("allsk", (G * per, self.F), torch.int32)
        if self.F % G != 0:
            dist.all_gather_into_tensor(allsk, local_sketches.contiguous(), group=self.group)
            return allsk
        w = self.F // G
        wire = torch.int16 if self.compact else torch.int32
        send = local_sketches.view(per, G, w).permute(1, 0, 2).to(wire).contiguous()  # [dest][q][slot in dest's range]
        recv = self._buf("skrecv", (G, per, w), wire)                                  # [source][q][my slots]
        # pure data movement: as bytes (neither RCCL nor gloo has a 16-bit integer type)
        dist.all_to_all_single(recv.view(-1).view(torch.uint8), send.view(-1).view(torch.uint8), group=self.group)
        sb = self.rank * w
        allsk.view(G, per, self.F)[:, :, sb:sb + w] = recv
        return allsk

    # -- step 4, dense
    def reduce_counts(self, counts):
        """[world*per, stride] int16 partial counters -> [per, stride] int16 summed,
        rank r keeping queries [r*per, (r+1)*per)."""
        nq = counts.shape[0]
        per = nq // self.world
        words = counts.view(torch.int32)  # [nq, stride/2], packed u16 pairs
        out = self._buf("red", (per, self.stride // 2), torch.int32)
        if self.exchange == "all_to_all":
            # direct exchange over all links, then a local sum of the world partials
            recv = self._buf("a2a", (self.world, per, self.stride // 2), torch.int32)
            dist.all_to_all_single(recv.view(-1), words.reshape(-1), group=self.group)
            torch.sum(recv, dim=0, out=out)
        else:
            # (also the fallback of an overflowing sparse step)
            dist.reduce_scatter_tensor(out.view(-1), words.reshape(-1), group=self.group)
        return out.view(torch.int16)

    # -- step 4, sparse
    def reduce_candidates(self, counts):
        """Same result rows as reduce_counts for every genome that can reach
        min_score (all other entries of the returned rows are 0)."""
        G, C = self.world, self.cand_cap
        nq = counts.shape[0]
        per = nq // G
        thr = self.plan.cand_threshold   # ceil(min_score / G), as the product computes it
        cand = self._buf("cand", (nq, C), torch.int32)
        ncand = self._buf("ncand", (nq,), torch.int32)
        self.e.candidates_dev(counts, nq, self.stride, self.N, thr, C, cand, ncand)
        cand_all = self._buf("cand_all", (G, nq, C), torch.int32)
        ncand_all = self._buf("ncand_all", (G, nq), torch.int32)
        dist.all_gather_into_tensor(cand_all.view(-1), cand.view(-1), group=self.group)
        dist.all_gather_into_tensor(ncand_all.view(-1), ncand, group=self.group)
        self._step_overflow = (ncand_all > C).any().to(torch.int32)
        self.overflow |= self._step_overflow
        # union per query: the G lists side by side (duplicates are harmless)
        union = cand_all.permute(1, 0, 2).reshape(nq, G * C)                    # [q][G*C], -1 = no candidate
        valid = union >= 0
        idx = union.clamp(min=0).to(torch.int64)
        mine = torch.gather(counts.view(nq, self.stride), 1, idx).to(torch.int32) & 0xFFFF
        mine = torch.where(valid, mine, torch.zeros_like(mine))                  # this shard's partial counts
        total = self._buf("cand_tot", (per, G * C), torch.int32)
        dist.reduce_scatter_tensor(total.view(-1), mine.contiguous().view(-1), group=self.group)
        # dense rows of this rank's queries holding the candidates' summed counts
        red = self._buf("red16", (per, self.stride), torch.int16)
        red.zero_()
        own = slice(self.rank * per, (self.rank + 1) * per)
        vals = torch.where(valid[own], total, torch.zeros_like(total))
        # u16 bit pattern into int16 storage
        vals16 = torch.where(vals >= 32768, vals - 65536, vals).to(torch.int16)
        red.scatter_(1, idx[own], vals16)  # duplicates write the same value; invalid entries write 0 at id 0...
        # ...which must not clobber a real candidate at id 0: rewrite id 0 from the valid entries only
        is0 = valid[own] & (idx[own] == 0)
        v0 = torch.where(is0, vals, torch.zeros_like(vals)).amax(dim=1)
        red[:, 0] = torch.where(v0 >= 32768, v0 - 65536, v0).to(torch.int16)
        return red

    def step(self, local_sketches, hit_off, hit_counts, hit_gids, capacity, check_overflow=True):
        """One query batch.  local_sketches: [per, F] int32 of this rank's share.
        Fills hit_off[per+1] (int64), hit_counts / hit_gids (int32, capacity).

        The engine's kernels and the torch ops / collectives of this module must run on ONE
        stream: the engine is (re)bound to torch's current stream here (a handle's own stream is
        non-blocking and orders against nothing else).  A sparse step whose candidate lists
        overflow is redone with the dense exchange (one 4-byte read-back per step;
        check_overflow=False leaves the check to the caller, who then must test `overflow`)."""
        if self.device.type == "cuda" and hasattr(self.e, "set_stream"):
            self.e.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        per = local_sketches.shape[0]
        nq = per * self.world
        allsk = self.exchange_sketches(local_sketches)
        counts = self._buf("counts", (nq, self.stride), torch.int16)
        self.e.query_counts_dev(allsk, nq, counts, self.stride)
        if self.exchange == "sparse":
            red = self.reduce_candidates(counts)
            if check_overflow:
                # every rank sees the same all-gathered list sizes, so all ranks take the same branch
                if int(self._step_overflow.item()):
                    red = self.reduce_counts(counts)
        else:
            red = self.reduce_counts(counts)
        self.e.hits_from_counts_dev(red, per, self.stride, 0, self.N, hit_off, hit_counts, hit_gids,
                                    capacity)
        return red
