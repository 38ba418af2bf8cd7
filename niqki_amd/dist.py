"""Slot-range sharding of one index over the GPUs of a node (one process per
GPU, torch.distributed; backend "nccl" is RCCL over xGMI, "gloo" on CPU).

Rank r of G owns sketch slots [r*F/G, (r+1)*F/G) of EVERY indexed genome
(SURVEY.md 8e).  The hit count of a genome is a sum over slots, so a query step
has exactly one exchange of partial results:

  1. every rank sketches its share of the query batch        (no comm)
  2. all_gather of the int32 query sketches                   (F*4 B per query)
  3. gather-histogram on the local slot range for ALL queries (no comm)
  4. sum of the per-genome u16 hit vectors across ranks, scattered by query:
     reduce_scatter on the counters viewed as int32 pairs -- a count never
     exceeds F <= 2^15, so the two u16 halves of a word cannot carry
     (RCCL has no 16-bit integer type)
  5. every rank thresholds + orders the hits of its share of the queries

The engine argument is anything with the *_dev methods of niqki_amd.Engine (the
tests drive this module on CPU tensors over gloo with a stand-in engine).
"""
import torch
import torch.distributed as dist


def slot_range(rank, world, F):
    """Slots [begin, end) owned by `rank`; F = 2^S and world need not divide it."""
    return (F * rank) // world, (F * (rank + 1)) // world


def padded_batch(nq, world):
    """Queries per rank for a batch of nq (the batch is padded to world * this)."""
    return (nq + world - 1) // world


class ShardedQuery:
    def __init__(self, engine, n_genomes, F, device, group=None, exchange="reduce_scatter"):
        self.e = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.N = n_genomes
        self.F = F
        self.stride = (n_genomes + 1) & ~1
        self.device = device
        self.exchange = exchange
        self._bufs = {}

    def _buf(self, name, shape, dtype):
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = torch.zeros(shape, dtype=dtype, device=self.device)
            self._bufs[name] = b
        return b

    def exchange_sketches(self, local_sketches):
        """[per, F] int32 on every rank -> [world*per, F] (query order: rank major)."""
        per = local_sketches.shape[0]
        allsk = self._buf("allsk", (self.world * per, self.F), torch.int32)
        dist.all_gather_into_tensor(allsk, local_sketches.contiguous(), group=self.group)
        return allsk

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
            dist.reduce_scatter_tensor(out.view(-1), words.reshape(-1), group=self.group)
        return out.view(torch.int16)

    def step(self, local_sketches, hit_off, hit_counts, hit_gids, capacity):
        """One query batch.  local_sketches: [per, F] int32 of this rank's share.
        Fills hit_off[per+1] (int64), hit_counts / hit_gids (int32, capacity)."""
        per = local_sketches.shape[0]
        nq = per * self.world
        allsk = self.exchange_sketches(local_sketches)
        counts = self._buf("counts", (nq, self.stride), torch.int16)
        self.e.query_counts_dev(allsk, nq, counts, self.stride)
        red = self.reduce_counts(counts)
        self.e.hits_from_counts_dev(red, per, self.stride, 0, self.N, hit_off, hit_counts, hit_gids,
                                    capacity)
        return red
