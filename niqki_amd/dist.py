"""Slot-range sharding of one index over the GPUs of a node, one process per GPU
(torch.distributed.run).

The exchange itself lives behind the C ABI (niqki_group_*, niqki_amd/csrc/nq_group.hip): RCCL
all-to-all of int16 sketch slices, slot-shard gather, sparse candidate exchange or dense
reduce-scatter of packed u16 hit vectors, per-rank threshold.  On a GPU, ShardedQuery is a thin
caller of it: torch.distributed only carries the 128-byte group id from rank 0 to the others.

torch.distributed's backend may be gloo (ranks that share a device, NIQKI_GROUP_TRANSPORT=ipc: the id
then travels as a CPU tensor).  The same protocol written with torch collectives lives with the tests
(tests/torch_exchange.py): the CPU suite runs it over gloo with a stand-in engine.

Rank r of G owns sketch slots [r*F/G, (r+1)*F/G) of EVERY indexed genome.  The hit
count of a genome is a sum over slots, so a query step has exactly one exchange of partial results:

  1. every rank sketches its share of the query batch                 (no comm)
  2. sketch exchange: each rank needs only ITS slot range of every query, so the
     sketches go through one all_to_all of F/G-slot slices (2F/G bytes per
     query and peer as int16) instead of an all_gather of whole sketches
  3. gather-histogram on the local slot range for ALL queries          (no comm)
  4. cross-shard sum of the per-genome hit vectors, scattered by query:
       "dense"   reduce_scatter of the u16 counters viewed as int32 pairs -- a
                 count never exceeds F <= 2^15, so the two halves of a word
                 cannot carry (RCCL has no 16-bit integer type); 2N bytes per
                 query per rank
       "sparse"  only candidates travel: a genome whose summed count reaches
                 min_score has a partial count >= ceil(min_score/G) on some
                 rank, so ranks all_gather those (few) candidate ids, look their
                 own partial counts up for the union, and reduce_scatter just
                 these values; exact, a few KB per query.  Needs
                 min_score >= 4*G; a step whose candidate lists overflow their
                 capacity is redone densely.
  5. every rank thresholds + orders the hits of its share of the queries
"""
import torch
import torch.distributed as dist


def slot_range(rank, world, F):
    """Slots [begin, end) owned by `rank` (F = 2^S): niqki_group_slot_range's cut, floor(F * r / G).  Stated here in
    Python so that the gloo / CPU protocol (tests/torch_exchange.py) also runs on a host without the built library
    and the HIP runtime; tests/test_dist_cpu.py asserts that this is the library's cut wherever the library loads."""
    return (F * rank) // world, (F * (rank + 1)) // world


def padded_batch(nq, world):
    """Queries per rank for a batch of nq (the batch is padded to world * this)."""
    return (nq + world - 1) // world


class ShardedQuery:
    """One rank's end of a slot-sharded query: the C ABI group (RCCL or the ipc transport inside
    libniqki_hip.so)."""

    def __init__(self, engine, n_genomes, F, device, group=None, exchange="auto", min_score=None,
                 cand_cap=1024, compact_sketches=True):
        import numpy as np
        from . import capi
        self.e = engine
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = torch.device(device)
        on_cpu = dist.get_backend(group) == "gloo"
        ident = torch.zeros(capi.GROUP_ID_BYTES, dtype=torch.uint8, device="cpu" if on_cpu else self.device)
        if self.rank == 0:
            ident.copy_(torch.from_numpy(capi.group_new_id()))
        dist.broadcast(ident, src=0, group=group)           # the only torch collective on this path
        self.g = capi.Group([engine], first_rank=self.rank, world=self.world, group_id=ident.cpu().numpy().astype(np.uint8))
        self.g.set_option("exchange", {"auto": 0, "sparse": 1, "reduce_scatter": 2, "dense": 2}[exchange])
        self.g.set_option("cand_cap", cand_cap)
        self.cand_cap = cand_cap
        self.exchange = "sparse" if self.g.stat("sparse") else "reduce_scatter"
        self.transport = ("local", "rccl", "ipc")[self.g.stat("transport")]

    @property
    def ranks_seen(self):
        """Ranks the transport itself knows of (RCCL: ncclCommCount; ipc: peers mapped): world when all is well."""
        return self.g.stat("ranks_seen")

    @property
    def overflows(self):
        return self.g.stat("overflows")

    def step(self, local_sketches, hit_off, hit_counts, hit_gids, capacity):
        """One query batch.  local_sketches: [per, F] int32 of this rank's share (device).
        Fills hit_off[per+1] (int64), hit_counts / hit_gids (int32, capacity).  The engine's
        stream is (re)bound to torch's current stream so that the library's kernels and RCCL
        calls order against the caller's torch work."""
        self.e.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.g.query_dev([local_sketches], local_sketches.shape[0], [hit_off], [hit_counts], [hit_gids], capacity)

    def begin(self, local_sketches, hit_off, hit_counts, hit_gids, capacity):
        """niqki_group_query_begin: the batch is enqueued, the host does not wait (end() does)."""
        self.e.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.g.query_begin_dev([local_sketches], local_sketches.shape[0], [hit_off], [hit_counts], [hit_gids], capacity)

    def end(self):
        self.g.query_end()

    def insert(self, local_sketches, n_total):
        """Batch of world * per sketches, this rank holding rows [rank*per, (rank+1)*per); the
        first n_total rows (rank major) are inserted."""
        self.e.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.g.insert_dev([local_sketches], local_sketches.shape[0], n_total)

    def close(self):
        self.g.close()
