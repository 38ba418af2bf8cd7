// nq_group.hip -- one index sharded by sketch-slot range over several GPUs (include/niqki_hip.h,
// niqki_group_*).  Rank r of G owns slots [F*r/G, F*(r+1)/G) of EVERY genome; the hit count of a
// genome is a sum over slots (src/niqki_index.cpp:652-661 runs over all of them), so a query batch
// needs exactly one exchange of partial results:
//
//   1. every rank has sketched its share of the batch (`per` queries)            no communication
//   2. slice exchange: a rank needs only ITS slots of every query -> one all-to-all of int16
//      slices (2*F/G bytes per query and peer)                                    slice_pack / unpack
//   3. gather-histogram over the local slots for ALL G*per queries                gather_kernel
//   4. cross-shard sum of the per-genome hit vectors, scattered by query:
//        dense   reduce-scatter of the u16 counters as packed pairs in u32 words -- a count never
//                exceeds F <= 2^15, so the halves cannot carry (RCCL has no 16-bit integer type)
//        sparse  a genome whose summed count reaches min_score has a partial count of at least
//                ceil(min_score / G) on some rank: ranks all-gather those candidate ids
//                (candidates_kernel), look their own partial counts up for the union
//                (cand_lookup_kernel), reduce-scatter just these values and scatter the sums into
//                otherwise empty counter rows (cand_scatter_kernel).  Exact; a candidate list that
//                overflows its capacity makes every rank redo the step densely.
//   5. every rank thresholds + orders the hits of its own `per` queries          hits_* kernels
//
// Transport: RCCL (librccl, loaded on first use) -- one communicator per local rank, so the same
// code serves one rank per process (bench.py under torch.distributed.run) and all ranks in one
// process (`niqki --gpus N`).  When two shards of a single-process group share a device (tests,
// emulation of G shards on one GPU), which RCCL refuses, plain device-to-device copies and a
// summing kernel stand in for the collectives.
#include "nq_handle.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace {

using nqi::Buf;

// ---- librccl, resolved at first use ---------------------------------------------------------
struct Rccl {
  void *lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string why;
  bool load() {
    if (lib) return true;
    for (const char *name : {"librccl.so.1", "librccl.so"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (lib) break;
    }
    if (!lib) { why = std::string("cannot load librccl: ") + dlerror(); return false; }
#define NQ_SYM(field, sym)                                                  \
  field = (decltype(field))dlsym(lib, sym);                                 \
  if (!field) { why = std::string("librccl lacks ") + sym; lib = nullptr; return false; }
    NQ_SYM(GetUniqueId, "ncclGetUniqueId")
    NQ_SYM(CommInitRank, "ncclCommInitRank")
    NQ_SYM(CommDestroy, "ncclCommDestroy")
    NQ_SYM(GroupStart, "ncclGroupStart")
    NQ_SYM(GroupEnd, "ncclGroupEnd")
    NQ_SYM(Send, "ncclSend")
    NQ_SYM(Recv, "ncclRecv")
    NQ_SYM(AllGather, "ncclAllGather")
    NQ_SYM(ReduceScatter, "ncclReduceScatter")
    NQ_SYM(GetErrorString, "ncclGetErrorString")
#undef NQ_SYM
    return true;
  }
};
Rccl &rccl() {
  static Rccl r;
  return r;
}

constexpr uint32_t kMaxWorld = 64;

}  // namespace

namespace nq {

// ---- exchange kernels ----------------------------------------------------------------------
// first slot of rank r (niqki_group_slot_range)
__host__ __device__ inline uint32_t cut(uint32_t F, uint32_t r, uint32_t G) { return (uint32_t)(((uint64_t)F * r) / G); }

// [per][F] int32 sketches -> [G][per][w_max] int16 slices (slot slice of destination g; cells that
// are empty or outside [0, R) -- never indexed or queried, src/niqki_index.cpp:364,:654 -- travel as -1)
__global__ __launch_bounds__(256) void slice_pack_kernel(const int32_t *sk, uint32_t per, uint32_t F, uint32_t R, uint32_t G,
                                                        uint32_t w_max, int16_t *out) {
  const uint32_t q = blockIdx.x, g = blockIdx.y;
  const uint32_t b = cut(F, g, G), w = cut(F, g + 1, G) - b;
  const int32_t *row = sk + (uint64_t)q * F + b;
  int16_t *dst = out + ((uint64_t)g * per + q) * w_max;
  for (uint32_t j = threadIdx.x; j < w_max; j += 256) {
    int32_t x = j < w ? row[j] : -1;
    dst[j] = (x >= 0 && (uint32_t)x < R) ? (int16_t)x : (int16_t)-1;
  }
}

// [nq][w_max] int16 slices of MY slots -> [nq][f_local] int32 sketch rows for the query kernels
__global__ __launch_bounds__(256) void slice_unpack_kernel(const int16_t *in, uint32_t w_max, uint32_t f_local, int32_t *out) {
  const uint32_t q = blockIdx.x;
  for (uint32_t j = threadIdx.x; j < f_local; j += 256) out[(uint64_t)q * f_local + j] = in[(uint64_t)q * w_max + j];
}

// A rank's candidates travel as one blob: nq lists of C ids, then the nq list sizes (one all-gather).
__host__ __device__ inline uint64_t cand_blob_ints(uint32_t nq, uint32_t C) { return (uint64_t)nq * C + nq; }

// mine[q][g*C + c] = this shard's partial count of candidate c of rank g for query q (0 where the
// list has no entry), as u16: the cross-shard sums stay <= F <= 2^15, so the ranks add them as packed
// pairs in u32 words without a carry (half the bytes of the reduce-scatter); flag |= some list
// overflowed its capacity
__global__ __launch_bounds__(256) void cand_lookup_kernel(const uint16_t *counts, uint64_t stride, uint32_t nq, uint32_t G,
                                                         uint32_t C, const int32_t *cand_all, uint16_t *mine, uint32_t *flag) {
  const uint32_t q = blockIdx.x;
  const uint16_t *row = counts + (uint64_t)q * stride;
  const uint64_t blob = cand_blob_ints(nq, C);
  for (uint32_t i = threadIdx.x; i < G * C; i += 256) {
    const uint32_t g = i / C, c = i % C;
    const int32_t id = cand_all[g * blob + (uint64_t)q * C + c];
    mine[(uint64_t)q * G * C + i] = id >= 0 ? row[id] : (uint16_t)0;
  }
  if (threadIdx.x < G && (uint32_t)cand_all[threadIdx.x * blob + (uint64_t)nq * C + q] > C) atomicOr(flag, 1u);
}

// summed candidate counts into the (zeroed) counter rows of this rank's own queries
__global__ __launch_bounds__(256) void cand_scatter_kernel(const uint16_t *tot, uint32_t per, uint32_t first_q, uint32_t nq,
                                                          uint32_t G, uint32_t C, const int32_t *cand_all, uint16_t *red,
                                                          uint64_t stride) {
  const uint32_t ql = blockIdx.x, q = first_q + ql;
  const uint64_t blob = cand_blob_ints(nq, C);
  for (uint32_t i = threadIdx.x; i < G * C; i += 256) {
    const uint32_t g = i / C, c = i % C;
    const int32_t id = cand_all[g * blob + (uint64_t)q * C + c];
    if (id >= 0) red[(uint64_t)ql * stride + id] = tot[(uint64_t)ql * G * C + i];  // duplicates write the same sum
  }
}

struct SumSrc { const uint32_t *p[kMaxWorld]; };
// stand-in for reduce-scatter inside one process: out[i] = sum over ranks of src[r][off + i]
__global__ __launch_bounds__(256) void sum_rows_kernel(SumSrc src, uint32_t G, uint64_t off, uint64_t n, uint32_t *out) {
  const uint64_t step = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    uint32_t s = 0;
    for (uint32_t r = 0; r < G; ++r) s += src.p[r][off + i];
    out[i] = s;
  }
}

}  // namespace nq

struct niqki_group {
  uint32_t world = 1, n_local = 1, first = 0;
  std::vector<niqki_index *> sh;     // local shards, rank first + l
  bool use_rccl = false;
  std::vector<ncclComm_t> comm;      // per local rank (use_rccl)
  int exchange = 0;                  // 0 = choose, 1 = sparse, 2 = dense reduce-scatter
  uint32_t cand_cap = 256;
  uint64_t overflows = 0;            // sparse steps redone densely
  std::string err;
  struct Ws {
    Buf send, recv, allsk, counts, cand, cand_all, mine, tot, red, flag, hitoff, hc, hg, stpad;
    hipEvent_t ev = nullptr;
  };
  std::vector<Ws> ws;
};

namespace {

int gfail(niqki_group *g, int code, const std::string &msg) {
  if (g) g->err = msg;
  return code;
}

#define NQ_G(g, l, call)                                                                      \
  do {                                                                                        \
    int rc_ = (call);                                                                         \
    if (rc_) return gfail(g, rc_, std::string(#call) + ": " + niqki_last_error((g)->sh[l]));  \
  } while (0)
#define NQ_GH(g, call)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (call);                                                                   \
    if (e_ != hipSuccess) return gfail(g, NIQKI_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define NQ_GN(g, call)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (call);                                                                 \
    if (r_ != ncclSuccess) return gfail(g, NIQKI_E_HIP, std::string(#call) + ": " + rccl().GetErrorString(r_)); \
  } while (0)

// every local stream waits for everything enqueued so far on all local streams (local transport)
int cross_wait(niqki_group *g) {
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    NQ_GH(g, hipEventRecord(g->ws[l].ev, g->sh[l]->stream));
  }
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    for (uint32_t s = 0; s < g->n_local; ++s)
      if (s != l) NQ_GH(g, hipStreamWaitEvent(g->sh[l]->stream, g->ws[s].ev, 0));
  }
  return NIQKI_OK;
}

// recv[l] = [world][bytes] <- send[s] + l * bytes of every rank s
int all_to_all(niqki_group *g, Buf niqki_group::Ws::*send, Buf niqki_group::Ws::*recv, size_t bytes) {
  if (g->use_rccl) {
    NQ_GN(g, rccl().GroupStart());
    for (uint32_t l = 0; l < g->n_local; ++l) {
      const char *s = (const char *)(g->ws[l].*send).p;
      char *r = (char *)(g->ws[l].*recv).p;
      for (uint32_t p = 0; p < g->world; ++p) {
        NQ_GN(g, rccl().Send(s + (size_t)p * bytes, bytes, ncclUint8, (int)p, g->comm[l], g->sh[l]->stream));
        NQ_GN(g, rccl().Recv(r + (size_t)p * bytes, bytes, ncclUint8, (int)p, g->comm[l], g->sh[l]->stream));
      }
    }
    NQ_GN(g, rccl().GroupEnd());
    return NIQKI_OK;
  }
  int rc = cross_wait(g);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    for (uint32_t s = 0; s < g->n_local; ++s)
      NQ_GH(g, hipMemcpyAsync((char *)(g->ws[l].*recv).p + (size_t)s * bytes, (const char *)(g->ws[s].*send).p + (size_t)l * bytes,
                              bytes, hipMemcpyDeviceToDevice, g->sh[l]->stream));
  }
  return cross_wait(g);
}

// recv[l] = [world][bytes] <- send[s] of every rank s
int all_gather(niqki_group *g, Buf niqki_group::Ws::*send, Buf niqki_group::Ws::*recv, size_t bytes) {
  if (g->use_rccl) {
    NQ_GN(g, rccl().GroupStart());
    for (uint32_t l = 0; l < g->n_local; ++l)
      NQ_GN(g, rccl().AllGather((g->ws[l].*send).p, (g->ws[l].*recv).p, bytes, ncclUint8, g->comm[l], g->sh[l]->stream));
    NQ_GN(g, rccl().GroupEnd());
    return NIQKI_OK;
  }
  int rc = cross_wait(g);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    for (uint32_t s = 0; s < g->n_local; ++s)
      NQ_GH(g, hipMemcpyAsync((char *)(g->ws[l].*recv).p + (size_t)s * bytes, (g->ws[s].*send).p, bytes,
                              hipMemcpyDeviceToDevice, g->sh[l]->stream));
  }
  return cross_wait(g);
}

// recv[l][i] = sum over ranks s of send[s][rank(l) * count + i], u32 words
int reduce_scatter_u32(niqki_group *g, Buf niqki_group::Ws::*send, Buf niqki_group::Ws::*recv, size_t count) {
  if (g->use_rccl) {
    NQ_GN(g, rccl().GroupStart());
    for (uint32_t l = 0; l < g->n_local; ++l)
      NQ_GN(g, rccl().ReduceScatter((g->ws[l].*send).p, (g->ws[l].*recv).p, count, ncclUint32, ncclSum, g->comm[l], g->sh[l]->stream));
    NQ_GN(g, rccl().GroupEnd());
    return NIQKI_OK;
  }
  int rc = cross_wait(g);
  if (rc) return rc;
  nq::SumSrc src{};
  for (uint32_t s = 0; s < g->n_local; ++s) src.p[s] = (const uint32_t *)(g->ws[s].*send).p;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    if (count == 0) continue;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((count + 255) / 256, 8192);
    hipLaunchKernelGGL(nq::sum_rows_kernel, dim3(blocks), dim3(256), 0, g->sh[l]->stream, src, g->world, (uint64_t)l * count,
                       (uint64_t)count, (uint32_t *)(g->ws[l].*recv).p);
    NQ_GH(g, hipGetLastError());
  }
  return cross_wait(g);
}

// steps 1-3 shared by insert and query: local sketches -> compact rows of all world*per sketches
// restricted to each rank's slots (ws.allsk, stride f_local)
int exchange_slices(niqki_group *g, const int32_t *const *local_sketches, uint32_t per) {
  const uint32_t G = g->world, F = g->sh[0]->d.F, R = g->sh[0]->d.R;
  const uint32_t w_max = (F + G - 1) / G;
  const size_t bytes = (size_t)per * w_max * 2;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].send, bytes * G));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].recv, bytes * G));
    nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
    hipLaunchKernelGGL(nq::slice_pack_kernel, dim3(per, G), dim3(256), 0, ix->stream, local_sketches[l], per, F, R, G, w_max,
                       (int16_t *)g->ws[l].send.p);
    NQ_GH(g, hipGetLastError());
  }
  int rc = all_to_all(g, &niqki_group::Ws::send, &niqki_group::Ws::recv, bytes);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
    NQ_GH(g, hipSetDevice(ix->device));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].allsk, (size_t)G * per * f_local * 4));
    nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
    hipLaunchKernelGGL(nq::slice_unpack_kernel, dim3(G * per), dim3(256), 0, ix->stream, (const int16_t *)g->ws[l].recv.p, w_max,
                       f_local, (int32_t *)g->ws[l].allsk.p);
    NQ_GH(g, hipGetLastError());
  }
  return NIQKI_OK;
}

}  // namespace

extern "C" {

void niqki_group_slot_range(uint32_t rank, uint32_t world, uint32_t S, uint32_t *slot_begin, uint32_t *slot_end) {
  const uint32_t F = 1u << S;
  if (slot_begin) *slot_begin = nq::cut(F, rank, world);
  if (slot_end) *slot_end = nq::cut(F, rank + 1, world);
}

int niqki_group_new_id(uint8_t id[NIQKI_GROUP_ID_BYTES]) {
  if (!id) return NIQKI_E_INVALID;
  if (!rccl().load()) return NIQKI_E_STATE;
  static_assert(NIQKI_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "group id is an ncclUniqueId");
  ncclUniqueId u;
  if (rccl().GetUniqueId(&u) != ncclSuccess) return NIQKI_E_HIP;
  std::memcpy(id, &u, NIQKI_GROUP_ID_BYTES);
  return NIQKI_OK;
}

int niqki_group_create(niqki_index *const *shards, uint32_t n_local, uint32_t first_rank, uint32_t world, const uint8_t *id,
                       niqki_group **out) {
  if (!shards || !out || n_local == 0 || world == 0 || world > kMaxWorld || first_rank + n_local > world) return NIQKI_E_INVALID;
  *out = nullptr;
  niqki_group *g = new (std::nothrow) niqki_group();
  if (!g) return NIQKI_E_NOMEM;
  g->world = world; g->n_local = n_local; g->first = first_rank;
  g->sh.assign(shards, shards + n_local);
  g->ws.resize(n_local);
  auto bail = [&](int code, const std::string &why) {
    if (g->sh[0]) g->sh[0]->err = why;   // readable through niqki_last_error(shards[0])
    niqki_group_destroy(g);
    return code;
  };
  const uint32_t S = shards[0] ? shards[0]->d.S : 0;
  bool shared_device = false;
  for (uint32_t l = 0; l < n_local; ++l) {
    niqki_index *ix = shards[l];
    if (!ix) { delete g; return NIQKI_E_INVALID; }
    uint32_t b, e;
    niqki_group_slot_range(first_rank + l, world, S, &b, &e);
    if (ix->d.S != S || ix->d.slot_begin != b || ix->d.slot_end != e)
      return bail(NIQKI_E_INVALID, "shard " + std::to_string(first_rank + l) + " must own slots [" + std::to_string(b) + ", " +
                                       std::to_string(e) + ") (niqki_group_slot_range)");
    if (ix->resident_bytes) return bail(NIQKI_E_INVALID, "a paged index (resident_bytes) cannot be a shard of a group");
    if (ix->d.S > 15) return bail(NIQKI_E_INVALID, "groups need S <= 15 (the exchange sums u16 counters)");
    if (ix->d.K != shards[0]->d.K || ix->d.W != shards[0]->d.W || ix->d.min_score != shards[0]->d.min_score ||
        ix->n_genomes != shards[0]->n_genomes)
      return bail(NIQKI_E_INVALID, "the shards of a group must agree in K, W, min_score and genome count");
    for (uint32_t m = 0; m < l; ++m) shared_device |= shards[m]->device == ix->device;
  }
  if (const char *t = std::getenv("NIQKI_GROUP_TRANSPORT")) shared_device |= !std::strcmp(t, "local");
  g->use_rccl = !(n_local == world && shared_device);
  if (shared_device && n_local != world) return bail(NIQKI_E_INVALID, "shards that share a device need all ranks in one process");
  for (uint32_t l = 0; l < n_local; ++l) {
    if (hipSetDevice(shards[l]->device) != hipSuccess || hipEventCreateWithFlags(&g->ws[l].ev, hipEventDisableTiming) != hipSuccess)
      return bail(NIQKI_E_HIP, "hipEventCreate failed");
  }
  if (g->use_rccl) {
    if (!rccl().load()) return bail(NIQKI_E_STATE, rccl().why);
    ncclUniqueId u;
    if (id) std::memcpy(&u, id, NIQKI_GROUP_ID_BYTES);
    else if (n_local != world) return bail(NIQKI_E_INVALID, "a group spanning several processes needs the id of niqki_group_new_id");
    else if (rccl().GetUniqueId(&u) != ncclSuccess) return bail(NIQKI_E_HIP, "ncclGetUniqueId failed");
    g->comm.assign(n_local, nullptr);
    ncclResult_t r = rccl().GroupStart();
    for (uint32_t l = 0; l < n_local && r == ncclSuccess; ++l) {
      if (hipSetDevice(shards[l]->device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
      r = rccl().CommInitRank(&g->comm[l], (int)world, u, (int)(first_rank + l));
    }
    const ncclResult_t r2 = rccl().GroupEnd();
    if (r != ncclSuccess || r2 != ncclSuccess)
      return bail(NIQKI_E_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r != ncclSuccess ? r : r2));
  }
  *out = g;
  return NIQKI_OK;
}

void niqki_group_destroy(niqki_group *g) {
  if (!g) return;
  for (uint32_t l = 0; l < g->n_local && l < g->ws.size(); ++l) {
    if (g->sh[l]) {
      (void)hipSetDevice(g->sh[l]->device);
      (void)hipStreamSynchronize(g->sh[l]->stream);
    }
    auto &w = g->ws[l];
    for (Buf *b : {&w.send, &w.recv, &w.allsk, &w.counts, &w.cand, &w.cand_all, &w.mine, &w.tot, &w.red,
                   &w.flag, &w.hitoff, &w.hc, &w.hg, &w.stpad})
      if (b->p) (void)hipFree(b->p);
    if (w.ev) (void)hipEventDestroy(w.ev);
    if (l < g->comm.size() && g->comm[l]) (void)rccl().CommDestroy(g->comm[l]);
  }
  delete g;
}

const char *niqki_group_last_error(const niqki_group *g) { return g ? g->err.c_str() : ""; }

int niqki_group_set_option(niqki_group *g, const char *key, int64_t value) {
  if (!g || !key) return NIQKI_E_INVALID;
  if (!std::strcmp(key, "exchange")) {
    if (value < 0 || value > 2) return gfail(g, NIQKI_E_INVALID, "exchange: 0 = choose, 1 = sparse, 2 = dense");
    g->exchange = (int)value;
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "cand_cap")) {
    if (value < 2 || value > 65536 || (value & 1)) return gfail(g, NIQKI_E_INVALID, "cand_cap must be even, in 2..65536");
    g->cand_cap = (uint32_t)value;
    return NIQKI_OK;
  }
  return gfail(g, NIQKI_E_INVALID, std::string("unknown group option ") + key);
}

int niqki_group_get_stat(const niqki_group *g, const char *key, uint64_t *value) {
  if (!g || !key || !value) return NIQKI_E_INVALID;
  if (!std::strcmp(key, "overflows")) { *value = g->overflows; return NIQKI_OK; }
  if (!std::strcmp(key, "rccl")) { *value = g->use_rccl ? 1 : 0; return NIQKI_OK; }
  if (!std::strcmp(key, "sparse")) {
    const uint32_t ms = g->sh[0]->d.min_score;
    *value = (g->exchange == 1 || (g->exchange == 0 && ms >= 4 * g->world)) ? 1 : 0;
    return NIQKI_OK;
  }
  return NIQKI_E_INVALID;
}

int niqki_group_insert(niqki_group *g, const int32_t *const *local_sketches, uint32_t per, uint32_t n_total) {
  if (!g || !local_sketches) return NIQKI_E_INVALID;
  if ((uint64_t)n_total > (uint64_t)per * g->world) return gfail(g, NIQKI_E_INVALID, "n_total exceeds world * per");
  if (per == 0) return NIQKI_OK;
  int rc = exchange_slices(g, local_sketches, per);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    // rows are rank major = batch order: the first n_total of them are the batch
    NQ_G(g, l, nqi::insert_dev(ix, (const int32_t *)g->ws[l].allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, n_total));
  }
  return NIQKI_OK;
}

int niqki_group_query(niqki_group *g, const int32_t *const *local_sketches, uint32_t per, uint64_t *const *hit_off,
                      uint32_t *const *hit_counts, uint32_t *const *hit_gids, uint64_t capacity, int mem) {
  if (!g || !local_sketches || !hit_off) return NIQKI_E_INVALID;
  const uint32_t G = g->world, nq = G * per;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    NQ_G(g, l, nqi::build_if_needed(g->sh[l]));
    if (g->sh[l]->built_n != g->sh[0]->built_n) return gfail(g, NIQKI_E_STATE, "the shards hold different numbers of genomes");
  }
  const uint32_t N = g->sh[0]->built_n;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  if (per == 0) return NIQKI_OK;
  int rc = exchange_slices(g, local_sketches, per);
  if (rc) return rc;
  const uint32_t min_score = g->sh[0]->d.min_score;
  bool sparse = g->exchange == 1 || (g->exchange == 0 && min_score >= 4 * G);
  if (min_score < G || N == 0) sparse = false;   // ceil(min_score / G) must be >= 1
  const uint32_t C = g->cand_cap, thr = (min_score + G - 1) / G;
  // 3. partial hit vectors of all queries over the local slots; for the sparse exchange the gather kernel
  //    also leaves every query's candidates (partial count >= ceil(min_score / G))
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    NQ_G(g, l, nqi::ensure(ix, w.counts, std::max<size_t>((size_t)nq * stride * 2, 4)));
    if (N == 0) NQ_GH(g, hipMemsetAsync(w.counts.p, 0, std::max<size_t>((size_t)nq * stride * 2, 4), ix->stream));
    nq::CandOut co;
    if (sparse) {
      NQ_G(g, l, nqi::ensure(ix, w.cand, (size_t)nq::cand_blob_ints(nq, C) * 4));
      co.cand = (int32_t *)w.cand.p;
      co.n = (int32_t *)w.cand.p + (size_t)nq * C;
      co.thr = thr;
      co.cap = C;
    }
    NQ_G(g, l, nqi::counts_dev(ix, (const int32_t *)w.allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, nq,
                               (uint16_t *)w.counts.p, stride, nullptr, sparse ? &co : nullptr));
  }
  // 4. cross-shard sum, scattered by query
  Buf niqki_group::Ws::*red = &niqki_group::Ws::red;
  if (sparse) {
    for (uint32_t l = 0; l < g->n_local; ++l) {
      niqki_index *ix = g->sh[l];
      NQ_GH(g, hipSetDevice(ix->device));
      auto &w = g->ws[l];
      NQ_G(g, l, nqi::ensure(ix, w.cand_all, (size_t)G * nq::cand_blob_ints(nq, C) * 4));
      NQ_G(g, l, nqi::ensure(ix, w.mine, (size_t)nq * G * C * 2));
      NQ_G(g, l, nqi::ensure(ix, w.tot, (size_t)per * G * C * 2));
      NQ_G(g, l, nqi::ensure(ix, w.red, (size_t)per * stride * 2));
      NQ_G(g, l, nqi::ensure(ix, w.flag, 4));
    }
    if ((rc = all_gather(g, &niqki_group::Ws::cand, &niqki_group::Ws::cand_all, (size_t)nq::cand_blob_ints(nq, C) * 4))) return rc;
    for (uint32_t l = 0; l < g->n_local; ++l) {
      niqki_index *ix = g->sh[l];
      NQ_GH(g, hipSetDevice(ix->device));
      auto &w = g->ws[l];
      nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
      NQ_GH(g, hipMemsetAsync(w.flag.p, 0, 4, ix->stream));
      hipLaunchKernelGGL(nq::cand_lookup_kernel, dim3(nq), dim3(256), 0, ix->stream, (const uint16_t *)w.counts.p, stride, nq, G, C,
                         (const int32_t *)w.cand_all.p, (uint16_t *)w.mine.p, (uint32_t *)w.flag.p);
      NQ_GH(g, hipGetLastError());
    }
    if ((rc = reduce_scatter_u32(g, &niqki_group::Ws::mine, &niqki_group::Ws::tot, (size_t)per * G * C / 2))) return rc;
    // every rank saw the same all-gathered list sizes, so all ranks (and processes) take the same branch
    uint32_t over = 0;
    NQ_GH(g, hipSetDevice(g->sh[0]->device));
    NQ_GH(g, hipMemcpyAsync(&over, g->ws[0].flag.p, 4, hipMemcpyDeviceToHost, g->sh[0]->stream));
    NQ_GH(g, hipStreamSynchronize(g->sh[0]->stream));
    if (over) {
      ++g->overflows;
      sparse = false;
    } else {
      for (uint32_t l = 0; l < g->n_local; ++l) {
        niqki_index *ix = g->sh[l];
        NQ_GH(g, hipSetDevice(ix->device));
        auto &w = g->ws[l];
        nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
        NQ_GH(g, hipMemsetAsync(w.red.p, 0, (size_t)per * stride * 2, ix->stream));
        hipLaunchKernelGGL(nq::cand_scatter_kernel, dim3(per), dim3(256), 0, ix->stream, (const uint16_t *)w.tot.p, per,
                           (g->first + l) * per, nq, G, C, (const int32_t *)w.cand_all.p, (uint16_t *)w.red.p, stride);
        NQ_GH(g, hipGetLastError());
      }
    }
  }
  if (!sparse) {
    for (uint32_t l = 0; l < g->n_local; ++l) {
      NQ_GH(g, hipSetDevice(g->sh[l]->device));
      NQ_G(g, l, nqi::ensure(g->sh[l], g->ws[l].red, std::max<size_t>((size_t)per * stride * 2, 4)));
    }
    nqi::Span sp(g->sh[0], NIQKI_KC_EXCHANGE);
    if ((rc = reduce_scatter_u32(g, &niqki_group::Ws::counts, red, (size_t)per * (stride / 2)))) return rc;
  }
  // 5. threshold + order of this rank's queries
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    if (mem == NIQKI_MEM_DEVICE) {
      NQ_G(g, l, nqi::hits_dev(ix, (const uint16_t *)(w.*red).p, per, stride, 0, N, (unsigned long long *)hit_off[l], hit_counts[l],
                               hit_gids[l], capacity, false, nullptr));
      continue;
    }
    NQ_G(g, l, nqi::ensure(ix, w.hitoff, (size_t)(per + 1) * 8));
    NQ_G(g, l, nqi::ensure(ix, w.hc, (size_t)std::max<uint64_t>(capacity, 1) * 4));
    NQ_G(g, l, nqi::ensure(ix, w.hg, (size_t)std::max<uint64_t>(capacity, 1) * 4));
    uint64_t total = 0;
    rc = nqi::hits_dev(ix, (const uint16_t *)(w.*red).p, per, stride, 0, N, (unsigned long long *)w.hitoff.p, (uint32_t *)w.hc.p,
                       (uint32_t *)w.hg.p, capacity, true, &total);
    if (rc && rc != NIQKI_E_CAPACITY) return gfail(g, rc, niqki_last_error(ix));
    NQ_GH(g, hipMemcpyAsync(hit_off[l], w.hitoff.p, (size_t)(per + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
    if (rc == NIQKI_OK && total) {
      NQ_GH(g, hipMemcpyAsync(hit_counts[l], w.hc.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_GH(g, hipMemcpyAsync(hit_gids[l], w.hg.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
    }
    NQ_GH(g, hipStreamSynchronize(ix->stream));
    if (rc == NIQKI_E_CAPACITY) return gfail(g, rc, "hit capacity too small; hit_off holds the sizes needed");
  }
  return NIQKI_OK;
}

namespace {
// per local rank: `per` sketch rows = the first n_entry[l] sketches of the shard's staged batch
// (niqki_stage_raw on that shard), the rest empty sketches (-1)
int staged_rows(niqki_group *g, uint32_t per, const uint32_t *n_entry, std::vector<const int32_t *> &rows) {
  rows.assign(g->n_local, nullptr);
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    const uint32_t n = n_entry ? n_entry[l] : 0;
    if (n > per) return gfail(g, NIQKI_E_INVALID, "a rank's staged entries exceed `per`");
    const size_t row = (size_t)ix->d.F * 4;
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].stpad, std::max<size_t>((size_t)per * row, 4)));
    if (n) {
      if (!ix->staged.valid || ix->staged.n_entry < n) return gfail(g, NIQKI_E_STATE, "rank " + std::to_string(g->first + l) + " has no staged batch of that size");
      NQ_G(g, l, nqi::staged_sketch_ws(ix));
      NQ_GH(g, hipMemcpyAsync(g->ws[l].stpad.p, ix->ws_stsk.p, (size_t)n * row, hipMemcpyDeviceToDevice, ix->stream));
    }
    if (n < per) NQ_GH(g, hipMemsetAsync((char *)g->ws[l].stpad.p + (size_t)n * row, 0xFF, (size_t)(per - n) * row, ix->stream));
    rows[l] = (const int32_t *)g->ws[l].stpad.p;
  }
  return NIQKI_OK;
}
}  // namespace

int niqki_group_staged_insert(niqki_group *g, uint32_t per, const uint32_t *n_entry) {
  if (!g || !n_entry) return NIQKI_E_INVALID;
  if (g->n_local != g->world) return gfail(g, NIQKI_E_STATE, "staged group calls need all ranks in one process");
  std::vector<const int32_t *> rows;
  int rc = staged_rows(g, per, n_entry, rows);
  if (rc) return rc;
  if (per == 0) return NIQKI_OK;
  if ((rc = exchange_slices(g, rows.data(), per))) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
    // ids follow the entries: rank after rank, each rank's valid rows
    for (uint32_t s = 0; s < g->world; ++s)
      NQ_G(g, l, nqi::insert_dev(ix, (const int32_t *)g->ws[l].allsk.p + (size_t)s * per * f_local, f_local, 0, n_entry[s]));
  }
  return NIQKI_OK;
}

int niqki_group_staged_query(niqki_group *g, uint32_t per, const uint32_t *n_entry, uint64_t *const *hit_off,
                             uint32_t *const *hit_counts, uint32_t *const *hit_gids, uint64_t capacity, int mem) {
  if (!g || !n_entry) return NIQKI_E_INVALID;
  if (g->n_local != g->world) return gfail(g, NIQKI_E_STATE, "staged group calls need all ranks in one process");
  std::vector<const int32_t *> rows;
  int rc = staged_rows(g, per, n_entry, rows);
  if (rc) return rc;
  return niqki_group_query(g, rows.data(), per, hit_off, hit_counts, hit_gids, capacity, mem);
}

}  // extern "C"
