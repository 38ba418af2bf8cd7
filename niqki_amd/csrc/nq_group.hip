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
// Transports:
//   rccl   librccl, loaded on first use -- one communicator per local rank, so the same code serves one
//          rank per process (bench.py under torch.distributed.run) and all ranks in one process
//          (`niqki --gpus N`).
//   local  all ranks in one process and two shards share a device (tests, emulation of G shards on one
//          GPU), which RCCL refuses: plain device-to-device copies and a summing kernel stand in for
//          the collectives.
//   ipc    NIQKI_GROUP_TRANSPORT=ipc, one rank per process: every rank maps its peers' exchange buffers
//          (hipIpcGetMemHandle / hipIpcOpenMemHandle, handles passed through a POSIX shared-memory block
//          named by the group id) and PULLS what it needs with a copy or summing kernel -- over xGMI
//          that is a direct all-to-all on all links instead of a ring.  Ordering is by sequence numbers in
//          device memory: tiny signal / wait kernels on the ranks' streams, no host synchronisation in a
//          step.  Ranks may share a device (how the N > 1 path runs on a one-GPU box).
#include "nq_handle.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
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
  decltype(&ncclCommCount) CommCount = nullptr;
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
    NQ_SYM(CommCount, "ncclCommCount")
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

// ---- ipc transport: the block the processes of a group share ------------------------------------
constexpr uint32_t kIpcBufs = 4;      // exchange buffers a rank exposes: send (slices), cand, mine, counts
constexpr uint32_t kIpcMagic = 0x4E514950u;
struct IpcRank {
  // Handles name whole allocations (hipMalloc may carve a pointer out of a larger block, whose base is what
  // hipIpcGetMemHandle takes): the pointer is base + offset in the owner's and in every peer's mapping.
  hipIpcMemHandle_t flags;            // the rank's sequence words (below)
  uint64_t flags_off;
  hipIpcMemHandle_t arena;            // ONE allocation holds the rank's exchange buffers
  uint64_t arena_off;
  uint64_t buf_off[kIpcBufs];         // ... buffer i at arena + buf_off[i]
  uint64_t gen;                       // bumped when `arena` names a new allocation
  int32_t device;
  uint32_t words_kind;                // where the rank's sequence words live (kWords*)
  uint32_t arena_fine;                // 1 = the arena is fine-grained device memory
  char pci[32];                       // the rank's device (hipDeviceGetPCIBusId): peers check that they can reach it
};
// Memory kinds of a rank's sequence words.  A peer GPU polls them from a RUNNING kernel, so they must be
// coherent at system scope by contract, not by observed behaviour: fine-grained device memory (what RCCL keeps
// its own flags in), or -- where that cannot be made or exported -- words in the POSIX shared block itself,
// page-locked and mapped into every process (hipHostRegister: host-coherent).  Plain hipMalloc memory
// (coarse-grained: coherent across devices only at kernel boundaries) is the last resort and is reported.
constexpr uint32_t kWordsCoarse = 0, kWordsFine = 1, kWordsHost = 2;
constexpr uint32_t kFlagWordsMax = 64;
struct IpcShared {
  std::atomic<uint32_t> init, bar_count, bar_gen, abort;
  std::atomic<uint32_t> export_lock;   // one rank at a time allocates and exports its arena
  IpcRank r[kMaxWorld];
  alignas(256) uint32_t host_words[kMaxWorld][kFlagWordsMax];   // kWordsHost: rank r's sequence words
};
// sequence words of a rank (device memory, mapped by every peer): ready[b] = uses of buffer b whose data
// is complete, done[b] = uses whose data this rank has finished reading from ALL peers, err = a wait timed out
constexpr uint32_t kFlagReady = 0, kFlagDone = kIpcBufs, kFlagErr = 2 * kIpcBufs, kFlagWords = kFlagWordsMax;

}  // namespace

namespace nq {

// ---- exchange kernels ----------------------------------------------------------------------
// first slot of rank r (niqki_group_slot_range)
__host__ __device__ inline uint32_t cut(uint32_t F, uint32_t r, uint32_t G) { return (uint32_t)(((uint64_t)F * r) / G); }

// [per][F] int32 sketches -> [G][per][w_max] int16 slices (slot slice of destination g; cells that
// are empty or outside [0, R) -- never indexed or queried, src/niqki_index.cpp:364,:654 -- travel as -1)
// (dst_stride: int16 cells between two destinations' parts, >= per * w_max)
__global__ __launch_bounds__(256) void slice_pack_kernel(const int32_t *sk, uint32_t per, uint32_t F, uint32_t R, uint32_t G,
                                                        uint32_t w_max, int16_t *out, uint64_t dst_stride) {
  const uint32_t q = blockIdx.x, g = blockIdx.y;
  const uint32_t b = cut(F, g, G), w = cut(F, g + 1, G) - b;
  const int32_t *row = sk + (uint64_t)q * F + b;
  int16_t *dst = out + (uint64_t)g * dst_stride + (uint64_t)q * w_max;
  for (uint32_t j = threadIdx.x; j < w_max; j += 256) {
    int32_t x = j < w ? row[j] : -1;
    dst[j] = (x >= 0 && (uint32_t)x < R) ? (int16_t)x : (int16_t)-1;
  }
}

// [nq][w_max] int16 slices of MY slots -> [nq][f_local] int32 sketch rows for the query kernels
// (row q = source rank q / per, its query q % per; src_stride: int16 cells between two sources' parts)
__global__ __launch_bounds__(256) void slice_unpack_kernel(const int16_t *in, uint32_t w_max, uint32_t f_local, int32_t *out, uint32_t per,
                                                          uint64_t src_stride) {
  const uint32_t q = blockIdx.x;
  const int16_t *src = in + (uint64_t)(q / per) * src_stride + (uint64_t)(q % per) * w_max;
  for (uint32_t j = threadIdx.x; j < f_local; j += 256) out[(uint64_t)q * f_local + j] = src[j];
}

// A rank's candidates travel as one blob: nq lists of C ids, the nq list sizes, the nq sizes of its survivor
// lists (one all-gather).
__host__ __device__ inline uint64_t cand_blob_ints(uint32_t nq, uint32_t C) { return (uint64_t)nq * C + 2ull * nq; }

// ---- sparse exchange without counter rows ---------------------------------------------------------
// The ids a query's ranks proposed, m per query: id(q, i) = p[(i / C) * blob + q * C + i % C]
// (the all-gathered blobs: C ids per rank; a flat [nq][m] array: C = m).
struct IdView {
  const int32_t *p;
  uint64_t blob;
  uint32_t C;
  __device__ int32_t at(uint32_t q, uint32_t i) const { return p[(uint64_t)(i / C) * blob + (uint64_t)q * C + i % C]; }
};
__device__ inline uint32_t id_hash(uint32_t id) { return (id * 2654435761u) >> 7; }
constexpr uint32_t kSurvMax = 4096;       // survivors per query a lookup can index (LDS table of 2x)
constexpr uint32_t kCandMax = 4096;       // ids per query cand_hits_kernel can order in LDS

// mine[q][i] = this shard's partial count of id(q, i) (0 for -1): from the shard's survivor list of the
// query (every genome with a partial count >= surv_thr, indexed here by an LDS hash table) or, for an id
// that is not among them -- another rank's candidate that is weak here --, counted exactly as the slots
// in which the genome's stored sketch equals the query's (the identity behind the matrix path, DESIGN.md
// 4.6): f_local strided 2-byte reads per such id, rare by construction.  flag[0] |= a list overflowed.
// CT: uint16_t, or uint32_t where the cross-shard sums can pass 2^16 - 1 (S = 16)
template <typename CT>
__global__ __launch_bounds__(256) void surv_lookup_kernel(const int2 *surv, const int32_t *surv_n, uint32_t SC, uint32_t T, uint32_t m,
                                                         IdView ids, const int32_t *sk, uint32_t q_stride, uint32_t q_off,
                                                         uint32_t f_local, const uint16_t *store, uint64_t store_cap, uint32_t R,
                                                         CT *mine, uint32_t *flag) {
  extern __shared__ __align__(8) int lds_tab[];
  int2 *tab = (int2 *)lds_tab;                  // T x {id, count}
  uint32_t *missing = (uint32_t *)(tab + T);    // one bit per position i < m <= kCandMax
  __shared__ uint32_t acc;
  const uint32_t q = blockIdx.x, tid = threadIdx.x, mw = (m + 31) / 32;
  for (uint32_t i = tid; i < T; i += 256) tab[i] = make_int2(-1, 0);
  for (uint32_t i = tid; i < mw; i += 256) missing[i] = 0;
  __syncthreads();
  const uint32_t ns_all = (uint32_t)surv_n[q], ns = ns_all < SC ? ns_all : SC;   // (an overflow is every rank's to see:
  for (uint32_t i = tid; i < ns; i += 256) {                                     //  blob_overflow_kernel)
    const int2 e = surv[(uint64_t)q * SC + i];
    uint32_t h = id_hash((uint32_t)e.x) & (T - 1);
    while (atomicCAS(&tab[h].x, -1, e.x) != -1) h = (h + 1) & (T - 1);   // (ids of one list are distinct)
    tab[h].y = e.y;
  }
  __syncthreads();
  for (uint32_t i = tid; i < m; i += 256) {
    const int32_t id = ids.at(q, i);
    uint32_t c = 0;
    if (id >= 0) {
      uint32_t h = id_hash((uint32_t)id) & (T - 1);
      bool found = false;
      for (int2 e = tab[h]; e.x != -1; h = (h + 1) & (T - 1), e = tab[h])
        if (e.x == id) { c = (uint32_t)e.y; found = true; break; }
      if (!found) atomicOr(&missing[i >> 5], 1u << (i & 31));
    }
    mine[(uint64_t)q * m + i] = (CT)c;
  }
  __syncthreads();
  // ids this shard holds no survivor entry for: counted exactly from the sketch store, one after the other
  const int32_t *row = sk + (uint64_t)q * q_stride + q_off;
  for (uint32_t w = 0; w < mw; ++w) {
    for (uint32_t bits = missing[w]; bits; bits &= bits - 1) {   // (uniform: every thread reads the same word)
      const uint32_t i = w * 32 + (uint32_t)__builtin_ctz(bits);
      const uint32_t id = (uint32_t)ids.at(q, i);
      if (tid == 0) acc = 0;
      __syncthreads();
      uint32_t c = 0;
      for (uint32_t s_ = tid; s_ < f_local; s_ += 256) {
        const int32_t fp = row[s_];
        c += (fp >= 0 && (uint32_t)fp < R && store[(uint64_t)s_ * store_cap + id] == (uint16_t)fp) ? 1u : 0u;
      }
      if (c) atomicAdd(&acc, c);
      __syncthreads();
      if (tid == 0) mine[(uint64_t)q * m + i] = (CT)acc;
      __syncthreads();
    }
  }
  (void)flag;
}

// The hits of one query from its candidates alone: ids id(first_q + ql, i) with summed counts tot[ql][i]
// (the same id may appear under several ranks, with the same sum): distinct ids whose sum reaches min_score,
// ordered like greater<pair<count, gid>> (src/niqki_index.cpp:685), as 64-bit keys count << 32 | gid.
template <typename CT>
__global__ __launch_bounds__(256) void cand_hits_kernel(const CT *tot, uint32_t first_q, uint32_t m, IdView ids, uint32_t min_score,
                                                       uint32_t T, uint32_t P, unsigned long long *keys, uint32_t *n_out) {
  extern __shared__ __align__(8) int lds_tab[];
  unsigned long long *list = (unsigned long long *)lds_tab;   // P keys
  int32_t *set = (int32_t *)(list + P);                        // T ids
  __shared__ uint32_t cnt;
  const uint32_t ql = blockIdx.x, q = first_q + ql, tid = threadIdx.x;
  for (uint32_t i = tid; i < T; i += 256) set[i] = -1;
  for (uint32_t i = tid; i < P; i += 256) list[i] = 0ull;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (uint32_t i = tid; i < m; i += 256) {
    const int32_t id = ids.at(q, i);
    if (id < 0) continue;
    uint32_t h = id_hash((uint32_t)id) & (T - 1);
    for (;;) {
      const int32_t prev = atomicCAS(&set[h], -1, id);
      if (prev == -1) {   // first of its copies
        const uint32_t c = tot[(uint64_t)ql * m + i];
        if (c >= min_score) list[atomicAdd(&cnt, 1u)] = ((unsigned long long)c << 32) | (uint32_t)id;
        break;
      }
      if (prev == id) break;
      h = (h + 1) & (T - 1);
    }
  }
  __syncthreads();
  const uint32_t n = cnt;
  // bitonic sort, descending (empty places hold 0 and end up last; a real key is never 0: count >= 1 unless
  // min_score is 0, and then gid 0 with count 0 keeps its place among the n first by the write below)
  for (uint32_t k = 2; k <= P; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      for (uint32_t i = tid; i < P; i += 256) {
        const uint32_t l = i ^ j;
        if (l > i) {
          const unsigned long long x = list[i], y = list[l];
          const bool desc = (i & k) == 0;
          if ((x < y) == desc) { list[i] = y; list[l] = x; }
        }
      }
      __syncthreads();
    }
  for (uint32_t i = tid; i < n; i += 256) keys[(uint64_t)ql * m + i] = list[i];
  if (tid == 0) n_out[ql] = n;
}

// S = 16 groups, dense exchange: a shard's counters (<= 2^15 each) as u32 words for the sum, and the sums
// (<= 2^16) back as two u16 planes lo = min(sum, 2^15), hi = sum - lo for the hit kernels' 32-bit form
__global__ __launch_bounds__(256) void widen_rows_kernel(const uint16_t *in, uint64_t n, uint32_t *out) {
  const uint64_t step = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) out[i] = in[i];
}
__global__ __launch_bounds__(256) void split_sums_kernel(const uint32_t *in, uint64_t n, uint16_t *lo, uint16_t *hi) {
  const uint64_t step = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) {
    const uint32_t v = in[i], l = v < 32768u ? v : 32768u;
    lo[i] = (uint16_t)l;
    hi[i] = (uint16_t)(v - l);
  }
}

// flag |= some rank's candidate or survivor list of some query holds more than its capacity
__global__ __launch_bounds__(256) void blob_overflow_kernel(const int32_t *cand_all, uint64_t blob, uint32_t nq, uint32_t G, uint32_t C,
                                                           uint32_t SC, uint32_t *flag) {
  const uint32_t q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  bool over = false;
  for (uint32_t g = 0; g < G; ++g) {
    const int32_t *tail = cand_all + g * blob + (uint64_t)nq * C;
    over |= (uint32_t)tail[q] > C || (uint32_t)tail[nq + q] > SC;
  }
  if (over) atomicOr(flag, 1u);
}

// hit_off[0..nq] = exclusive prefix of n (one workgroup)
__global__ __launch_bounds__(1024) void cand_hits_scan_kernel(const uint32_t *n, uint32_t nq, unsigned long long *hit_off) {
  __shared__ unsigned long long part[1024];
  const uint32_t tid = threadIdx.x, per_t = (nq + 1023) / 1024;
  unsigned long long s = 0;
  for (uint32_t i = tid * per_t; i < (tid + 1) * per_t && i < nq; ++i) s += n[i];
  part[tid] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1) {
    const unsigned long long v = tid >= d ? part[tid - d] : 0ull;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  unsigned long long run = tid ? part[tid - 1] : 0ull;
  for (uint32_t i = tid * per_t; i < (tid + 1) * per_t && i < nq; ++i) { hit_off[i] = run; run += n[i]; }
  if (tid == 1023) hit_off[nq] = part[1023];
}
__global__ __launch_bounds__(256) void cand_hits_write_kernel(const unsigned long long *keys, const uint32_t *n, uint32_t m,
                                                             const unsigned long long *hit_off, uint32_t *hc, uint32_t *hg,
                                                             unsigned long long capacity) {
  const uint32_t ql = blockIdx.x;
  const unsigned long long base = hit_off[ql];
  for (uint32_t i = threadIdx.x; i < n[ql]; i += 256) {
    if (base + i >= capacity) break;
    const unsigned long long k = keys[(uint64_t)ql * m + i];
    hc[base + i] = (uint32_t)(k >> 32);
    hg[base + i] = (uint32_t)k;
  }
}

// ---- ipc transport kernels ------------------------------------------------------------------------
struct FlagPtrs { uint32_t *p[kMaxWorld]; };
struct PeerSrc { const uint8_t *p[kMaxWorld]; };

// everything enqueued before on this stream is complete and visible: publish sequence number `seq`
__global__ void ipc_signal_kernel(uint32_t *word, uint32_t seq) {
  __threadfence_system();
  __hip_atomic_store(word, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// lane r waits until rank r's word has reached seq (one wave; every lane leaves after `timeout` ticks
// of the 100 MHz clock at the latest and reports through *err)
__global__ __launch_bounds__(64) void ipc_wait_kernel(FlagPtrs f, uint32_t world, uint32_t word, uint32_t seq, uint32_t *err,
                                                     unsigned long long timeout) {
  const uint32_t r = threadIdx.x;
  if (r >= world) return;
  const unsigned long long t0 = wall_clock64();
  while ((int32_t)(__hip_atomic_load(f.p[r] + word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
    if (wall_clock64() - t0 > timeout) {
      __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(32);
  }
}
// dst[s * bytes ..] = src.p[s][off ..  off + bytes) for every rank s (bytes and off multiples of 16)
__global__ __launch_bounds__(256) void ipc_pull_kernel(PeerSrc src, uint64_t off, uint64_t bytes, uint8_t *dst) {
  const uint32_t s = blockIdx.y;
  const uint4 *in = (const uint4 *)(src.p[s] + off);
  uint4 *out = (uint4 *)(dst + (uint64_t)s * bytes);
  const uint64_t n = bytes / 16, step = (uint64_t)gridDim.x * 256;
  for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += step) out[i] = in[i];
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
  enum Transport { kLocal, kRccl, kIpc } transport = kLocal;
  std::vector<ncclComm_t> comm;      // per local rank (kRccl)
  int exchange = 0;                  // 0 = choose, 1 = sparse, 2 = dense reduce-scatter
  uint32_t cand_cap = 256;
  uint32_t surv_cap = 1024;          // survivors (partial count >= half the candidate threshold) kept per query and shard
  uint64_t overflows = 0;            // sparse steps redone densely
  bool wide = false;                 // S = 16: cross-shard sums reach 2^16, they travel and add up as u32
  std::string err;
  struct Ws {
    Buf send, recv, allsk, counts, cand, cand_all, mine, tot, red, flag, hitoff, hc, hg, stpad, surv, keys, nhit, rows16, sum32;
    hipEvent_t ev = nullptr;
  };
  std::vector<Ws> ws;
  // ipc transport (n_local == 1)
  struct Ipc {
    IpcShared *shm = nullptr;
    std::string name;
    uint32_t *flags = nullptr;                 // my sequence words (device)
    uint32_t *peer_flags[kMaxWorld] = {};      // every rank's, mine included
    void *peer_buf[kMaxWorld][kIpcBufs] = {};  // mapped exchange buffers (mine: the local pointer)
    void *peer_base[kMaxWorld] = {};           // what hipIpcOpenMemHandle returned for a peer's arena / flags
    void *peer_flags_base[kMaxWorld] = {};
    uint64_t peer_gen[kMaxWorld] = {};
    Buf arena;                                 // my exchange buffers (ws.send / cand / mine / counts are views of it)
    bool ready = false;                        // ipc_setup went through: peers exist and wait in barriers
    uint32_t words_kind = kWordsCoarse;        // where MY sequence words live
    bool arena_fine = false;                   // my arena is fine-grained device memory
    bool shm_registered = false;               // the shared block is page-locked and mapped for this device
    uint32_t (*host_words_dev)[kFlagWordsMax] = nullptr;   // device view of IpcShared::host_words
    uint32_t seq[kIpcBufs] = {};               // uses of each exchange buffer so far
  } ipc;
  // a query batch between niqki_group_query_begin and _end
  struct Pending {
    bool active = false, sparse = false, host = false;
    uint32_t per = 0, N = 0;
    uint64_t stride = 0, capacity = 0;
    uint64_t *const *hit_off = nullptr;
    uint32_t *const *hit_counts = nullptr, *const *hit_gids = nullptr;
    std::vector<uint64_t *> off_v;
    std::vector<uint32_t *> hc_v, hg_v;
  } pend;
  uint32_t *host_flags = nullptr;    // pinned: {candidate overflow, ipc wait timeout} of the batch in flight
  hipEvent_t ev_flags = nullptr;
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

// An RCCL group call that is closed on EVERY way out: a ncclSend / ncclRecv / collective that fails between
// GroupStart and GroupEnd must not leave the communicators with an open group (the NQ_G* macros return from inside).
struct RcclGroupScope {
  bool open = false;
  ncclResult_t begin() {
    const ncclResult_t r = rccl().GroupStart();
    open = r == ncclSuccess;
    return r;
  }
  ncclResult_t end() {
    open = false;
    return rccl().GroupEnd();
  }
  ~RcclGroupScope() {
    if (open) (void)rccl().GroupEnd();
  }
};

// ---- ipc transport, host side ---------------------------------------------------------------------
double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
constexpr double kIpcHostTimeout = 120.0;                       // seconds a process waits for its peers
constexpr unsigned long long kIpcDeviceTimeout = 3000000000ull; // 30 s of the 100 MHz clock for a wait kernel

// all processes of the group have arrived (sense-reversing counter in the shared block)
int ipc_barrier(niqki_group *g) {
  IpcShared *sh = g->ipc.shm;
  const uint32_t gen = sh->bar_gen.load(std::memory_order_acquire);
  if (sh->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == g->world) {
    sh->bar_count.store(0, std::memory_order_relaxed);
    sh->bar_gen.fetch_add(1, std::memory_order_release);
    return NIQKI_OK;
  }
  const double t0 = now_s();
  while (sh->bar_gen.load(std::memory_order_acquire) == gen) {
    if (sh->abort.load(std::memory_order_relaxed)) return gfail(g, NIQKI_E_STATE, "a peer process of the group gave up");
    if (now_s() - t0 > kIpcHostTimeout) {
      sh->abort.store(1, std::memory_order_relaxed);
      return gfail(g, NIQKI_E_STATE, "timed out waiting for the other processes of the group");
    }
    usleep(50);
  }
  return NIQKI_OK;
}

Buf niqki_group::Ws::*const kIpcBufMember[kIpcBufs] = {&niqki_group::Ws::send, &niqki_group::Ws::cand, &niqki_group::Ws::mine,
                                                        &niqki_group::Ws::counts};
int ipc_buf_index(Buf niqki_group::Ws::*m) {
  for (uint32_t i = 0; i < kIpcBufs; ++i)
    if (kIpcBufMember[i] == m) return (int)i;
  return -1;
}

int ipc_export_alloc(niqki_group *g, size_t bytes, bool fine, void **out, hipIpcMemHandle_t *h, uint64_t *off);
bool env_is(const char *name, const char *value) {
  const char *e = std::getenv(name);
  return e && !std::strcmp(e, value);
}
// the shared block page-locked and mapped for my device (kWordsHost words of any rank are read through it)
int ipc_register_block(niqki_group *g) {
  auto &ic = g->ipc;
  if (ic.shm_registered) return NIQKI_OK;
  NQ_GH(g, hipSetDevice(g->sh[0]->device));
  NQ_GH(g, hipHostRegister(ic.shm, sizeof(IpcShared), hipHostRegisterMapped | hipHostRegisterPortable));
  void *d = nullptr;
  hipError_t e = hipHostGetDevicePointer(&d, ic.shm->host_words, 0);
  if (e != hipSuccess) { (void)hipHostUnregister(ic.shm); return gfail(g, NIQKI_E_HIP, std::string("hipHostGetDevicePointer(shared block): ") + hipGetErrorString(e)); }
  ic.host_words_dev = (uint32_t (*)[kFlagWordsMax])d;
  ic.shm_registered = true;
  return NIQKI_OK;
}
// a peer's exported allocation into this process (a few tries: the same transient failure as on the export side)
hipError_t ipc_open(void **q, const hipIpcMemHandle_t &h) {
  hipError_t e = hipSuccess;
  for (int attempt = 0; attempt < 4; ++attempt) {
    if (attempt) { (void)hipGetLastError(); usleep(2000u << attempt); }
    e = hipIpcOpenMemHandle(q, h, hipIpcMemLazyEnablePeerAccess);
    if (e == hipSuccess) break;
  }
  return e;
}

int ipc_setup(niqki_group *g, const uint8_t *id) {
  auto &ic = g->ipc;
  niqki_index *ix = g->sh[0];
  char name[64];
  std::snprintf(name, sizeof name, "/niqki_grp_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", id[0], id[1], id[2], id[3], id[4],
                id[5], id[6], id[7], id[8], id[9], id[10], id[11]);
  ic.name = name;
  int fd = -1;
  const double t0 = now_s();
  if (g->first == 0) {
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return gfail(g, NIQKI_E_STATE, std::string("shm_open(create) failed for ") + name);
    if (ftruncate(fd, (off_t)sizeof(IpcShared)) != 0) { close(fd); shm_unlink(name); return gfail(g, NIQKI_E_STATE, "ftruncate of the group block failed"); }
  } else {
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= sizeof(IpcShared)) break;
      if (fd >= 0) { close(fd); fd = -1; }
      if (now_s() - t0 > kIpcHostTimeout) return gfail(g, NIQKI_E_STATE, "rank 0 of the group never created its shared block");
      usleep(200);
    }
  }
  void *m = mmap(nullptr, sizeof(IpcShared), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) return gfail(g, NIQKI_E_STATE, "mmap of the group block failed");
  ic.shm = (IpcShared *)m;
  if (g->first == 0) ic.shm->init.store(kIpcMagic, std::memory_order_release);
  else
    while (ic.shm->init.load(std::memory_order_acquire) != kIpcMagic) {
      if (now_s() - t0 > kIpcHostTimeout) return gfail(g, NIQKI_E_STATE, "the group's shared block was never initialised");
      usleep(50);
    }
  // my sequence words: fine-grained device memory that peers map (NIQKI_IPC_WORDS=host / coarse force the other kinds)
  NQ_GH(g, hipSetDevice(ix->device));
  IpcRank &me = ic.shm->r[g->first];
  auto give_up = [&](int code, const std::string &why) {   // the peers sit in a barrier: let them leave at once
    ic.shm->abort.store(1, std::memory_order_relaxed);
    return gfail(g, code, why);
  };
  int rc = NIQKI_OK;
  ic.words_kind = env_is("NIQKI_IPC_WORDS", "host") ? kWordsHost : env_is("NIQKI_IPC_WORDS", "coarse") ? kWordsCoarse : kWordsFine;
  if (ic.words_kind != kWordsHost) {
    void *fl = nullptr;
    rc = ipc_export_alloc(g, kFlagWords * 4, ic.words_kind == kWordsFine, &fl, &me.flags, &me.flags_off);
    if (rc && ic.words_kind == kWordsFine) { ic.words_kind = kWordsHost; rc = NIQKI_OK; }   // cannot be made or exported here
    else if (rc) return give_up(rc, g->err);
    else ic.flags = (uint32_t *)fl;
  }
  if (ic.words_kind == kWordsHost) {
    if ((rc = ipc_register_block(g))) return give_up(rc, g->err);
    ic.flags = ic.host_words_dev[g->first];
    std::memset(ic.shm->host_words[g->first], 0, sizeof ic.shm->host_words[g->first]);
  } else {
    NQ_GH(g, hipMemset(ic.flags, 0, kFlagWords * 4));
    NQ_GH(g, hipDeviceSynchronize());
  }
  me.words_kind = ic.words_kind;
  me.device = ix->device;
  std::memset(me.pci, 0, sizeof me.pci);
  if (hipDeviceGetPCIBusId(me.pci, (int)sizeof me.pci - 1, ix->device) != hipSuccess) { (void)hipGetLastError(); me.pci[0] = 0; }
  if ((rc = ipc_barrier(g))) return rc;
  for (uint32_t s = 0; s < g->world; ++s) {
    if (s == g->first) { ic.peer_flags[s] = ic.flags; continue; }
    const IpcRank &pr = ic.shm->r[s];
    // a peer on another device must be reachable from mine (xGMI / PCIe peer access): say so now rather than
    // through a wait kernel's 30 s time-out.  (A peer whose device this process cannot see is left to the mapping.)
    if (pr.pci[0] && me.pci[0] && std::strcmp(pr.pci, me.pci) != 0) {
      int peer_dev = -1, can = 1;
      if (hipDeviceGetByPCIBusId(&peer_dev, pr.pci) == hipSuccess && peer_dev >= 0) {
        if (hipDeviceCanAccessPeer(&can, ix->device, peer_dev) != hipSuccess) { (void)hipGetLastError(); can = 1; }
      } else {
        (void)hipGetLastError();
      }
      if (!can)
        return give_up(NIQKI_E_STATE, "ipc transport: device " + std::string(me.pci) + " (rank " + std::to_string(g->first) +
                                          ") has no peer access to device " + pr.pci + " (rank " + std::to_string(s) +
                                          "): use the rccl transport (NIQKI_GROUP_TRANSPORT=rccl)");
    }
    if (pr.words_kind == kWordsHost) {
      if ((rc = ipc_register_block(g))) return give_up(rc, g->err);
      ic.peer_flags[s] = ic.host_words_dev[s];
      continue;
    }
    void *q = nullptr;
    const hipError_t e = ipc_open(&q, pr.flags);
    if (e != hipSuccess) return give_up(NIQKI_E_HIP, std::string("hipIpcOpenMemHandle(sequence words of rank ") + std::to_string(s) + "): " + hipGetErrorString(e));
    ic.peer_flags_base[s] = q;
    ic.peer_flags[s] = (uint32_t *)((char *)q + pr.flags_off);
  }
  if ((rc = ipc_barrier(g))) return rc;
  if (g->first == 0) shm_unlink(name);   // everybody has it mapped: the name can go
  ic.ready = true;
  return NIQKI_OK;
}

void ipc_teardown(niqki_group *g) {
  auto &ic = g->ipc;
  if (!ic.shm) return;
  for (uint32_t s = 0; s < g->world; ++s) {
    if (s == g->first) continue;
    if (ic.peer_base[s]) (void)hipIpcCloseMemHandle(ic.peer_base[s]);
    if (ic.peer_flags_base[s]) (void)hipIpcCloseMemHandle(ic.peer_flags_base[s]);
  }
  if (ic.flags && ic.words_kind != kWordsHost) (void)hipFree(ic.flags);
  ic.flags = nullptr;
  if (ic.shm_registered) { (void)hipHostUnregister(ic.shm); ic.shm_registered = false; }
  if (ic.arena.p) (void)hipFree(ic.arena.p);
  for (uint32_t b = 0; b < kIpcBufs; ++b) g->ws[0].*kIpcBufMember[b] = Buf();   // (views of the arena)
  if (g->first == 0) shm_unlink(ic.name.c_str());   // (no-op once setup has finished)
  munmap(ic.shm, sizeof(IpcShared));
  ic.shm = nullptr;
}

// Device memory of `bytes` that peers can map: *h names the allocation it lies in, *off its place there
// (hipMalloc may carve a pointer out of a larger block, whose base is what hipIpcGetMemHandle takes).
// Exporting a fresh allocation now and then fails with "invalid argument" when two processes of one GPU do it
// at the same moment (seen once in four runs of two ranks on one MI355X, on the rank that came second): the
// ranks take turns, and a failed export is tried again on a new allocation.
// fine: fine-grained device memory (hipExtMallocWithFlags), coherent at system scope while kernels run.
int ipc_export_alloc(niqki_group *g, size_t bytes, bool fine, void **out, hipIpcMemHandle_t *h, uint64_t *off) {
  auto &ic = g->ipc;
  const double t0 = now_s();
  uint32_t unlocked = 0;
  while (!ic.shm->export_lock.compare_exchange_weak(unlocked, 1u, std::memory_order_acquire)) {
    unlocked = 0;
    if (now_s() - t0 > kIpcHostTimeout) return gfail(g, NIQKI_E_STATE, "a peer never released the group's export lock");
    usleep(50);
  }
  void *p = nullptr, *base = nullptr;
  size_t sz = 0;
  hipError_t e = hipSuccess;
  for (int attempt = 0; attempt < 6; ++attempt) {
    if (p) { (void)hipFree(p); p = nullptr; usleep(2000u << attempt); }
    e = fine ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocFinegrained) : hipMalloc(&p, bytes);
    if (e != hipSuccess) { p = nullptr; break; }
    e = hipMemGetAddressRange((hipDeviceptr_t *)&base, &sz, (hipDeviceptr_t)p);
    if (e != hipSuccess) break;
    e = hipIpcGetMemHandle(h, base);
    if (e == hipSuccess) break;
    (void)hipGetLastError();
  }
  ic.shm->export_lock.store(0u, std::memory_order_release);
  if (e != hipSuccess) {
    char msg[256];
    (void)hipGetLastError();
    std::snprintf(msg, sizeof msg, "%s exchange memory for peers (%zu bytes at %p, allocation %p of %zu bytes): %s",
                  fine ? "fine-grained" : "device", bytes, p, base, sz, hipGetErrorString(e));
    if (p) (void)hipFree(p);
    return gfail(g, NIQKI_E_HIP, msg);
  }
  *out = p;
  *off = (uint64_t)((char *)p - (char *)base);
  return NIQKI_OK;
}

// The exchange buffers of this batch, sized BEFORE any of them is used: every rank takes the same
// decisions (same shapes), so either nobody reallocates -- the steady state, no host traffic at all --
// or everybody does: then all streams drain, the old mappings are closed, the new buffers made, their
// handles published and mapped.
int ipc_prepare(niqki_group *g, const size_t need[kIpcBufs]) {
  auto &ic = g->ipc;
  auto &w = g->ws[0];
  niqki_index *ix = g->sh[0];
  bool grow = false;
  for (uint32_t b = 0; b < kIpcBufs; ++b) grow |= need[b] > (w.*kIpcBufMember[b]).n;
  if (!grow) return NIQKI_OK;
  NQ_GH(g, hipSetDevice(ix->device));
  NQ_GH(g, hipStreamSynchronize(ix->stream));
  int rc = ipc_barrier(g);   // every rank's stream has drained: nobody reads anybody's buffers
  if (rc) return rc;
  for (uint32_t s = 0; s < g->world; ++s)
    if (s != g->first && ic.peer_base[s]) {
      NQ_GH(g, hipIpcCloseMemHandle(ic.peer_base[s]));
      ic.peer_base[s] = nullptr;
      for (uint32_t b = 0; b < kIpcBufs; ++b) ic.peer_buf[s][b] = nullptr;
    }
  if ((rc = ipc_barrier(g))) return rc;
  // one allocation for the four buffers, each with a quarter of headroom (fewer remaps while batches grow)
  IpcRank &me = ic.shm->r[g->first];
  size_t off[kIpcBufs], size[kIpcBufs], total = 0;
  for (uint32_t b = 0; b < kIpcBufs; ++b) {
    const size_t have = (w.*kIpcBufMember[b]).n;
    size[b] = need[b] > have ? need[b] + need[b] / 4 : have;
    size[b] = (size[b] + 255) & ~(size_t)255;
    off[b] = total;
    total += size[b];
  }
  if (ic.arena.p) NQ_GH(g, hipFree(ic.arena.p));
  ic.arena = Buf();
  void *arena = nullptr;
  // fine-grained like the words (peers read it while my later kernels run); plain device memory where that fails
  const bool want_fine = !env_is("NIQKI_IPC_ARENA", "coarse");
  rc = want_fine ? ipc_export_alloc(g, std::max<size_t>(total, 256), true, &arena, &me.arena, &me.arena_off) : NIQKI_E_HIP;
  ic.arena_fine = rc == NIQKI_OK;
  if (rc && (rc = ipc_export_alloc(g, std::max<size_t>(total, 256), false, &arena, &me.arena, &me.arena_off))) {
    ic.shm->abort.store(1, std::memory_order_relaxed);
    return rc;
  }
  me.arena_fine = ic.arena_fine ? 1u : 0u;
  ic.arena.p = arena;
  ic.arena.n = std::max<size_t>(total, 256);
  for (uint32_t b = 0; b < kIpcBufs; ++b) {
    Buf &buf = w.*kIpcBufMember[b];
    buf.p = size[b] ? (char *)ic.arena.p + off[b] : nullptr;
    buf.n = size[b];
    me.buf_off[b] = off[b];
    ic.peer_buf[g->first][b] = buf.p;
  }
  me.gen += 1;
  if ((rc = ipc_barrier(g))) return rc;
  for (uint32_t s = 0; s < g->world; ++s) {
    if (s == g->first) continue;
    const IpcRank &pr = ic.shm->r[s];
    void *q = nullptr;
    NQ_GH(g, ipc_open(&q, pr.arena));
    ic.peer_base[s] = q;
    ic.peer_gen[s] = pr.gen;
    for (uint32_t b = 0; b < kIpcBufs; ++b) ic.peer_buf[s][b] = (char *)q + pr.arena_off + pr.buf_off[b];
  }
  return ipc_barrier(g);
}

// on my stream: wait until word `word` of every rank has reached seq
int ipc_wait_all(niqki_group *g, uint32_t word, uint32_t seq) {
  auto &ic = g->ipc;
  nq::FlagPtrs f{};
  for (uint32_t s = 0; s < g->world; ++s) f.p[s] = ic.peer_flags[s];
  hipLaunchKernelGGL(nq::ipc_wait_kernel, dim3(1), dim3(64), 0, g->sh[0]->stream, f, g->world, word, seq, ic.flags + kFlagErr,
                     kIpcDeviceTimeout);
  NQ_GH(g, hipGetLastError());
  return NIQKI_OK;
}
int ipc_signal(niqki_group *g, uint32_t word, uint32_t seq) {
  hipLaunchKernelGGL(nq::ipc_signal_kernel, dim3(1), dim3(1), 0, g->sh[0]->stream, g->ipc.flags + word, seq);
  NQ_GH(g, hipGetLastError());
  return NIQKI_OK;
}

// Before this rank overwrites exchange buffer `m`: its previous contents have been read by everybody.
// (A no-op on the other transports, whose collectives are ordered by their own streams / RCCL.)
int pre_produce(niqki_group *g, Buf niqki_group::Ws::*m) {
  if (g->transport != niqki_group::kIpc) return NIQKI_OK;
  const int b = ipc_buf_index(m);
  NQ_GH(g, hipSetDevice(g->sh[0]->device));
  return ipc_wait_all(g, kFlagDone + (uint32_t)b, g->ipc.seq[b]);
}

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

// ipc form of the three collectives: publish my buffer, wait for everybody's, pull (copy or sum), say so
template <typename Pull>
int ipc_collective(niqki_group *g, Buf niqki_group::Ws::*send, Pull &&pull) {
  auto &ic = g->ipc;
  const int b = ipc_buf_index(send);
  if (b < 0) return gfail(g, NIQKI_E_INVALID, "not an exchange buffer");
  NQ_GH(g, hipSetDevice(g->sh[0]->device));
  const uint32_t seq = ++ic.seq[b];
  int rc = ipc_signal(g, kFlagReady + (uint32_t)b, seq);
  if (!rc) rc = ipc_wait_all(g, kFlagReady + (uint32_t)b, seq);
  if (rc) return rc;
  for (uint32_t s = 0; s < g->world; ++s)
    if (!ic.peer_buf[s][b]) return gfail(g, NIQKI_E_STATE, "a peer's exchange buffer is not mapped");
  if ((rc = pull(b))) return rc;
  return ipc_signal(g, kFlagDone + (uint32_t)b, seq);
}

// recv[l] = [world][bytes] <- send[s] + l * bytes of every rank s (bytes: a multiple of 16)
int all_to_all(niqki_group *g, Buf niqki_group::Ws::*send, Buf niqki_group::Ws::*recv, size_t bytes) {
  if (g->transport == niqki_group::kRccl) {
    RcclGroupScope grp;
    NQ_GN(g, grp.begin());
    for (uint32_t l = 0; l < g->n_local; ++l) {
      NQ_GH(g, hipSetDevice(g->sh[l]->device));   // (several communicators in one thread: each call on its own device)
      const char *s = (const char *)(g->ws[l].*send).p;
      char *r = (char *)(g->ws[l].*recv).p;
      for (uint32_t p = 0; p < g->world; ++p) {
        NQ_GN(g, rccl().Send(s + (size_t)p * bytes, bytes, ncclUint8, (int)p, g->comm[l], g->sh[l]->stream));
        NQ_GN(g, rccl().Recv(r + (size_t)p * bytes, bytes, ncclUint8, (int)p, g->comm[l], g->sh[l]->stream));
      }
    }
    NQ_GN(g, grp.end());
    return NIQKI_OK;
  }
  if (g->transport == niqki_group::kIpc)
    return ipc_collective(g, send, [&](int b) {
      nq::PeerSrc src{};
      for (uint32_t s = 0; s < g->world; ++s) src.p[s] = (const uint8_t *)g->ipc.peer_buf[s][b];
      const uint32_t bx = (uint32_t)std::min<uint64_t>((bytes / 16 + 255) / 256, 1024);
      hipLaunchKernelGGL(nq::ipc_pull_kernel, dim3(std::max(bx, 1u), g->world), dim3(256), 0, g->sh[0]->stream, src,
                         (uint64_t)g->first * bytes, (uint64_t)bytes, (uint8_t *)(g->ws[0].*recv).p);
      NQ_GH(g, hipGetLastError());
      return (int)NIQKI_OK;
    });
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

// recv[l] = [world][bytes] <- send[s] of every rank s (bytes: a multiple of 16)
int all_gather(niqki_group *g, Buf niqki_group::Ws::*send, Buf niqki_group::Ws::*recv, size_t bytes) {
  if (g->transport == niqki_group::kRccl) {
    RcclGroupScope grp;
    NQ_GN(g, grp.begin());
    for (uint32_t l = 0; l < g->n_local; ++l) {
      NQ_GH(g, hipSetDevice(g->sh[l]->device));
      NQ_GN(g, rccl().AllGather((g->ws[l].*send).p, (g->ws[l].*recv).p, bytes, ncclUint8, g->comm[l], g->sh[l]->stream));
    }
    NQ_GN(g, grp.end());
    return NIQKI_OK;
  }
  if (g->transport == niqki_group::kIpc)
    return ipc_collective(g, send, [&](int b) {
      nq::PeerSrc src{};
      for (uint32_t s = 0; s < g->world; ++s) src.p[s] = (const uint8_t *)g->ipc.peer_buf[s][b];
      const uint32_t bx = (uint32_t)std::min<uint64_t>((bytes / 16 + 255) / 256, 1024);
      hipLaunchKernelGGL(nq::ipc_pull_kernel, dim3(std::max(bx, 1u), g->world), dim3(256), 0, g->sh[0]->stream, src, (uint64_t)0,
                         (uint64_t)bytes, (uint8_t *)(g->ws[0].*recv).p);
      NQ_GH(g, hipGetLastError());
      return (int)NIQKI_OK;
    });
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
  if (g->transport == niqki_group::kRccl) {
    RcclGroupScope grp;
    NQ_GN(g, grp.begin());
    for (uint32_t l = 0; l < g->n_local; ++l) {
      NQ_GH(g, hipSetDevice(g->sh[l]->device));
      NQ_GN(g, rccl().ReduceScatter((g->ws[l].*send).p, (g->ws[l].*recv).p, count, ncclUint32, ncclSum, g->comm[l], g->sh[l]->stream));
    }
    NQ_GN(g, grp.end());
    return NIQKI_OK;
  }
  const uint32_t blocks = (uint32_t)std::min<uint64_t>((count + 255) / 256, 8192);
  if (g->transport == niqki_group::kIpc)
    return ipc_collective(g, send, [&](int b) {
      if (count == 0) return (int)NIQKI_OK;
      nq::SumSrc src{};
      for (uint32_t s = 0; s < g->world; ++s) src.p[s] = (const uint32_t *)g->ipc.peer_buf[s][b];
      hipLaunchKernelGGL(nq::sum_rows_kernel, dim3(blocks), dim3(256), 0, g->sh[0]->stream, src, g->world, (uint64_t)g->first * count,
                         (uint64_t)count, (uint32_t *)(g->ws[0].*recv).p);
      NQ_GH(g, hipGetLastError());
      return (int)NIQKI_OK;
    });
  int rc = cross_wait(g);
  if (rc) return rc;
  nq::SumSrc src{};
  for (uint32_t s = 0; s < g->n_local; ++s) src.p[s] = (const uint32_t *)(g->ws[s].*send).p;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    if (count == 0) continue;
    hipLaunchKernelGGL(nq::sum_rows_kernel, dim3(blocks), dim3(256), 0, g->sh[l]->stream, src, g->world, (uint64_t)l * count,
                       (uint64_t)count, (uint32_t *)(g->ws[l].*recv).p);
    NQ_GH(g, hipGetLastError());
  }
  return cross_wait(g);
}

// bytes of one destination's slices of `per` queries (16-byte granular for the pull kernel)
size_t slice_bytes(uint32_t per, uint32_t w_max) { return (((size_t)per * w_max * 2) + 15) & ~(size_t)15; }
// the decisions and sizes of one query batch (niqki_group_plan_batch: also what the CPU test of the protocol uses)
void plan_batch(uint32_t G, uint32_t S, uint32_t min_score, int exchange, uint32_t per, uint32_t N, uint32_t C, niqki_group_plan *o) {
  const uint32_t F = 1u << S, nq_ = G * per;
  bool sparse = exchange == 1 || (exchange == 0 && min_score >= 4 * G);
  if (min_score < G || N == 0) sparse = false;                    // ceil(min_score / G) must be >= 1
  if ((uint64_t)G * C > nq::kCandMax) sparse = false;             // (cand_hits_kernel orders a query's candidates in LDS)
  o->sparse = sparse ? 1u : 0u;
  o->cand_threshold = (min_score + G - 1) / G;
  o->surv_threshold = std::max(1u, o->cand_threshold / 2);
  o->slice_slots = (F + G - 1) / G;
  o->slice_bytes = slice_bytes(per, o->slice_slots);
  o->cand_blob_bytes = (((size_t)nq::cand_blob_ints(nq_, C) * 4) + 15) & ~(size_t)15;
  o->row_stride = NIQKI_ROW_STRIDE(N);
  const bool wide = S > 15;                                       // cross-shard sums reach 2^16: they travel as u32
  o->sum_words = sparse ? (wide ? (uint64_t)per * G * C : (uint64_t)per * G * C / 2)
                        : (wide ? (uint64_t)per * o->row_stride : (uint64_t)per * (o->row_stride / 2));
}
size_t blob_bytes(uint32_t nq_, uint32_t C) { return (((size_t)nq::cand_blob_ints(nq_, C) * 4) + 15) & ~(size_t)15; }

// steps 1-3 shared by insert and query: local sketches -> compact rows of all world*per sketches
// restricted to each rank's slots (ws.allsk, stride f_local)
int exchange_slices(niqki_group *g, const int32_t *const *local_sketches, uint32_t per) {
  const uint32_t G = g->world, F = g->sh[0]->d.F, R = g->sh[0]->d.R;
  const uint32_t w_max = (F + G - 1) / G;
  const size_t bytes = slice_bytes(per, w_max);
  int rc = pre_produce(g, &niqki_group::Ws::send);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].send, bytes * G));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].recv, bytes * G));
    nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
    hipLaunchKernelGGL(nq::slice_pack_kernel, dim3(per, G), dim3(256), 0, ix->stream, local_sketches[l], per, F, R, G, w_max,
                       (int16_t *)g->ws[l].send.p, (uint64_t)(bytes / 2));
    NQ_GH(g, hipGetLastError());
  }
  {
    nqi::Span sp(g->sh[0], NIQKI_KC_EXCHANGE);
    if ((rc = all_to_all(g, &niqki_group::Ws::send, &niqki_group::Ws::recv, bytes))) return rc;
  }
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
    NQ_GH(g, hipSetDevice(ix->device));
    NQ_G(g, l, nqi::ensure(ix, g->ws[l].allsk, (size_t)G * per * f_local * 4));
    nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
    hipLaunchKernelGGL(nq::slice_unpack_kernel, dim3(G * per), dim3(256), 0, ix->stream, (const int16_t *)g->ws[l].recv.p, w_max,
                       f_local, (int32_t *)g->ws[l].allsk.p, per, (uint64_t)(bytes / 2));
    NQ_GH(g, hipGetLastError());
  }
  return NIQKI_OK;
}

// exchange buffer sizes of a batch (ipc: made before the first of them is used)
int prepare_batch(niqki_group *g, uint32_t per, uint32_t N, bool query, bool sparse) {
  if (g->transport != niqki_group::kIpc) return NIQKI_OK;
  const uint32_t G = g->world, F = g->sh[0]->d.F, nq_ = G * per, C = g->cand_cap;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  size_t need[kIpcBufs] = {slice_bytes(per, (F + G - 1) / G) * G, 0, 0, 0};
  if (query) {
    const size_t cw = g->wide ? 4 : 2;   // bytes per exchanged count
    if (sparse) {   // no counter rows (they are made only if a batch has to be redone densely)
      need[1] = blob_bytes(nq_, C);
      need[2] = (size_t)nq_ * G * C * cw;
    } else {
      need[3] = std::max<size_t>((size_t)nq_ * stride * cw, 4);
    }
  }
  return ipc_prepare(g, need);
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
  static_assert(NIQKI_GROUP_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "group id is an ncclUniqueId");
  const char *t = std::getenv("NIQKI_GROUP_TRANSPORT");
  if (t && !std::strcmp(t, "ipc")) {   // no RCCL involved: 128 random bytes name the group
    FILE *f = std::fopen("/dev/urandom", "rb");
    const bool ok = f && std::fread(id, 1, NIQKI_GROUP_ID_BYTES, f) == NIQKI_GROUP_ID_BYTES;
    if (f) std::fclose(f);
    return ok ? NIQKI_OK : NIQKI_E_STATE;
  }
  if (!rccl().load()) return NIQKI_E_STATE;
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
  g->wide = shards[0] && shards[0]->d.S > 15;
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
    if (ix->d.S > 15 && world < 2)
      return bail(NIQKI_E_INVALID, "an S = 16 group needs at least two shards (a shard counts at most 2^15 slots in u16)");
    if (ix->d.K != shards[0]->d.K || ix->d.W != shards[0]->d.W || ix->d.min_score != shards[0]->d.min_score ||
        ix->n_genomes != shards[0]->n_genomes)
      return bail(NIQKI_E_INVALID, "the shards of a group must agree in K, W, min_score and genome count");
    for (uint32_t m = 0; m < l; ++m) shared_device |= shards[m]->device == ix->device;
  }
  const char *tenv = std::getenv("NIQKI_GROUP_TRANSPORT");
  const bool want_ipc = tenv && !std::strcmp(tenv, "ipc"), want_local = tenv && !std::strcmp(tenv, "local");
  const bool want_rccl = tenv && !std::strcmp(tenv, "rccl");   // asked for by name: RCCL decides whether it takes the devices
  if (want_ipc && n_local == 1 && world > 1) g->transport = niqki_group::kIpc;
  else if (n_local == world && !want_rccl && (shared_device || want_local || want_ipc)) g->transport = niqki_group::kLocal;
  else g->transport = niqki_group::kRccl;
  if (g->transport != niqki_group::kIpc && shared_device && n_local != world && !want_rccl)
    return bail(NIQKI_E_INVALID, "shards that share a device need all ranks in one process (or NIQKI_GROUP_TRANSPORT=ipc)");
  if (g->transport == niqki_group::kIpc && n_local != 1)
    return bail(NIQKI_E_INVALID, "the ipc transport wants one rank per process (or all ranks in one: local)");
  for (uint32_t l = 0; l < n_local; ++l) {
    if (hipSetDevice(shards[l]->device) != hipSuccess || hipEventCreateWithFlags(&g->ws[l].ev, hipEventDisableTiming) != hipSuccess)
      return bail(NIQKI_E_HIP, "hipEventCreate failed");
  }
  if (hipSetDevice(shards[0]->device) != hipSuccess || hipHostMalloc((void **)&g->host_flags, 64, hipHostMallocDefault) != hipSuccess ||
      hipEventCreateWithFlags(&g->ev_flags, hipEventDisableTiming) != hipSuccess)
    return bail(NIQKI_E_HIP, "pinned flag words / event could not be made");
  g->host_flags[0] = g->host_flags[1] = 0;
  if (g->transport == niqki_group::kRccl) {
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
  } else if (g->transport == niqki_group::kIpc) {
    if (!id) return bail(NIQKI_E_INVALID, "a group spanning several processes needs the id of niqki_group_new_id");
    const int rc = ipc_setup(g, id);
    if (rc) return bail(rc, g->err);
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
  }
  if (g->transport == niqki_group::kIpc && g->ipc.shm) {
    if (g->ipc.ready) (void)ipc_barrier(g);   // nobody still reads my buffers (a dead peer costs the timeout)
    ipc_teardown(g);
  }
  for (uint32_t l = 0; l < g->n_local && l < g->ws.size(); ++l) {
    auto &w = g->ws[l];   // (ipc_teardown has dropped the views of its arena)
    for (Buf *b : {&w.send, &w.recv, &w.allsk, &w.counts, &w.cand, &w.cand_all, &w.mine, &w.tot, &w.red,
                   &w.flag, &w.hitoff, &w.hc, &w.hg, &w.stpad, &w.surv, &w.keys, &w.nhit, &w.rows16, &w.sum32})
      if (b->p) (void)hipFree(b->p);
    if (w.ev) (void)hipEventDestroy(w.ev);
    if (l < g->comm.size() && g->comm[l]) (void)rccl().CommDestroy(g->comm[l]);
  }
  if (g->host_flags) (void)hipHostFree(g->host_flags);
  if (g->ev_flags) (void)hipEventDestroy(g->ev_flags);
  delete g;
}

const char *niqki_group_last_error(const niqki_group *g) { return g ? g->err.c_str() : ""; }

int niqki_group_plan_batch(uint32_t world, uint32_t S, uint32_t min_score, int exchange_option, uint32_t per, uint32_t n_genomes,
                           uint32_t cand_cap, niqki_group_plan *out) {
  if (!out || world == 0 || world > kMaxWorld || S == 0 || S > 16 || exchange_option < 0 || exchange_option > 2) return NIQKI_E_INVALID;
  plan_batch(world, S, min_score, exchange_option, per, n_genomes, cand_cap, out);
  return NIQKI_OK;
}

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
  if (!std::strcmp(key, "surv_cap")) {
    if (value < 2 || value > (int64_t)nq::kSurvMax) return gfail(g, NIQKI_E_INVALID, "surv_cap must be in 2..4096");
    g->surv_cap = (uint32_t)value;
    return NIQKI_OK;
  }
  return gfail(g, NIQKI_E_INVALID, std::string("unknown group option ") + key);
}

int niqki_group_get_stat(const niqki_group *g, const char *key, uint64_t *value) {
  if (!g || !key || !value) return NIQKI_E_INVALID;
  if (!std::strcmp(key, "overflows")) { *value = g->overflows; return NIQKI_OK; }
  if (!std::strcmp(key, "rccl")) { *value = g->transport == niqki_group::kRccl ? 1 : 0; return NIQKI_OK; }
  if (!std::strcmp(key, "transport")) { *value = (uint64_t)g->transport; return NIQKI_OK; }   // 0 local, 1 rccl, 2 ipc
  if (!std::strcmp(key, "ipc_words_kind")) { *value = g->ipc.words_kind; return NIQKI_OK; }   // 0 coarse, 1 fine-grained, 2 host block
  if (!std::strcmp(key, "ipc_arena_fine")) { *value = g->ipc.arena_fine ? 1 : 0; return NIQKI_OK; }
  if (!std::strcmp(key, "ranks_seen")) {
    // how many ranks the transport itself knows of: the communicator's size as RCCL reports it, the peers whose
    // sequence words this process has mapped (ipc), the shards of the process (local)
    if (g->transport == niqki_group::kRccl) {
      int n = 0;
      if (g->comm.empty() || !g->comm[0] || rccl().CommCount(g->comm[0], &n) != ncclSuccess) return NIQKI_E_STATE;
      *value = (uint64_t)n;
    } else if (g->transport == niqki_group::kIpc) {
      uint64_t n = 0;
      for (uint32_t s = 0; s < g->world; ++s) n += g->ipc.peer_flags[s] ? 1 : 0;
      *value = n;
    } else {
      *value = g->n_local;
    }
    return NIQKI_OK;
  }
  if (!std::strcmp(key, "sparse")) {
    const uint32_t ms = g->sh[0]->d.min_score;
    niqki_group_plan plan;
    plan_batch(g->world, g->sh[0]->d.S, ms, g->exchange, 1, std::max(1u, g->sh[0]->n_genomes), g->cand_cap, &plan);
    *value = plan.sparse;
    return NIQKI_OK;
  }
  return NIQKI_E_INVALID;
}

int niqki_group_insert(niqki_group *g, const int32_t *const *local_sketches, uint32_t per, uint32_t n_total) {
  if (!g || !local_sketches) return NIQKI_E_INVALID;
  if (g->pend.active) return gfail(g, NIQKI_E_STATE, "a query batch is in flight (niqki_group_query_end)");
  if ((uint64_t)n_total > (uint64_t)per * g->world) return gfail(g, NIQKI_E_INVALID, "n_total exceeds world * per");
  if (per == 0) return NIQKI_OK;
  int rc = prepare_batch(g, per, 0, false, false);
  if (rc) return rc;
  if ((rc = exchange_slices(g, local_sketches, per))) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    // rows are rank major = batch order: the first n_total of them are the batch
    NQ_G(g, l, nqi::insert_dev(ix, (const int32_t *)g->ws[l].allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, n_total));
  }
  return NIQKI_OK;
}

namespace {

uint32_t pow2_at_least(uint32_t x) {
  uint32_t p = 1;
  while (p < x) p <<= 1;
  return p;
}

// this shard's partial counts of the ids a query's ranks proposed (nq::surv_lookup_kernel)
// (wide: `mine` holds uint32_t values)
hipError_t launch_surv_lookup(niqki_index *ix, const int2 *surv, const int32_t *surv_n, uint32_t SC, uint32_t nq_, uint32_t m,
                              const nq::IdView &ids, const int32_t *sk, uint32_t q_stride, uint32_t q_off, void *mine,
                              uint32_t *flag, bool wide = false) {
  if (nq_ == 0 || m == 0) return hipSuccess;
  const uint32_t T = pow2_at_least(2 * std::max(SC, 1u));
  const size_t lds = (size_t)T * 8 + (size_t)((m + 31) / 32) * 4;
  const uint32_t f_local = ix->d.slot_end - ix->d.slot_begin;
  if (wide) {
    hipError_t e = hipFuncSetAttribute((const void *)nq::surv_lookup_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nq::surv_lookup_kernel<uint32_t>, dim3(nq_), dim3(256), lds, ix->stream, surv, surv_n, SC, T, m, ids, sk, q_stride,
                       q_off, f_local, (const uint16_t *)ix->store, (uint64_t)ix->cap, ix->d.R, (uint32_t *)mine, flag);
  } else {
    hipError_t e = hipFuncSetAttribute((const void *)nq::surv_lookup_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(nq::surv_lookup_kernel<uint16_t>, dim3(nq_), dim3(256), lds, ix->stream, surv, surv_n, SC, T, m, ids, sk, q_stride,
                       q_off, f_local, (const uint16_t *)ix->store, (uint64_t)ix->cap, ix->d.R, (uint16_t *)mine, flag);
  }
  return hipGetLastError();
}

// hits of `per` queries (first_q ..) from candidate ids and their summed counts; keys / nhit: scratch
int hits_from_candidates(niqki_index *ix, const void *tot, uint32_t first_q, uint32_t per, uint32_t m, const nq::IdView &ids,
                         uint32_t min_score, Buf &keys, Buf &nhit, unsigned long long *hit_off, uint32_t *hc, uint32_t *hg,
                         uint64_t capacity, bool wide = false) {
  if (per == 0) return NIQKI_OK;
  if (m > nq::kCandMax) return nqi::fail(ix, NIQKI_E_INVALID, "too many candidate ids per query");
  int rc = nqi::ensure(ix, keys, std::max<size_t>((size_t)per * m * 8, 8));
  if (!rc) rc = nqi::ensure(ix, nhit, (size_t)per * 4);
  if (rc) return rc;
  const uint32_t P = pow2_at_least(std::max(m, 2u)), T = 2 * P;
  const size_t lds = (size_t)P * 8 + (size_t)T * 4;
  nqi::Span sp(ix, NIQKI_KC_HITS);
  if (wide) {
    NQ_HIP(ix, hipFuncSetAttribute((const void *)nq::cand_hits_kernel<uint32_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nq::cand_hits_kernel<uint32_t>, dim3(per), dim3(256), lds, ix->stream, (const uint32_t *)tot, first_q, m, ids,
                       min_score, T, P, (unsigned long long *)keys.p, (uint32_t *)nhit.p);
  } else {
    NQ_HIP(ix, hipFuncSetAttribute((const void *)nq::cand_hits_kernel<uint16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nq::cand_hits_kernel<uint16_t>, dim3(per), dim3(256), lds, ix->stream, (const uint16_t *)tot, first_q, m, ids,
                       min_score, T, P, (unsigned long long *)keys.p, (uint32_t *)nhit.p);
  }
  hipLaunchKernelGGL(nq::cand_hits_scan_kernel, dim3(1), dim3(1024), 0, ix->stream, (const uint32_t *)nhit.p, per, hit_off);
  hipLaunchKernelGGL(nq::cand_hits_write_kernel, dim3(per), dim3(256), 0, ix->stream, (const unsigned long long *)keys.p,
                     (const uint32_t *)nhit.p, m, (const unsigned long long *)hit_off, hc, hg, (unsigned long long)capacity);
  NQ_HIP(ix, hipGetLastError());
  return NIQKI_OK;
}

// 3 (dense form): counter rows of all queries over the local slots
int gather_rows(niqki_group *g, uint32_t nq_, uint32_t N, uint64_t stride) {
  int rc = pre_produce(g, &niqki_group::Ws::counts);
  if (rc) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    const size_t cells = std::max<size_t>((size_t)nq_ * stride, 2);
    NQ_G(g, l, nqi::ensure(ix, w.counts, cells * (g->wide ? 4 : 2)));
    Buf &rows = g->wide ? w.rows16 : w.counts;   // S = 16: u16 rows of the shard first, widened for the sum
    if (g->wide) NQ_G(g, l, nqi::ensure(ix, w.rows16, cells * 2));
    if (N == 0) NQ_GH(g, hipMemsetAsync(rows.p, 0, cells * 2, ix->stream));
    NQ_G(g, l, nqi::counts_dev(ix, (const int32_t *)w.allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, nq_, (uint16_t *)rows.p,
                               stride, nullptr, nullptr));
    if (g->wide) {
      hipLaunchKernelGGL(nq::widen_rows_kernel, dim3(4096), dim3(256), 0, ix->stream, (const uint16_t *)w.rows16.p,
                         (uint64_t)nq_ * stride, (uint32_t *)w.counts.p);
      NQ_GH(g, hipGetLastError());
    }
  }
  return NIQKI_OK;
}

// 4b + 5 of a batch whose partial hit vectors (ws.counts) are complete: dense reduce-scatter, threshold, order
int finish_dense(niqki_group *g, uint32_t per, uint32_t N, uint64_t stride) {
  int rc;
  const size_t cells = std::max<size_t>((size_t)per * stride, 2);
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    NQ_G(g, l, nqi::ensure(g->sh[l], g->ws[l].red, cells * 2 * (g->wide ? 2 : 1)));   // (S = 16: two planes)
    if (g->wide) NQ_G(g, l, nqi::ensure(g->sh[l], g->ws[l].sum32, cells * 4));
  }
  nqi::Span sp(g->sh[0], NIQKI_KC_EXCHANGE);
  if (!g->wide) return reduce_scatter_u32(g, &niqki_group::Ws::counts, &niqki_group::Ws::red, (size_t)per * (stride / 2));
  if ((rc = reduce_scatter_u32(g, &niqki_group::Ws::counts, &niqki_group::Ws::sum32, (size_t)per * stride))) return rc;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    hipLaunchKernelGGL(nq::split_sums_kernel, dim3(4096), dim3(256), 0, ix->stream, (const uint32_t *)w.sum32.p, (uint64_t)per * stride,
                       (uint16_t *)w.red.p, (uint16_t *)w.red.p + cells);
    NQ_GH(g, hipGetLastError());
  }
  (void)N;
  return NIQKI_OK;
}

// 5. threshold + order of this rank's queries from ws.red into the caller's device buffers or the
// library's staging buffers (host results: copied out by query_end)
int run_hits(niqki_group *g, uint32_t per, uint32_t N, uint64_t stride) {
  auto &pd = g->pend;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    const uint16_t *plane2 = g->wide ? (const uint16_t *)w.red.p + std::max<size_t>((size_t)per * stride, 2) : nullptr;
    if (!pd.host) {
      NQ_G(g, l, nqi::hits_dev(ix, (const uint16_t *)w.red.p, per, stride, 0, N, (unsigned long long *)pd.hit_off[l], pd.hit_counts[l],
                               pd.hit_gids[l], pd.capacity, false, nullptr, plane2));
      continue;
    }
    NQ_G(g, l, nqi::ensure(ix, w.hitoff, (size_t)(per + 1) * 8));
    NQ_G(g, l, nqi::ensure(ix, w.hc, (size_t)std::max<uint64_t>(pd.capacity, 1) * 4));
    NQ_G(g, l, nqi::ensure(ix, w.hg, (size_t)std::max<uint64_t>(pd.capacity, 1) * 4));
    NQ_G(g, l, nqi::hits_dev(ix, (const uint16_t *)w.red.p, per, stride, 0, N, (unsigned long long *)w.hitoff.p, (uint32_t *)w.hc.p,
                             (uint32_t *)w.hg.p, pd.capacity, false, nullptr, plane2));
  }
  return NIQKI_OK;
}

}  // namespace

int niqki_group_query_begin(niqki_group *g, const int32_t *const *local_sketches, uint32_t per, uint64_t *const *hit_off,
                            uint32_t *const *hit_counts, uint32_t *const *hit_gids, uint64_t capacity, int mem) {
  if (!g || !local_sketches || !hit_off) return NIQKI_E_INVALID;
  if (g->pend.active) return gfail(g, NIQKI_E_STATE, "a query batch is already in flight (niqki_group_query_end)");
  const uint32_t G = g->world, nq = G * per;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    NQ_GH(g, hipSetDevice(g->sh[l]->device));
    NQ_G(g, l, nqi::build_if_needed(g->sh[l]));
    if (g->sh[l]->built_n != g->sh[0]->built_n) return gfail(g, NIQKI_E_STATE, "the shards hold different numbers of genomes");
  }
  const uint32_t N = g->sh[0]->built_n;
  const uint64_t stride = NIQKI_ROW_STRIDE(N);
  auto &pd = g->pend;
  pd.per = per; pd.N = N; pd.stride = stride; pd.capacity = capacity; pd.host = mem != NIQKI_MEM_DEVICE;
  pd.off_v.assign(hit_off, hit_off + g->n_local);
  pd.hc_v.assign(hit_counts, hit_counts + g->n_local);
  pd.hg_v.assign(hit_gids, hit_gids + g->n_local);
  pd.hit_off = pd.off_v.data(); pd.hit_counts = pd.hc_v.data(); pd.hit_gids = pd.hg_v.data();
  pd.sparse = false;
  pd.active = true;
  if (per == 0) return NIQKI_OK;
  const uint32_t min_score = g->sh[0]->d.min_score;
  niqki_group_plan plan;
  plan_batch(G, g->sh[0]->d.S, min_score, g->exchange, per, N, g->cand_cap, &plan);
  const bool sparse = plan.sparse != 0;
  pd.sparse = sparse;
  int rc = prepare_batch(g, per, N, true, sparse);
  if (!rc) rc = exchange_slices(g, local_sketches, per);
  if (rc) { pd.active = false; return rc; }
  const uint32_t C = g->cand_cap, SC = g->surv_cap, thr = plan.cand_threshold;
  if (sparse) {
    // 3. gather over the local slots WITHOUT counter rows: what leaves the kernel is, per query, the candidates
    //    (partial count >= ceil(min_score / G): their ids travel) and the survivors (partial count >= half of
    //    that, with their counts: what the other ranks' candidates are looked up in)
    const size_t blob = blob_bytes(nq, C);
    if ((rc = pre_produce(g, &niqki_group::Ws::cand))) return rc;
    for (uint32_t l = 0; l < g->n_local; ++l) {
      niqki_index *ix = g->sh[l];
      NQ_GH(g, hipSetDevice(ix->device));
      auto &w = g->ws[l];
      NQ_G(g, l, nqi::ensure(ix, w.cand, blob));
      NQ_G(g, l, nqi::ensure(ix, w.surv, (size_t)nq * SC * 8));
      NQ_G(g, l, nqi::ensure(ix, w.cand_all, (size_t)G * blob));
      NQ_G(g, l, nqi::ensure(ix, w.mine, (size_t)nq * G * C * (g->wide ? 4 : 2)));
      NQ_G(g, l, nqi::ensure(ix, w.tot, (size_t)per * G * C * (g->wide ? 4 : 2)));
      NQ_G(g, l, nqi::ensure(ix, w.flag, 4));
      nq::CandOut co;
      co.cand = (int32_t *)w.cand.p;
      co.n = (int32_t *)w.cand.p + (size_t)nq * C;
      co.thr = thr;
      co.cap = C;
      co.surv = (int2 *)w.surv.p;
      co.surv_n = co.n + nq;
      co.surv_thr = plan.surv_threshold;
      co.surv_cap = SC;
      NQ_G(g, l, nqi::counts_dev(ix, (const int32_t *)w.allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, nq, nullptr, stride, nullptr, &co));
    }
    // 4. the candidates of all ranks, my partial counts of them, their sums
    {
      nqi::Span sp(g->sh[0], NIQKI_KC_EXCHANGE);
      if ((rc = all_gather(g, &niqki_group::Ws::cand, &niqki_group::Ws::cand_all, blob))) return rc;
    }
    if ((rc = pre_produce(g, &niqki_group::Ws::mine))) return rc;
    for (uint32_t l = 0; l < g->n_local; ++l) {
      niqki_index *ix = g->sh[l];
      NQ_GH(g, hipSetDevice(ix->device));
      auto &w = g->ws[l];
      nqi::Span sp(ix, NIQKI_KC_EXCHANGE);
      NQ_GH(g, hipMemsetAsync(w.flag.p, 0, 4, ix->stream));
      const nq::IdView ids{(const int32_t *)w.cand_all.p, (uint64_t)(blob / 4), C};
      NQ_GH(g, launch_surv_lookup(ix, (const int2 *)w.surv.p, (const int32_t *)w.cand.p + (size_t)nq * C + nq, SC, nq, G * C, ids,
                                  (const int32_t *)w.allsk.p, ix->d.slot_end - ix->d.slot_begin, 0, w.mine.p,
                                  (uint32_t *)w.flag.p, g->wide));
      // a candidate list of ANY rank that overflowed (same words on every rank: all decide alike)
      hipLaunchKernelGGL(nq::blob_overflow_kernel, dim3((nq + 255) / 256), dim3(256), 0, ix->stream, (const int32_t *)w.cand_all.p,
                         (uint64_t)(blob / 4), nq, G, C, SC, (uint32_t *)w.flag.p);
      NQ_GH(g, hipGetLastError());
    }
    {
      nqi::Span sp(g->sh[0], NIQKI_KC_EXCHANGE);
      if ((rc = reduce_scatter_u32(g, &niqki_group::Ws::mine, &niqki_group::Ws::tot, g->wide ? (size_t)per * G * C : (size_t)per * G * C / 2))) return rc;
    }
    // 5. The overflow word stays on the device: the hits are made from the candidates' sums right away, and
    // query_end -- the one place the host waits -- redoes the batch densely in the (rare) case that a list
    // overflowed.
    for (uint32_t l = 0; l < g->n_local; ++l) {
      niqki_index *ix = g->sh[l];
      NQ_GH(g, hipSetDevice(ix->device));
      auto &w = g->ws[l];
      unsigned long long *off = (unsigned long long *)pd.hit_off[l];
      uint32_t *hc = pd.hit_counts[l], *hg = pd.hit_gids[l];
      if (pd.host) {
        NQ_G(g, l, nqi::ensure(ix, w.hitoff, (size_t)(per + 1) * 8));
        NQ_G(g, l, nqi::ensure(ix, w.hc, (size_t)std::max<uint64_t>(capacity, 1) * 4));
        NQ_G(g, l, nqi::ensure(ix, w.hg, (size_t)std::max<uint64_t>(capacity, 1) * 4));
        off = (unsigned long long *)w.hitoff.p; hc = (uint32_t *)w.hc.p; hg = (uint32_t *)w.hg.p;
      }
      const nq::IdView ids{(const int32_t *)w.cand_all.p, (uint64_t)(blob / 4), C};
      NQ_G(g, l, hits_from_candidates(ix, w.tot.p, (g->first + l) * per, per, G * C, ids, min_score, w.keys, w.nhit, off, hc, hg,
                                      capacity, g->wide));
    }
  } else {
    if ((rc = gather_rows(g, nq, N, stride))) return rc;
    if ((rc = finish_dense(g, per, N, stride))) return rc;
    if ((rc = run_hits(g, per, N, stride))) return rc;
  }
  // the two words query_end looks at, on their way to pinned memory behind everything above
  NQ_GH(g, hipSetDevice(g->sh[0]->device));
  g->host_flags[0] = g->host_flags[1] = 0;
  if (sparse) NQ_GH(g, hipMemcpyAsync(&g->host_flags[0], g->ws[0].flag.p, 4, hipMemcpyDeviceToHost, g->sh[0]->stream));
  if (g->transport == niqki_group::kIpc)
    NQ_GH(g, hipMemcpyAsync(&g->host_flags[1], g->ipc.flags + kFlagErr, 4, hipMemcpyDeviceToHost, g->sh[0]->stream));
  NQ_GH(g, hipEventRecord(g->ev_flags, g->sh[0]->stream));
  return NIQKI_OK;
}

int niqki_group_query_end(niqki_group *g) {
  if (!g) return NIQKI_E_INVALID;
  auto &pd = g->pend;
  if (!pd.active) return gfail(g, NIQKI_E_STATE, "no query batch in flight");
  pd.active = false;
  const uint32_t per = pd.per, N = pd.N;
  if (per == 0) return NIQKI_OK;
  NQ_GH(g, hipSetDevice(g->sh[0]->device));
  NQ_GH(g, hipEventSynchronize(g->ev_flags));
  if (g->host_flags[1]) return gfail(g, NIQKI_E_HIP, "a peer of the group did not answer in time (ipc transport)");
  int rc;
  if (pd.sparse && g->host_flags[0]) {
    // a list overflowed on some rank: the batch again with counter rows (the sketch slices are still in ws.allsk)
    ++g->overflows;
    if ((rc = prepare_batch(g, per, N, true, false))) return rc;
    if ((rc = gather_rows(g, g->world * per, N, pd.stride))) return rc;
    if ((rc = finish_dense(g, per, N, pd.stride))) return rc;
    if ((rc = run_hits(g, per, N, pd.stride))) return rc;
  }
  if (!pd.host) return NIQKI_OK;
  for (uint32_t l = 0; l < g->n_local; ++l) {
    niqki_index *ix = g->sh[l];
    NQ_GH(g, hipSetDevice(ix->device));
    auto &w = g->ws[l];
    NQ_GH(g, hipMemcpyAsync(pd.hit_off[l], w.hitoff.p, (size_t)(per + 1) * 8, hipMemcpyDeviceToHost, ix->stream));
    NQ_GH(g, hipStreamSynchronize(ix->stream));
    const uint64_t total = pd.hit_off[l][per];
    if (total > pd.capacity) return gfail(g, NIQKI_E_CAPACITY, "hit capacity too small; hit_off holds the sizes needed");
    if (total) {
      NQ_GH(g, hipMemcpyAsync(pd.hit_counts[l], w.hc.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_GH(g, hipMemcpyAsync(pd.hit_gids[l], w.hg.p, (size_t)total * 4, hipMemcpyDeviceToHost, ix->stream));
      NQ_GH(g, hipStreamSynchronize(ix->stream));
    }
  }
  return NIQKI_OK;
}

int niqki_group_query(niqki_group *g, const int32_t *const *local_sketches, uint32_t per, uint64_t *const *hit_off,
                      uint32_t *const *hit_counts, uint32_t *const *hit_gids, uint64_t capacity, int mem) {
  int rc = niqki_group_query_begin(g, local_sketches, per, hit_off, hit_counts, hit_gids, capacity, mem);
  if (rc) {
    if (g) g->pend.active = false;
    return rc;
  }
  return niqki_group_query_end(g);
}

// ---- the row-free sparse exchange step by step (what a shard of a group runs; device memory only) ----
int niqki_query_survivors(niqki_index *ix, const int32_t *sketches, uint32_t nq, uint32_t cand_threshold, uint32_t surv_threshold,
                          uint32_t cand_cap, uint32_t surv_cap, int32_t *cand, int32_t *n_cand, int32_t *surv, int32_t *n_surv,
                          int mem) {
  if (!ix || (nq && (!sketches || !cand || !n_cand || !surv || !n_surv)) || !cand_cap || !surv_cap) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return nqi::fail(ix, NIQKI_E_INVALID, "niqki_query_survivors is device-memory only");
  NQ_HIP(ix, hipSetDevice(ix->device));
  nq::CandOut co;
  co.cand = cand; co.n = n_cand; co.thr = cand_threshold; co.cap = cand_cap;
  co.surv = (int2 *)surv; co.surv_n = n_surv; co.surv_thr = surv_threshold; co.surv_cap = surv_cap;
  return nqi::counts_dev(ix, sketches, ix->d.F, ix->resident_bytes ? ix->full_begin : ix->d.slot_begin, nq, nullptr, 0, nullptr, &co);
}

int niqki_survivor_counts(niqki_index *ix, const int32_t *sketches, uint32_t nq, const int32_t *ids, uint32_t m, const int32_t *surv,
                          const int32_t *n_surv, uint32_t surv_cap, uint16_t *counts, int mem) {
  if (!ix || (nq && m && (!sketches || !ids || !surv || !n_surv || !counts))) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return nqi::fail(ix, NIQKI_E_INVALID, "niqki_survivor_counts is device-memory only");
  if (ix->resident_bytes) return nqi::fail(ix, NIQKI_E_STATE, "not on a paged index (the sketch store is in host memory)");
  if (m > nq::kCandMax || surv_cap > nq::kSurvMax || !surv_cap) return nqi::fail(ix, NIQKI_E_INVALID, "at most 4096 ids per query and 4096 survivors");
  NQ_HIP(ix, hipSetDevice(ix->device));
  int rc = nqi::ensure(ix, ix->ws_misc, 256);
  if (rc) return rc;
  const nq::IdView view{ids, 0, std::max(m, 1u)};
  NQ_HIP(ix, launch_surv_lookup(ix, (const int2 *)surv, n_surv, surv_cap, nq, m, view, sketches, ix->d.F, ix->d.slot_begin, counts,
                                (uint32_t *)ix->ws_misc.p));
  return NIQKI_OK;
}

int niqki_hits_from_candidates(niqki_index *ix, const int32_t *ids, const uint16_t *totals, uint32_t nq, uint32_t m, uint64_t *hit_off,
                               uint32_t *hit_counts, uint32_t *hit_gids, uint64_t capacity, int mem) {
  if (!ix || !hit_off || (nq && m && (!ids || !totals))) return NIQKI_E_INVALID;
  if (mem != NIQKI_MEM_DEVICE) return nqi::fail(ix, NIQKI_E_INVALID, "niqki_hits_from_candidates is device-memory only");
  NQ_HIP(ix, hipSetDevice(ix->device));
  if (nq == 0 || m == 0) {
    NQ_HIP(ix, hipMemsetAsync(hit_off, 0, (size_t)(nq + 1) * 8, ix->stream));
    return NIQKI_OK;
  }
  const nq::IdView view{ids, 0, m};
  return hits_from_candidates(ix, totals, 0, nq, m, view, ix->d.min_score, ix->ws_tc, ix->ws_tg, (unsigned long long *)hit_off,
                              hit_counts, hit_gids, capacity);
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
