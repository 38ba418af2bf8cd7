// tests/fake_rccl/fake_rccl.hip -- TEST INFRASTRUCTURE ONLY: a stand-in `librccl.so.1` for ranks that all live in
// ONE process (any devices, also one shared device -- which the real RCCL refuses).
//
// Why: nq_group.hip's RCCL branch (ncclSend / ncclRecv pairs inside a group call, ncclAllGather,
// ncclReduceScatter(ncclUint32, ncclSum), one communicator per local rank) can only run with more than one rank on
// a box with more than one GPU, and this pool hands out one-GPU boxes.  With this library first on the loader's
// path (LD_LIBRARY_PATH, tests/test_gpu_fake_rccl.py) every count, offset, datatype, communicator and stream the
// product passes to RCCL at world 2 / 3 / 8 is executed and checked end to end against a whole-range handle and the
// oracle.  Semantics are the ones RCCL documents: operations between ncclGroupStart and ncclGroupEnd are issued
// together at the outermost ncclGroupEnd; a send matches the receive its peer posted for it (in order); a
// collective needs the same call from every rank of the communicator.  Everything is enqueued on the callers'
// streams (hipMemcpyAsync + one summing kernel), ordered across the streams by events, never synchronised.
// The product never references this file: libniqki_hip.so dlopens "librccl.so.1" by name (tests/test_abi.py checks).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

struct ncclComm {
  int rank = 0, world = 0, device = 0;
  std::string uid;
};

namespace {

std::mutex g_m;
std::map<std::string, std::vector<ncclComm *>> g_groups;   // communicators by unique id
std::atomic<uint64_t> g_next_id{1};

enum Kind { kSend, kRecv, kAllGather, kReduceScatter };
struct Op {
  Kind kind;
  ncclComm *comm;
  const void *send = nullptr;
  void *recv = nullptr;
  size_t count = 0;      // elements
  size_t esize = 1;      // bytes per element
  int peer = -1;
  hipStream_t stream = nullptr;
  bool done = false;
};
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;
thread_local ncclResult_t t_err = ncclSuccess;

// counters a test reads back (fake_rccl_stat): proof that THIS library served the calls
std::atomic<uint64_t> n_init{0}, n_send{0}, n_recv{0}, n_allgather{0}, n_reduce_scatter{0}, n_groups{0}, n_bytes{0};
std::atomic<int64_t> fail_send_after{-1};   // fake_rccl_fail_send(n): the n-th ncclSend from now on fails (0 = the next)

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    case ncclFloat16: return 2;
    default: return 0;
  }
}

struct SumSrc { const uint32_t *p[64]; };
__global__ void fake_sum_u32(SumSrc src, int world, size_t off, size_t n, uint32_t *out) {
  const size_t step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += step) {
    uint32_t s = 0;
    for (int r = 0; r < world; ++r) s += src.p[r][off + i];
    out[i] = s;
  }
}

// every stream of `ops` waits for everything enqueued so far on all of them
hipError_t cross_wait(const std::vector<Op> &ops) {
  std::vector<std::pair<hipStream_t, int>> streams;
  for (const Op &o : ops) {
    bool seen = false;
    for (auto &s : streams) seen |= s.first == o.stream;
    if (!seen) streams.push_back({o.stream, o.comm->device});
  }
  if (streams.size() < 2) return hipSuccess;
  std::vector<hipEvent_t> ev(streams.size());
  hipError_t e = hipSuccess;
  for (size_t i = 0; i < streams.size() && e == hipSuccess; ++i) {
    e = hipSetDevice(streams[i].second);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventRecord(ev[i], streams[i].first);
  }
  for (size_t i = 0; i < streams.size() && e == hipSuccess; ++i) {
    e = hipSetDevice(streams[i].second);
    for (size_t j = 0; j < streams.size() && e == hipSuccess; ++j)
      if (j != i) e = hipStreamWaitEvent(streams[i].first, ev[j], 0);
  }
  for (hipEvent_t x : ev)
    if (x) (void)hipEventDestroy(x);   // (destruction is deferred until the recorded work has passed)
  return e;
}

ncclResult_t flush() {
  std::vector<Op> ops;
  ops.swap(t_ops);
  if (t_err != ncclSuccess) { const ncclResult_t r = t_err; t_err = ncclSuccess; return r; }
  if (ops.empty()) return ncclSuccess;
  std::lock_guard<std::mutex> lk(g_m);
  int dev0 = 0;
  (void)hipGetDevice(&dev0);
  if (cross_wait(ops) != hipSuccess) return ncclUnhandledCudaError;
  ncclResult_t rc = ncclSuccess;
  for (size_t i = 0; i < ops.size() && rc == ncclSuccess; ++i) {
    Op &o = ops[i];
    if (o.done) continue;
    if (hipSetDevice(o.comm->device) != hipSuccess) { rc = ncclUnhandledCudaError; break; }
    if (o.kind == kRecv) {
      // the first unmatched send of rank `peer` addressed to me
      Op *s = nullptr;
      for (Op &c : ops)
        if (!c.done && c.kind == kSend && c.comm->uid == o.comm->uid && c.comm->rank == o.peer && c.peer == o.comm->rank) { s = &c; break; }
      if (!s) { rc = ncclInvalidUsage; break; }   // (all ranks live in this process: its send must be in this group call)
      if (s->count * s->esize != o.count * o.esize) { rc = ncclInvalidArgument; break; }
      if (o.count && hipMemcpyAsync(o.recv, s->send, o.count * o.esize, hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
      n_bytes += o.count * o.esize;
      s->done = o.done = true;
    } else if (o.kind == kAllGather) {
      for (int r = 0; r < o.comm->world && rc == ncclSuccess; ++r) {
        const Op *src = nullptr;
        for (const Op &c : ops)
          if (c.kind == kAllGather && c.comm->uid == o.comm->uid && c.comm->rank == r && c.count * c.esize == o.count * o.esize) { src = &c; break; }
        if (!src) { rc = ncclInvalidUsage; break; }
        if (o.count && hipMemcpyAsync((char *)o.recv + (size_t)r * o.count * o.esize, src->send, o.count * o.esize,
                                      hipMemcpyDeviceToDevice, o.stream) != hipSuccess) rc = ncclUnhandledCudaError;
        n_bytes += o.count * o.esize;
      }
      o.done = true;
    } else if (o.kind == kReduceScatter) {
      if (o.comm->world > 64) { rc = ncclInvalidArgument; break; }
      SumSrc src{};
      for (int r = 0; r < o.comm->world; ++r) {
        const Op *p = nullptr;
        for (const Op &c : ops)
          if (c.kind == kReduceScatter && c.comm->uid == o.comm->uid && c.comm->rank == r && c.count == o.count) { p = &c; break; }
        if (!p) { rc = ncclInvalidUsage; break; }
        src.p[r] = (const uint32_t *)p->send;
      }
      if (rc != ncclSuccess) break;
      if (o.count) {
        const unsigned blocks = (unsigned)std::min<size_t>((o.count + 255) / 256, 4096);
        hipLaunchKernelGGL(fake_sum_u32, dim3(blocks), dim3(256), 0, o.stream, src, o.comm->world, (size_t)o.comm->rank * o.count, o.count,
                           (uint32_t *)o.recv);
        if (hipGetLastError() != hipSuccess) rc = ncclUnhandledCudaError;
      }
      n_bytes += o.count * 4 * (size_t)o.comm->world;
      o.done = true;
    }
  }
  if (rc == ncclSuccess)
    for (const Op &o : ops)
      if (!o.done) { rc = ncclInvalidUsage; break; }   // a send nobody received
  if (rc == ncclSuccess && cross_wait(ops) != hipSuccess) rc = ncclUnhandledCudaError;   // senders may reuse their buffers
  (void)hipSetDevice(dev0);
  return rc;
}

ncclResult_t post(Op o) {
  if (!o.comm) return ncclInvalidArgument;
  t_ops.push_back(o);
  if (t_depth == 0) return flush();
  return ncclSuccess;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof *id);
  const uint64_t v = g_next_id++;
  std::memcpy(id->internal, "FAKERCCL", 8);
  std::memcpy(id->internal + 8, &v, 8);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ncclComm *c = new ncclComm();
  c->rank = rank;
  c->world = nranks;
  c->uid.assign(id.internal, id.internal + sizeof id.internal);
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return ncclUnhandledCudaError; }
  std::lock_guard<std::mutex> lk(g_m);
  auto &v = g_groups[c->uid];
  for (ncclComm *p : v)
    if (p->rank == rank) { delete c; return ncclInvalidUsage; }
  v.push_back(c);
  ++n_init;
  *comm = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  if (!comm) return ncclSuccess;
  std::lock_guard<std::mutex> lk(g_m);
  auto it = g_groups.find(comm->uid);
  if (it != g_groups.end()) {
    auto &v = it->second;
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i] == comm) { v.erase(v.begin() + i); break; }
    if (v.empty()) g_groups.erase(it);
  }
  delete comm;
  return ncclSuccess;
}

// how many ranks of the communicator this library has really seen join (all of them live in this process):
// a communicator that was told "world 8" but only met 5 ranks says 5
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
  if (!comm || !count) return ncclInvalidArgument;
  std::lock_guard<std::mutex> lk(g_m);
  auto it = g_groups.find(comm->uid);
  *count = it == g_groups.end() ? 0 : (int)it->second.size();
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *rank) {
  if (!comm || !rank) return ncclInvalidArgument;
  *rank = comm->rank;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  ++t_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_depth == 0) return ncclInvalidUsage;
  if (--t_depth) return ncclSuccess;
  ++n_groups;
  return flush();
}

ncclResult_t ncclSend(const void *sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || peer < 0 || peer >= comm->world || !type_size(datatype) || (count && !sendbuff)) return ncclInvalidArgument;
  if (fail_send_after.load() >= 0 && fail_send_after.fetch_sub(1) == 0) {
    t_err = ncclInternalError;   // (what RCCL does: the group call reports the failure, at the latest from ncclGroupEnd)
    return ncclInternalError;
  }
  ++n_send;
  Op o{kSend, comm};
  o.send = sendbuff; o.count = count; o.esize = type_size(datatype); o.peer = peer; o.stream = stream;
  return post(o);
}

ncclResult_t ncclRecv(void *recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  if (!comm || peer < 0 || peer >= comm->world || !type_size(datatype) || (count && !recvbuff)) return ncclInvalidArgument;
  ++n_recv;
  Op o{kRecv, comm};
  o.recv = recvbuff; o.count = count; o.esize = type_size(datatype); o.peer = peer; o.stream = stream;
  return post(o);
}

ncclResult_t ncclAllGather(const void *sendbuff, void *recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm,
                           hipStream_t stream) {
  if (!comm || !type_size(datatype) || (sendcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
  ++n_allgather;
  Op o{kAllGather, comm};
  o.send = sendbuff; o.recv = recvbuff; o.count = sendcount; o.esize = type_size(datatype); o.stream = stream;
  return post(o);
}

ncclResult_t ncclReduceScatter(const void *sendbuff, void *recvbuff, size_t recvcount, ncclDataType_t datatype, ncclRedOp_t op,
                               ncclComm_t comm, hipStream_t stream) {
  if (!comm || (recvcount && (!sendbuff || !recvbuff))) return ncclInvalidArgument;
  if (datatype != ncclUint32 || op != ncclSum) return ncclInvalidArgument;   // (all the product asks for)
  ++n_reduce_scatter;
  Op o{kReduceScatter, comm};
  o.send = sendbuff; o.recv = recvbuff; o.count = recvcount; o.esize = 4; o.stream = stream;
  return post(o);
}

const char *ncclGetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "unhandled cuda error (fake rccl)";
    case ncclInternalError: return "internal error (fake rccl)";
    case ncclInvalidArgument: return "invalid argument (fake rccl)";
    case ncclInvalidUsage: return "invalid usage (fake rccl)";
    default: return "error (fake rccl)";
  }
}

// ---- what the tests read back ----
// what: 0 communicators made, 1 sends, 2 receives, 3 all-gathers, 4 reduce-scatters, 5 outermost group calls closed,
//       6 bytes moved, 7 the calling thread's open group depth (0 after every product call, also a failed one),
//       8 communicators alive
uint64_t fake_rccl_stat(int what) {
  switch (what) {
    case 0: return n_init;
    case 1: return n_send;
    case 2: return n_recv;
    case 3: return n_allgather;
    case 4: return n_reduce_scatter;
    case 5: return n_groups;
    case 6: return n_bytes;
    case 7: return (uint64_t)t_depth;
    case 8: {
      std::lock_guard<std::mutex> lk(g_m);
      uint64_t n = 0;
      for (auto &kv : g_groups) n += kv.second.size();
      return n;
    }
    default: return 0;
  }
}
void fake_rccl_fail_send(int64_t after) { fail_send_after = after; }

}  // extern "C"
