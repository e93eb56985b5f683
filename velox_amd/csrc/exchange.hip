// Multi-GPU exchange behind the C ABI: RCCL communicators owned by
// libvx355 and the two collectives the hot path needs —
//   * the repartitioned join's exchange: every rank sends slice p of each column straight
//     to rank p: ncclGroupStart(); ncclSend / ncclRecv per peer and column; ncclGroupEnd()
//     (xGMI is point to point: 7 links per GPU, every slice rides its own link; a ring
//     would be per-link bound), preceded by an all-gather of the slice sizes;
//   * the partial -> final merge of a row-sharded aggregation: an all-gather of the
//     ranks' (small) partial results.
// A communicator is bound to one GPU. One process can own communicators for all the GPUs
// of a node (vx355_comm_create_all: "one process drives 8 GPUs") or one per process
// (vx355_comm_create: "one process per GPU", the id travels out of band).
//
// librccl is loaded at first use from the directory of the HIP runtime this library is
// linked against (a process may hold a second HIP runtime / RCCL pair, e.g. PyTorch's
// bundled one: streams of one runtime mean nothing to the other's RCCL).
#include "common.h"
#include "transport.h"

#include <dlfcn.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>

namespace vx {
namespace {

// (the part of rccl.h the exchange uses - RCCL keeps NCCL's ABI - is declared in transport.h)
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*CommCount)(ncclComm_t, int*) = nullptr;      // optional: what the communicator itself reports
  int (*CommUserRank)(ncclComm_t, int*) = nullptr;
  int (*CommCuDevice)(ncclComm_t, int*) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string path;
};

std::mutex gRcclMutex;
Rccl gRccl;

Rccl& rccl() {
  std::lock_guard<std::mutex> lock(gRcclMutex);
  if (gRccl.lib) {
    return gRccl;
  }
  if (const char* e = std::getenv("VX355_COMM_TRANSPORT")) {
    if (std::string(e) == "shm") {
      // ranks that share one GPU (RCCL refuses two ranks per device): the same table, served through
      // host shared memory (shm_transport.hip), CommInitAll - one process, one rank per entry of the
      // device list - included.
      Rccl r;
      r.lib = reinterpret_cast<void*>(&gRccl);
      r.path = "shm transport (VX355_COMM_TRANSPORT=shm)";
      r.GetUniqueId = shmx::GetUniqueId;
      r.CommInitRank = shmx::CommInitRank;
      r.CommInitAll = shmx::CommInitAll;
      r.CommDestroy = shmx::CommDestroy;
      r.Send = shmx::Send;
      r.Recv = shmx::Recv;
      r.AllGather = shmx::AllGather;
      r.GroupStart = shmx::GroupStart;
      r.GroupEnd = shmx::GroupEnd;
      r.CommCount = shmx::CommCount;
      r.CommUserRank = shmx::CommUserRank;
      r.CommCuDevice = shmx::CommCuDevice;
      r.GetErrorString = shmx::GetErrorString;
      gRccl = r;
      return gRccl;
    } else if (std::string(e) != "rccl" && e[0] != 0) {
      VX_THROW(VX355_EINVAL, "VX355_COMM_TRANSPORT is 'rccl' (default) or 'shm'");
    }
  }
  std::vector<std::string> candidates;
  if (const char* e = std::getenv("VX355_RCCL_PATH")) {
    candidates.push_back(e);
  }
  Dl_info info;
  if (dladdr(reinterpret_cast<const void*>(&hipStreamCreateWithFlags), &info) && info.dli_fname) {
    std::string dir = info.dli_fname;
    const size_t slash = dir.rfind('/');
    if (slash != std::string::npos) {
      candidates.push_back(dir.substr(0, slash) + "/librccl.so.1");
      candidates.push_back(dir.substr(0, slash) + "/librccl.so");
    }
  }
  candidates.push_back("/opt/rocm/lib/librccl.so.1");
  candidates.push_back("librccl.so.1");
  std::string tried;
  for (const auto& c : candidates) {
    void* h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      tried += c + " ";
      continue;
    }
    Rccl r;
    r.lib = h;
    r.path = c;
    auto sym = [&](const char* name) { return dlsym(h, name); };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
    r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
    r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.CommCount = reinterpret_cast<decltype(r.CommCount)>(sym("ncclCommCount"));
    r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(sym("ncclCommUserRank"));
    r.CommCuDevice = reinterpret_cast<decltype(r.CommCuDevice)>(sym("ncclCommCuDevice"));
    if (r.GetUniqueId && r.CommInitRank && r.CommInitAll && r.CommDestroy && r.Send && r.Recv && r.AllGather &&
        r.GroupStart && r.GroupEnd) {
      gRccl = r;
      return gRccl;
    }
    dlclose(h);
    tried += c + "(symbols missing) ";
  }
  VX_THROW(VX355_EUNSUPPORTED, "librccl not found (tried: " + tried + "); set VX355_RCCL_PATH");
}

void ncclOk(int rc, const char* what) {
  if (rc != kNcclSuccess) {
    Rccl& r = gRccl;
    VX_THROW(VX355_EINTERNAL, std::string("RCCL: ") + what + " failed: " +
                                  (r.GetErrorString ? r.GetErrorString(rc) : std::to_string(rc).c_str()));
  }
}

// RCCL prints a version banner on stdout when the first communicator is created (C stdio,
// flushed at exit: it lands behind whatever the host printed last). A library has no business
// writing to the host's stdout: the banner is sent to stderr instead.
class StdoutToStderr {
 public:
  StdoutToStderr() {
    fflush(stdout);
    saved_ = dup(1);
    if (saved_ >= 0) {
      dup2(2, 1);
    }
  }
  ~StdoutToStderr() {
    fflush(stdout);
    if (saved_ >= 0) {
      dup2(saved_, 1);
      close(saved_);
    }
  }

 private:
  int saved_ = -1;
};

// ncclGroupStart ... ncclGroupEnd that cannot be left open: an exception between the two would
// leave the thread's RCCL group open and the next collective would silently join it.
class GroupGuard {
 public:
  explicit GroupGuard(Rccl& r) : r_(r) { ncclOk(r_.GroupStart(), "ncclGroupStart"); }
  void end() {
    open_ = false;
    ncclOk(r_.GroupEnd(), "ncclGroupEnd");
  }
  ~GroupGuard() {
    if (open_) {
      (void)r_.GroupEnd();
    }
  }

 private:
  Rccl& r_;
  bool open_ = true;
};

}  // namespace
}  // namespace vx

using namespace vx;

struct vx355_comm {
  vx::Runtime* ctx = nullptr;  // execution context on the communicator's GPU
  void* comm = nullptr;        // ncclComm_t
  int32_t world = 1;
  int32_t rank = 0;
  // true: every collective of this communicator goes through RCCL, the rank's own slice included
  // (a send / recv pair to itself inside the group). Always so for world > 1 peers; a one-rank
  // communicator only with VX355_COMM_FORCE_RCCL=1 - the way to execute ncclCommInitRank, the
  // grouped send / recv, the 256 MiB message cutting and the receive-slot pipeline on a 1-GPU box.
  bool selfViaRccl = false;
  bool viaRccl() const { return world > 1 || selfViaRccl; }
  DevBuf countsDev;            // all-gather of the slice sizes
};

namespace vx {
bool commViaRccl(const vx355_comm* c) { return c->viaRccl(); }
Runtime* commContext(const vx355_comm* c) { return c ? c->ctx : nullptr; }
}  // namespace vx

namespace vx {
namespace {

// ---- PartitionedOutput -> Exchange edge of a repartitioned join -----------------------------
// (exec/PartitionedOutput.cpp:59-133 + exec/HashPartitionFunction.cpp:76-118 on the sending
// side, exec/Exchange.cpp on the receiving side; here both halves of the edge on one handle.)
constexpr int kExMaxCols = 16;
constexpr int kExMaxKeys = 8;
constexpr int kExSlots = 3;   // two exchanges in flight + the one the caller is consuming

struct HashPartArgs {
  ColView keys[kExMaxKeys];
  int32_t numKeys;
  int32_t kind;            // VX355_PART_MODULO / VX355_PART_BIT_RANGE
  uint32_t numPartitions;
  int32_t bitBegin;
  uint64_t mask;
  int64_t numRows;
  uint32_t* out;
};

// VectorHasher::hash over the key columns (hashValueAt + hashMix, nulls = kNullHash) and
// HashPartitionFunction::partition of that hash, fused: the 8-byte hashes never reach HBM.
__global__ __launch_bounds__(256) void k_hash_partition(HashPartArgs a) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  for (int64_t row = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; row < a.numRows; row += stride) {
    uint64_t h = 0;
    for (int k = 0; k < a.numKeys; ++k) {
      const ColView& c = a.keys[k];
      const uint64_t hv = colIsNull(c, row) ? kNullHash : hashValueAt(c, colIndex(c, row));
      h = k == 0 ? hv : hashMix(h, hv);
    }
    a.out[row] = a.kind == VX355_PART_MODULO ? static_cast<uint32_t>(h % a.numPartitions)
                                             : static_cast<uint32_t>((h >> a.bitBegin) & a.mask);
  }
}

// Sets *flag when a StringView column holds a non-inline string (size > 12: the view carries a
// pointer that means nothing on another GPU).
__global__ __launch_bounds__(256) void k_any_long_string(const uint4* views, int64_t n, uint32_t* flag) {
  const int64_t stride = static_cast<int64_t>(gridDim.x) * blockDim.x;
  bool any = false;
  for (int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += stride) {
    any = any || views[i].x > 12u;
  }
  if (any) {
    *flag = 1;
  }
}

struct ExSlot {
  std::vector<DevBuf> grouped;   // rows grouped by destination (send side)
  std::vector<DevBuf> received;  // rows of every source rank, rank order
  std::vector<int64_t> sendCounts, recvCounts;
  int64_t rows = 0;              // received rows
  hipEvent_t done = nullptr;     // recorded on the payload stream behind the last send / recv
  bool inFlight = false;
};

}  // namespace
}  // namespace vx

struct vx355_exchange {
  vx::Runtime* ctx = nullptr;
  vx355_comm* comm = nullptr;
  std::vector<int32_t> types, widths, keyCols;
  hipStream_t payload = nullptr;     // the RCCL sends / recvs of the column slices run here, so that
                                     // an entry point returns while they are still on the links
  hipEvent_t staged = nullptr;       // grouped columns complete (recorded on ctx->stream)
  vx::ExSlot slots[vx::kExSlots];
  int64_t sent = 0, receivedCount = 0;
  vx::DevBuf parts;
  bool topBits = false;              // VX355_EXCHANGE_TOP_BITS=1 (see destinationArgs)
};


namespace vx {
namespace {
// HashPartitionFunction of the edge for 'world' destinations: hash % world, what the reference's
// PartitionedOutput computes without a HashBitRange (exec/HashPartitionFunction.cpp:112-115) - a
// shim that precomputes partitions, or a CPU Velox peer on the same edge, sends a row to the same
// rank. (The tables behind the edge do not index with the hash's low bits: normalized keys go
// through twang_mix64, generic-mode slots through slotOfHash.) VX355_EXCHANGE_TOP_BITS=1 selects
// the HashBitRange flavour for power-of-two worlds: the top log2(world) bits.
HashPartArgs destinationArgs(const vx355_exchange& x, const DeviceBatch& db, uint32_t world, int64_t n) {
  HashPartArgs a{};
  for (size_t k = 0; k < x.keyCols.size(); ++k) {
    a.keys[k] = db.col(x.keyCols[k]);
  }
  a.numKeys = static_cast<int32_t>(x.keyCols.size());
  if (x.topBits && world > 1 && (world & (world - 1)) == 0) {
    int bits = 0;
    while ((1u << bits) < world) {
      ++bits;
    }
    a.kind = VX355_PART_BIT_RANGE;
    a.bitBegin = 64 - bits;
    a.mask = world - 1;
  } else {
    a.kind = VX355_PART_MODULO;
  }
  a.numPartitions = world;
  a.numRows = n;
  return a;
}
}  // namespace
}  // namespace vx

extern "C" {

int vx355_comm_get_unique_id(void* id_out) {
  try {
    VX_CHECK_ARG(id_out, "NULL argument");
    ncclUniqueId id;
    {
      StdoutToStderr quiet;
      ncclOk(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    }
    std::memcpy(id_out, id.internal, kUniqueIdBytes);
  VX_API_CATCH
}

int vx355_comm_create(const void* id, int32_t world, int32_t rank, vx355_comm** out) {
  VX_API_BEGIN
  VX_CHECK_ARG(id && out && world >= 1 && rank >= 0 && rank < world, "bad communicator arguments");
  auto c = std::make_unique<vx355_comm>();
  c->world = world;
  c->rank = rank;
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, kUniqueIdBytes);
  if (const char* e = std::getenv("VX355_COMM_FORCE_RCCL")) {
    c->selfViaRccl = std::atoi(e) != 0;
  }
  if (c->viaRccl()) {
    // (A one-rank communicator needs no RCCL: every collective is a device copy. Creating one
    // anyway is not free on this stack: after ncclCommInitRank the random-access kernels of the
    // same process ran 1.7x slower - k_join_probe 3.4 -> 5.8 ms, k_gather_deps 4.0 -> 7.1 ms on
    // the config-5 shape, profiles/r03_c5_rccl_init_effect.txt.)
    StdoutToStderr quiet;
    ncclOk(rccl().CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank");
  }
  try {
    c->ctx = Runtime::createContext();
  } catch (...) {
    (void)rccl().CommDestroy(c->comm);
    throw;
  }
  *out = c.release();
  VX_API_END
}

int vx355_comm_create_all(int32_t num_devices, const int32_t* devices, vx355_comm** out) {
  try {
    VX_CHECK_ARG(num_devices >= 1 && devices && out, "bad communicator arguments");
    std::vector<int> devs(devices, devices + num_devices);
    std::vector<void*> comms(num_devices, nullptr);
    for (int d : devs) {
      (void)Runtime::defaultContext(d);  // throws unless vx355_init(d) was called
    }
    if (num_devices > 1) {
      StdoutToStderr quiet;
      ncclOk(rccl().CommInitAll(comms.data(), num_devices, devs.data()), "ncclCommInitAll");
    }
    std::vector<std::unique_ptr<vx355_comm>> made;
    try {
      for (int32_t i = 0; i < num_devices; ++i) {
        vx::ContextScope scope(Runtime::defaultContext(devs[i]));  // re-binds the thread: RCCL moved it
        auto c = std::make_unique<vx355_comm>();
        c->world = num_devices;
        c->rank = i;
        c->comm = comms[i];
        c->ctx = Runtime::createContext();
        made.push_back(std::move(c));
      }
    } catch (...) {
      for (auto& c : made) {
        Runtime::destroyContext(c->ctx);
      }
      for (void* raw : comms) {
        if (raw) {
          (void)rccl().CommDestroy(raw);
        }
      }
      throw;
    }
    for (int32_t i = 0; i < num_devices; ++i) {
      out[i] = made[i].release();
    }
  VX_API_CATCH
}

// world / rank / device as the communicator itself reports them (ncclCommCount /
// ncclCommUserRank / ncclCommCuDevice), not as the caller said at creation.
int vx355_comm_info(const vx355_comm* c, int32_t* world, int32_t* rank, int32_t* device) {
  try {
    VX_CHECK_ARG(c, "NULL argument");
    Rccl& r = rccl();
    int w = c->world, k = c->rank, d = c->ctx->device;
    if (c->comm && r.CommCount && r.CommUserRank && r.CommCuDevice) {
      ncclOk(r.CommCount(c->comm, &w), "ncclCommCount");
      ncclOk(r.CommUserRank(c->comm, &k), "ncclCommUserRank");
      ncclOk(r.CommCuDevice(c->comm, &d), "ncclCommCuDevice");
    }
    if (world) {
      *world = w;
    }
    if (rank) {
      *rank = k;
    }
    if (device) {
      *device = d;
    }
  VX_API_CATCH
}

void* vx355_comm_stream(vx355_comm* c) { return c ? static_cast<void*>(c->ctx->stream) : nullptr; }

void vx355_comm_destroy(vx355_comm* c) {
  if (!c) {
    return;
  }
  Runtime* ctx = c->ctx;
  try {
    vx::ContextScope scope(ctx);
    if (c->comm) {
      (void)rccl().CommDestroy(c->comm);
    }
    delete c;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
}

}  // extern "C"

namespace vx {
namespace {

// recv[s] = rows rank s holds for this rank: one all-gather of 'world' int64 per rank on the
// current context's stream.
void exchangeCounts(vx355_comm* c, const int64_t* send, int64_t* recv) {
  auto& rt = Runtime::get();
  const size_t w = static_cast<size_t>(c->world);
  if (!c->viaRccl()) {
    recv[0] = send[0];
    return;
  }
  int64_t* dev = static_cast<int64_t*>(c->countsDev.ensure((w + w * w) * 8 + 64));
  copyIn(dev, send, VX355_MEM_HOST, w * 8);
  ncclOk(rccl().AllGather(dev, dev + w, w, kNcclInt64, c->comm, rt.stream), "ncclAllGather(counts)");
  std::vector<int64_t> all(w * w);
  copyOut(all.data(), VX355_MEM_HOST, dev + w, w * w * 8);
  for (size_t s = 0; s < w; ++s) {
    recv[s] = all[s * w + static_cast<size_t>(c->rank)];
  }
}

// Posts the slices of every column on 'stream': slice p of the send buffer goes straight to rank
// p, the slices of all ranks arrive in rank order; the rank's own slice is a device copy.
// Messages are cut into pieces of at most kMaxMessageBytes (both ends cut the same byte count the
// same way): RCCL 2.26.6 as shipped with this image delivers only the first half of a message above
// 2^30 bytes (tools/torch_a2a_repro.py shows it through torch.distributed on one rank;
// profiles/r03_rccl_large_message.txt).
constexpr int64_t kMaxMessageBytes = 256LL << 20;

void sendBytes(Rccl& r, vx355_comm* c, hipStream_t stream, const char* src, int64_t bytes, int32_t peer) {
  for (int64_t at = 0; at < bytes; at += kMaxMessageBytes) {
    ncclOk(r.Send(src + at, static_cast<size_t>(std::min(kMaxMessageBytes, bytes - at)), kNcclUint8, peer, c->comm,
                  stream),
           "ncclSend");
  }
}

void recvBytes(Rccl& r, vx355_comm* c, hipStream_t stream, char* dst, int64_t bytes, int32_t peer) {
  for (int64_t at = 0; at < bytes; at += kMaxMessageBytes) {
    ncclOk(r.Recv(dst + at, static_cast<size_t>(std::min(kMaxMessageBytes, bytes - at)), kNcclUint8, peer, c->comm,
                  stream),
           "ncclRecv");
  }
}

void postColumns(vx355_comm* c, hipStream_t stream, const void* const* sendCols, const int32_t* widths, int32_t numCols,
                 const int64_t* sendCounts, const int64_t* recvCounts, void* const* recvCols) {
  for (int32_t col = 0; col < numCols; ++col) {
    VX_CHECK_ARG(widths[col] >= 1, "column width");
  }
  int64_t totalSend = 0, totalRecv = 0;
  for (int32_t peer = 0; peer < c->world; ++peer) {
    VX_CHECK_ARG(sendCounts[peer] >= 0 && recvCounts[peer] >= 0, "negative slice size");
    totalSend += sendCounts[peer];
    totalRecv += recvCounts[peer];
  }
  for (int32_t col = 0; col < numCols; ++col) {
    VX_CHECK_ARG((totalSend == 0 || sendCols[col]) && (totalRecv == 0 || recvCols[col]), "NULL column");
  }
  VX_CHECK_ARG(sendCounts[c->rank] == recvCounts[c->rank], "a rank's slice for itself has one size");
  Rccl* r = c->viaRccl() ? &rccl() : nullptr;
  // Grouped point-to-point: all slices of all columns are posted together so that the seven
  // outgoing links of the GPU work concurrently.
  std::unique_ptr<GroupGuard> group;
  if (r) {
    group = std::make_unique<GroupGuard>(*r);
  }
  for (int32_t col = 0; col < numCols; ++col) {
    const int64_t w = widths[col];
    int64_t sendAt = 0, recvAt = 0;
    for (int32_t peer = 0; peer < c->world; ++peer) {
      const int64_t ns = sendCounts[peer], nr = recvCounts[peer];
      const char* src = static_cast<const char*>(sendCols[col]) + sendAt * w;
      char* dst = static_cast<char*>(recvCols[col]) + recvAt * w;
      if (peer == c->rank && !c->selfViaRccl) {
        if (ns > 0) {
          HIP_OK(hipMemcpyAsync(dst, src, static_cast<size_t>(ns * w), hipMemcpyDeviceToDevice, stream));
        }
      } else {
        sendBytes(*r, c, stream, src, ns * w, peer);
        recvBytes(*r, c, stream, dst, nr * w, peer);
      }
      sendAt += ns;
      recvAt += nr;
    }
  }
  if (group) {
    group->end();
  }
}

}  // namespace
}  // namespace vx

extern "C" {

int vx355_exchange_counts(vx355_comm* c, const int64_t* send_counts, int64_t* recv_counts) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send_counts && recv_counts, "NULL argument");
  exchangeCounts(c, send_counts, recv_counts);
  VX_API_END
}

int vx355_exchange_columns(vx355_comm* c, const void* const* send_cols, const int32_t* widths, int32_t num_cols,
                           const int64_t* send_counts, const int64_t* recv_counts, void* const* recv_cols) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send_counts && recv_counts && num_cols >= 0, "bad argument");
  VX_CHECK_ARG(num_cols == 0 || (send_cols && widths && recv_cols), "NULL column arrays");
  auto& rt = Runtime::get();
  postColumns(c, rt.stream, send_cols, widths, num_cols, send_counts, recv_counts, recv_cols);
  rt.sync();
  VX_API_END
}

int vx355_all_gather(vx355_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send && recv, "NULL argument");
  auto& rt = Runtime::get();
  if (bytes_per_rank) {
    if (!c->viaRccl()) {
      HIP_OK(hipMemcpyAsync(recv, send, bytes_per_rank, hipMemcpyDeviceToDevice, rt.stream));
    } else if (static_cast<int64_t>(bytes_per_rank) <= kMaxMessageBytes) {
      ncclOk(rccl().AllGather(send, recv, bytes_per_rank, kNcclInt8, c->comm, rt.stream), "ncclAllGather");
    } else {
      // large blocks: as cut point-to-point messages (see kMaxMessageBytes)
      Rccl& r = rccl();
      GroupGuard group(r);
      for (int32_t peer = 0; peer < c->world; ++peer) {
        char* dst = static_cast<char*>(recv) + static_cast<size_t>(peer) * bytes_per_rank;
        if (peer == c->rank && !c->selfViaRccl) {
          HIP_OK(hipMemcpyAsync(dst, send, bytes_per_rank, hipMemcpyDeviceToDevice, rt.stream));
        } else {
          sendBytes(r, c, rt.stream, static_cast<const char*>(send), static_cast<int64_t>(bytes_per_rank), peer);
          recvBytes(r, c, rt.stream, dst, static_cast<int64_t>(bytes_per_rank), peer);
        }
      }
      group.end();
    }
  }
  rt.sync();
  VX_API_END
}

// All-gather of blocks of different sizes: sizes[s] bytes arrive from rank s (sizes[rank] = this
// rank's block), back to back in rank order. Grouped point-to-point like the column exchange.
int vx355_all_gather_v(vx355_comm* c, const void* send, const int64_t* sizes, void* recv) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && sizes, "NULL argument");
  auto& rt = Runtime::get();
  int64_t total = 0;
  for (int32_t p = 0; p < c->world; ++p) {
    VX_CHECK_ARG(sizes[p] >= 0, "negative block size");
    total += sizes[p];
  }
  const int64_t mine = sizes[c->rank];
  VX_CHECK_ARG((mine == 0 || send) && (total == 0 || recv), "NULL buffer");
  Rccl* r = c->viaRccl() ? &rccl() : nullptr;
  std::unique_ptr<GroupGuard> group;
  if (r) {
    group = std::make_unique<GroupGuard>(*r);
  }
  int64_t at = 0;
  for (int32_t peer = 0; peer < c->world; ++peer) {
    char* dst = static_cast<char*>(recv) + at;
    if (peer == c->rank && !c->selfViaRccl) {
      if (mine > 0) {
        HIP_OK(hipMemcpyAsync(dst, send, static_cast<size_t>(mine), hipMemcpyDeviceToDevice, rt.stream));
      }
    } else {
      sendBytes(*r, c, rt.stream, static_cast<const char*>(send), mine, peer);
      recvBytes(*r, c, rt.stream, dst, sizes[peer], peer);
    }
    at += sizes[peer];
  }
  if (group) {
    group->end();
  }
  rt.sync();
  VX_API_END
}

// ---- the exchange edge ---------------------------------------------------------------------

int vx355_exchange_create(vx355_comm* c, const int32_t* col_types, int32_t num_cols, const int32_t* key_cols,
                          int32_t num_keys, vx355_exchange** out) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && col_types && key_cols && out, "NULL argument");
  VX_CHECK_ARG(num_cols >= 1 && num_cols <= kExMaxCols, "1..16 columns");
  VX_CHECK_ARG(num_keys >= 1 && num_keys <= kExMaxKeys, "1..8 partitioning keys");
  auto x = std::make_unique<vx355_exchange>();
  x->comm = c;
  for (int32_t i = 0; i < num_cols; ++i) {
    const int w = kindWidth(col_types[i]);
    if (w <= 0) {
      // BOOLEAN columns are bit packed: no per-row slice to send
      VX_THROW(VX355_EUNSUPPORTED, "exchange column of type kind " + std::to_string(col_types[i]));
    }
    x->types.push_back(col_types[i]);
    x->widths.push_back(w);
  }
  for (int32_t k = 0; k < num_keys; ++k) {
    VX_CHECK_ARG(key_cols[k] >= 0 && key_cols[k] < num_cols, "partitioning key column");
    x->keyCols.push_back(key_cols[k]);
  }
  if (const char* e = std::getenv("VX355_EXCHANGE_TOP_BITS")) {
    x->topBits = std::atoi(e) != 0;
  }
  x->ctx = Runtime::createContext();
  {
    vx::ContextScope scope(x->ctx);
    HIP_OK(hipStreamCreateWithFlags(&x->payload, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&x->staged, hipEventDisableTiming));
    for (auto& s : x->slots) {
      HIP_OK(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    }
  }
  *out = x.release();
  VX_API_END
}

int vx355_exchange_send(vx355_exchange* x, const vx355_batch* batch) {
  VX_API_BEGIN_CTX(VX_CTX_OF(x))
  VX_CHECK_ARG(x && batch, "NULL argument");
  auto& rt = Runtime::get();
  vx355_comm* c = x->comm;
  const int32_t numCols = static_cast<int32_t>(x->types.size());
  VX_CHECK_ARG(batch->num_cols == numCols, "batch does not have the exchange's columns");
  if (x->sent - x->receivedCount >= kExSlots - 1) {
    VX_THROW(VX355_EINVAL, "two exchanges are in flight: receive one before the next send");
  }
  ExSlot& slot = x->slots[x->sent % kExSlots];
  const int64_t n = batch->num_rows;
  std::vector<int32_t> all(numCols);
  for (int32_t i = 0; i < numCols; ++i) {
    all[i] = i;
    const vx355_column& col = batch->cols[i];
    VX_CHECK_ARG(col.type_kind == x->types[i], "column type differs from the exchange's");
    if (col.encoding != VX355_FLAT || col.nulls) {
      // The wire carries the rows' values only (Destination::flush serialises flat pages,
      // exec/PartitionedOutput.cpp:59-133): the shim flattens encodings; nulls are not carried yet.
      VX_THROW(VX355_EUNSUPPORTED, "exchange columns must be FLAT without nulls");
    }
  }
  DeviceBatch db;
  db.load(batch, all);   // host columns are staged; device columns aliased
  if (n > 0 && c->viaRccl()) {
    uint32_t* flag = nullptr;
    for (int32_t i = 0; i < numCols; ++i) {
      if (!isString(x->types[i])) {
        continue;
      }
      if (!flag) {
        flag = static_cast<uint32_t*>(x->parts.ensure(static_cast<size_t>(n) * 4 + 64));
        HIP_OK(hipMemsetAsync(flag, 0, 4, rt.stream));
      }
      VX_LAUNCH("k_any_long_string", k_any_long_string, streamGrid(n, 256), 256, 0,
                static_cast<const uint4*>(db.col(i).values), n, flag);
    }
    if (flag) {
      uint32_t any = 0;
      copyOut(&any, VX355_MEM_HOST, flag, 4);
      if (any) {
        VX_THROW(VX355_EUNSUPPORTED, "strings longer than 12 bytes travel as PrestoPages (vx355_presto_serialize)");
      }
    }
  }
  slot.sendCounts.assign(c->world, 0);
  slot.recvCounts.assign(c->world, 0);
  slot.grouped.resize(numCols);
  slot.received.resize(numCols);
  std::vector<const void*> in(numCols);
  std::vector<void*> grouped(numCols);
  for (int32_t i = 0; i < numCols; ++i) {
    in[i] = db.col(i).values;
  }
  if (!c->viaRccl()) {
    // one destination: nothing to hash, group or send — the rows go straight to the receive
    // buffers (one device copy: the caller's batch is only borrowed for this call)
    slot.sendCounts[0] = slot.recvCounts[0] = n;
    slot.rows = n;
    for (int32_t i = 0; i < numCols; ++i) {
      void* dst = slot.received[i].ensure(static_cast<size_t>(std::max<int64_t>(n, 1)) * x->widths[i] + 64);
      if (n > 0) {
        HIP_OK(hipMemcpyAsync(dst, in[i], static_cast<size_t>(n) * x->widths[i], hipMemcpyDeviceToDevice, rt.stream));
      }
    }
    HIP_OK(hipEventRecord(slot.done, rt.stream));
    slot.inFlight = true;
    ++x->sent;
    return VX355_OK;
  }
  for (int32_t i = 0; i < numCols; ++i) {
    grouped[i] = slot.grouped[i].ensure(static_cast<size_t>(std::max<int64_t>(n, 1)) * x->widths[i] + 64);
  }
  if (n > 0) {
    HashPartArgs a = destinationArgs(*x, db, static_cast<uint32_t>(c->world), n);
    a.out = static_cast<uint32_t*>(x->parts.ensure(static_cast<size_t>(n) * 4 + 64));
    VX_LAUNCH("k_hash_partition", k_hash_partition, streamGrid(n, 256), 256, 0, a);
    std::vector<int32_t> widths(x->widths);
    const int rc = vx355_partition_scatter(a.out, static_cast<int32_t>(n), c->world, in.data(), widths.data(), numCols,
                                           grouped.data(), slot.sendCounts.data(), VX355_MEM_DEVICE);
    if (rc != VX355_OK) {
      VX_THROW(rc, vx355_last_error());
    }
  }
  exchangeCounts(c, slot.sendCounts.data(), slot.recvCounts.data());
  slot.rows = 0;
  for (int64_t r : slot.recvCounts) {
    slot.rows += r;
  }
  std::vector<void*> received(numCols);
  for (int32_t i = 0; i < numCols; ++i) {
    received[i] = slot.received[i].ensure(static_cast<size_t>(std::max<int64_t>(slot.rows, 1)) * x->widths[i] + 64);
  }
  // The payload rides its own stream: this call returns while the slices are on the links, and
  // the caller hashes / groups the next batch or probes the previous one meanwhile.
  HIP_OK(hipEventRecord(x->staged, rt.stream));
  HIP_OK(hipStreamWaitEvent(x->payload, x->staged, 0));
  std::vector<const void*> sendPtrs(grouped.begin(), grouped.end());
  postColumns(c, x->payload, sendPtrs.data(), x->widths.data(), numCols, slot.sendCounts.data(),
              slot.recvCounts.data(), received.data());
  HIP_OK(hipEventRecord(slot.done, x->payload));
  slot.inFlight = true;
  ++x->sent;
  VX_API_END
}

int vx355_exchange_receive(vx355_exchange* x, vx355_column* cols_out, int64_t* rows_out) {
  VX_API_BEGIN_CTX(VX_CTX_OF(x))
  VX_CHECK_ARG(x && cols_out && rows_out, "NULL argument");
  VX_CHECK_ARG(x->receivedCount < x->sent, "nothing was sent");
  ExSlot& slot = x->slots[x->receivedCount % kExSlots];
  HIP_OK(hipEventSynchronize(slot.done));
  slot.inFlight = false;
  ++x->receivedCount;
  for (size_t i = 0; i < x->types.size(); ++i) {
    vx355_column col{};
    col.type_kind = x->types[i];
    col.encoding = VX355_FLAT;
    col.values = slot.received[i].ptr();
    col.mem = VX355_MEM_DEVICE;
    cols_out[i] = col;
  }
  *rows_out = slot.rows;
  VX_API_END
}

// The destination rank of every row of 'batch' as this edge computes it for 'num_destinations'
// ranks (0 = the communicator's size): VectorHasher::hash of the key columns +
// HashPartitionFunction::partition, the fused kernel vx355_exchange_send runs. For inspection and
// for parity tests of the N > 1 grouping on a box with one GPU.
int vx355_exchange_destinations(vx355_exchange* x, const vx355_batch* batch, int32_t num_destinations,
                                uint32_t* out, int32_t out_mem) {
  VX_API_BEGIN_CTX(VX_CTX_OF(x))
  VX_CHECK_ARG(x && batch && out, "NULL argument");
  VX_CHECK_ARG(num_destinations >= 0 && num_destinations <= 64, "0..64 destinations");
  const int64_t n = batch->num_rows;
  if (n == 0) {
    return VX355_OK;
  }
  DeviceBatch db;
  db.load(batch, x->keyCols);
  const uint32_t world = num_destinations ? static_cast<uint32_t>(num_destinations) : static_cast<uint32_t>(x->comm->world);
  HashPartArgs a = destinationArgs(*x, db, world, n);
  DevBuf tmp;
  a.out = out_mem == VX355_MEM_HOST ? static_cast<uint32_t*>(tmp.ensure(static_cast<size_t>(n) * 4 + 64)) : out;
  VX_LAUNCH("k_hash_partition", k_hash_partition, streamGrid(n, 256), 256, 0, a);
  if (out_mem == VX355_MEM_HOST) {
    copyOut(out, VX355_MEM_HOST, a.out, static_cast<size_t>(n) * 4);
  }
  VX_API_END
}

void* vx355_exchange_stream(vx355_exchange* x) { return x ? static_cast<void*>(x->ctx->stream) : nullptr; }

void vx355_exchange_destroy(vx355_exchange* x) {
  if (!x) {
    return;
  }
  Runtime* ctx = x->ctx;
  try {
    vx::ContextScope scope(ctx);
    if (x->payload) {
      (void)hipStreamSynchronize(x->payload);
      (void)hipStreamDestroy(x->payload);
    }
    if (x->staged) {
      (void)hipEventDestroy(x->staged);
    }
    for (auto& s : x->slots) {
      if (s.done) {
        (void)hipEventDestroy(s.done);
      }
    }
    delete x;
  } catch (...) {
  }
  Runtime::destroyContext(ctx);
}

}  // extern "C"
