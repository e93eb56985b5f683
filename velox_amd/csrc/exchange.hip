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

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>

namespace vx {
namespace {

// The part of rccl.h the exchange uses (RCCL keeps NCCL's ABI).
using ncclComm_t = void*;
constexpr int kUniqueIdBytes = 128;
struct ncclUniqueId {
  char internal[kUniqueIdBytes];
};
constexpr int kNcclSuccess = 0;
constexpr int kNcclInt8 = 0;    // ncclInt8 / ncclChar
constexpr int kNcclUint8 = 1;
constexpr int kNcclInt64 = 4;

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

}  // namespace
}  // namespace vx

using namespace vx;

struct vx355_comm {
  vx::Runtime* ctx = nullptr;  // execution context on the communicator's GPU
  void* comm = nullptr;        // ncclComm_t
  int32_t world = 1;
  int32_t rank = 0;
  DevBuf countsDev;            // all-gather of the slice sizes
};

extern "C" {

int vx355_comm_get_unique_id(void* id_out) {
  try {
    VX_CHECK_ARG(id_out, "NULL argument");
    ncclUniqueId id;
    ncclOk(rccl().GetUniqueId(&id), "ncclGetUniqueId");
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
  ncclOk(rccl().CommInitRank(&c->comm, world, uid, rank), "ncclCommInitRank");
  c->ctx = Runtime::createContext();
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
    ncclOk(rccl().CommInitAll(comms.data(), num_devices, devs.data()), "ncclCommInitAll");
    for (int32_t i = 0; i < num_devices; ++i) {
      vx::ContextScope scope(Runtime::defaultContext(devs[i]));
      auto c = std::make_unique<vx355_comm>();
      c->world = num_devices;
      c->rank = i;
      c->comm = comms[i];
      c->ctx = Runtime::createContext();
      out[i] = c.release();
    }
  VX_API_CATCH
}

int vx355_comm_info(const vx355_comm* c, int32_t* world, int32_t* rank, int32_t* device) {
  try {
    VX_CHECK_ARG(c, "NULL argument");
    if (world) {
      *world = c->world;
    }
    if (rank) {
      *rank = c->rank;
    }
    if (device) {
      *device = c->ctx->device;
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

// recv_counts[s] = rows rank s holds for this rank. One all-gather of 'world' int64 per rank.
int vx355_exchange_counts(vx355_comm* c, const int64_t* send_counts, int64_t* recv_counts) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send_counts && recv_counts, "NULL argument");
  auto& rt = Runtime::get();
  const size_t w = static_cast<size_t>(c->world);
  int64_t* dev = static_cast<int64_t*>(c->countsDev.ensure((w + w * w) * 8 + 64));
  copyIn(dev, send_counts, VX355_MEM_HOST, w * 8);
  ncclOk(rccl().AllGather(dev, dev + w, w, kNcclInt64, c->comm, rt.stream), "ncclAllGather(counts)");
  std::vector<int64_t> all(w * w);
  copyOut(all.data(), VX355_MEM_HOST, dev + w, w * w * 8);
  for (size_t s = 0; s < w; ++s) {
    recv_counts[s] = all[s * w + static_cast<size_t>(c->rank)];
  }
  VX_API_END
}

int vx355_exchange_columns(vx355_comm* c, const void* const* send_cols, const int32_t* widths, int32_t num_cols,
                           const int64_t* send_counts, const int64_t* recv_counts, void* const* recv_cols) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send_counts && recv_counts && num_cols >= 0, "bad argument");
  VX_CHECK_ARG(num_cols == 0 || (send_cols && widths && recv_cols), "NULL column arrays");
  auto& rt = Runtime::get();
  Rccl& r = rccl();
  // Grouped point-to-point: all slices of all columns are posted together so that the
  // seven outgoing links of the GPU work concurrently.
  ncclOk(r.GroupStart(), "ncclGroupStart");
  for (int32_t col = 0; col < num_cols; ++col) {
    const int64_t w = widths[col];
    VX_CHECK_ARG(w >= 1, "column width");
    int64_t sendAt = 0, recvAt = 0;
    for (int32_t peer = 0; peer < c->world; ++peer) {
      const int64_t ns = send_counts[peer], nr = recv_counts[peer];
      VX_CHECK_ARG(ns >= 0 && nr >= 0, "negative slice size");
      if (ns > 0) {
        ncclOk(r.Send(static_cast<const char*>(send_cols[col]) + sendAt * w, static_cast<size_t>(ns * w), kNcclUint8,
                      peer, c->comm, rt.stream),
               "ncclSend");
      }
      if (nr > 0) {
        ncclOk(r.Recv(static_cast<char*>(recv_cols[col]) + recvAt * w, static_cast<size_t>(nr * w), kNcclUint8, peer,
                      c->comm, rt.stream),
               "ncclRecv");
      }
      sendAt += ns;
      recvAt += nr;
    }
  }
  ncclOk(r.GroupEnd(), "ncclGroupEnd");
  rt.sync();
  VX_API_END
}

int vx355_all_gather(vx355_comm* c, const void* send, void* recv, size_t bytes_per_rank) {
  VX_API_BEGIN_CTX(VX_CTX_OF(c))
  VX_CHECK_ARG(c && send && recv, "NULL argument");
  auto& rt = Runtime::get();
  if (bytes_per_rank) {
    ncclOk(rccl().AllGather(send, recv, bytes_per_rank, kNcclInt8, c->comm, rt.stream), "ncclAllGather");
  }
  rt.sync();
  VX_API_END
}

}  // extern "C"
