// Second provider of the exchange's transport table (transport.h): ranks that share ONE GPU.
// RCCL refuses two ranks per device, so on a 1-GPU box the library's own exchange - the
// counts-first protocol between different ranks, grouped send / recv with several peers, the
// three-slot receive pipeline, the chunk-count agreement of vx355_join_repartition, the
// partial -> final merge - could only ever run with world = 1. Here the bytes of a message travel
//   sender HBM -> (hipMemcpy) -> a ring of pieces in a POSIX shared-memory segment -> (hipMemcpy) -> receiver HBM
// with RCCL's semantics: messages between two ranks match in posting order, a group's sends and
// receives progress together (so "everybody sends first" cannot deadlock), zero device-side
// assumptions. Speed is not the point (a few GB/s): the protocol code above the table is.
// One channel per ordered pair of ranks: kRing pieces of kPieceBytes.
#include "transport.h"

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace vx {
namespace shmx {
namespace {

constexpr uint32_t kMagic = 0x76783335;  // "vx35"
constexpr int kMaxWorld = 8;
constexpr size_t kPieceBytes = 1 << 20;
constexpr int kRing = 4;
constexpr int kErrInternal = 3;   // ncclInternalError
constexpr int kErrInvalid = 4;    // ncclInvalidArgument

struct Header {
  std::atomic<uint32_t> magic;
  uint32_t world;
  std::atomic<uint32_t> joined;
  std::atomic<uint32_t> left;
  std::atomic<uint32_t> aborted;
  uint32_t pad[11];
};

struct Channel {
  std::atomic<uint64_t> written;   // pieces the source has published
  std::atomic<uint64_t> consumed;  // pieces the destination has taken
  uint64_t pieceBytes[kRing];
  uint64_t pad[2];
};

struct Comm {
  int world = 0, rank = 0, device = 0;
  std::string name;
  size_t bytes = 0;
  char* base = nullptr;
  Header* header() const { return reinterpret_cast<Header*>(base); }
  Channel* channel(int src, int dst) const {
    return reinterpret_cast<Channel*>(base + sizeof(Header)) + (src * world + dst);
  }
  char* data(int src, int dst, uint64_t piece) const {
    char* first = base + sizeof(Header) + sizeof(Channel) * world * world;
    return first + (static_cast<size_t>(src * world + dst) * kRing + (piece % kRing)) * kPieceBytes;
  }
  static size_t segmentBytes(int world) {
    return sizeof(Header) + sizeof(Channel) * world * world + static_cast<size_t>(world) * world * kRing * kPieceBytes;
  }
};

struct Op {
  bool send;
  char* ptr;
  size_t bytes, done;
  int peer;
  Comm* comm;
  hipStream_t stream;
};

thread_local int tGroupDepth = 0;
thread_local std::vector<Op> tOps;
thread_local std::string tError;

double seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

double timeoutSeconds() {
  if (const char* e = std::getenv("VX355_SHM_TIMEOUT")) {
    return std::atof(e);
  }
  return 120.0;
}

size_t dtypeBytes(int dtype) {
  switch (dtype) {
    case kNcclInt8:
    case kNcclUint8:
      return 1;
    case 2:  // int32
    case 3:  // uint32
    case 7:  // float
      return 4;
    case kNcclInt64:
    case 5:  // uint64
    case 8:  // double
      return 8;
    case 6:  // half
    case 9:  // bfloat16
      return 2;
    default:
      return 0;
  }
}

// Runs the queued operations of the calling thread to completion.
int progressAll() {
  std::vector<Op> ops;
  ops.swap(tOps);
  if (ops.empty()) {
    return kNcclSuccess;
  }
  // what the streams have queued in front of the sends must have produced the bytes
  std::vector<hipStream_t> synced;
  for (const Op& op : ops) {
    bool seen = false;
    for (hipStream_t s : synced) {
      seen = seen || s == op.stream;
    }
    if (!seen) {
      if (hipStreamSynchronize(op.stream) != hipSuccess) {
        tError = "hipStreamSynchronize failed";
        return kErrInternal;
      }
      synced.push_back(op.stream);
    }
  }
  // a rank's messages to itself: matched in posting order, one device copy each
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& s = ops[i];
    if (!s.send || s.peer != s.comm->rank || s.done == s.bytes) {
      continue;
    }
    for (size_t j = 0; j < ops.size(); ++j) {
      Op& r = ops[j];
      if (r.send || r.comm != s.comm || r.peer != r.comm->rank || r.done == r.bytes) {
        continue;
      }
      if (r.bytes != s.bytes) {
        tError = "self send / recv sizes differ";
        return kErrInvalid;
      }
      if (hipMemcpy(r.ptr, s.ptr, s.bytes, hipMemcpyDeviceToDevice) != hipSuccess) {
        tError = "hipMemcpy (self) failed";
        return kErrInternal;
      }
      r.done = r.bytes;
      s.done = s.bytes;
      break;
    }
    if (s.done != s.bytes) {
      tError = "send to self without a matching recv in the group";
      return kErrInvalid;
    }
  }
  // every error path below marks the segment aborted: peers then fail at once instead of spinning
  // until VX355_SHM_TIMEOUT, and nobody reads a ring that holds half a message
  auto abortAll = [&]() {
    for (const Op& op : ops) {
      op.comm->header()->aborted.store(1);
    }
  };
  const double deadline = seconds() + timeoutSeconds();
  for (;;) {
    bool all = true, moved = false;
    // per channel only the FIRST unfinished operation may progress: messages match in posting order. A
    // channel belongs to a communicator: operations of different communicators in one group do not wait
    // for each other.
    std::vector<std::pair<const Comm*, int>> sendBusy, recvBusy;
    for (Op& op : ops) {
      if (op.done == op.bytes) {
        continue;
      }
      all = false;
      Comm* c = op.comm;
      auto& busy = op.send ? sendBusy : recvBusy;
      const std::pair<const Comm*, int> channelOf{c, op.peer};
      if (std::find(busy.begin(), busy.end(), channelOf) != busy.end()) {
        continue;
      }
      busy.push_back(channelOf);
      if (op.send) {
        Channel* ch = c->channel(c->rank, op.peer);
        const uint64_t w = ch->written.load(std::memory_order_relaxed);
        if (w - ch->consumed.load(std::memory_order_acquire) >= kRing) {
          continue;
        }
        const size_t n = std::min(kPieceBytes, op.bytes - op.done);
        if (hipMemcpy(c->data(c->rank, op.peer, w), op.ptr + op.done, n, hipMemcpyDeviceToHost) != hipSuccess) {
          tError = "hipMemcpy (device to shared memory) failed";
          abortAll();
          return kErrInternal;
        }
        ch->pieceBytes[w % kRing] = n;
        ch->written.store(w + 1, std::memory_order_release);
        op.done += n;
        moved = true;
      } else {
        Channel* ch = c->channel(op.peer, c->rank);
        const uint64_t r = ch->consumed.load(std::memory_order_relaxed);
        if (ch->written.load(std::memory_order_acquire) <= r) {
          continue;
        }
        const size_t n = ch->pieceBytes[r % kRing];
        if (n != std::min(kPieceBytes, op.bytes - op.done)) {
          tError = "message size mismatch between sender and receiver";
          c->header()->aborted.store(1);
          return kErrInvalid;
        }
        if (hipMemcpy(op.ptr + op.done, c->data(op.peer, c->rank, r), n, hipMemcpyHostToDevice) != hipSuccess) {
          tError = "hipMemcpy (shared memory to device) failed";
          abortAll();
          return kErrInternal;
        }
        ch->consumed.store(r + 1, std::memory_order_release);
        op.done += n;
        moved = true;
      }
    }
    if (all) {
      return kNcclSuccess;
    }
    if (!moved) {
      for (const Op& op : ops) {
        if (op.comm->header()->aborted.load()) {
          tError = "a peer aborted the exchange";
          return kErrInternal;
        }
      }
      if (seconds() > deadline) {
        tError = "timed out waiting for a peer (VX355_SHM_TIMEOUT seconds)";
        for (const Op& op : ops) {
          op.comm->header()->aborted.store(1);
        }
        return kErrInternal;
      }
      sched_yield();
    }
  }
}

int post(bool send, const void* buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t width = dtypeBytes(dtype);
  if (!c || width == 0 || peer < 0 || peer >= c->world) {
    tError = "bad send / recv argument";
    return kErrInvalid;
  }
  if (count == 0) {
    return kNcclSuccess;
  }
  tOps.push_back(Op{send, static_cast<char*>(const_cast<void*>(buf)), count * width, 0, peer, c, stream});
  return tGroupDepth > 0 ? kNcclSuccess : progressAll();
}

}  // namespace

int GetUniqueId(ncclUniqueId* id) {
  std::memset(id->internal, 0, sizeof(id->internal));
  std::random_device rd;
  std::snprintf(id->internal, sizeof(id->internal), "/vx355_shm_%d_%08x%08x", static_cast<int>(getpid()), rd(), rd());
  return kNcclSuccess;
}

int CommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > kMaxWorld || rank < 0 || rank >= world || id.internal[0] != '/') {
    tError = "shm transport: bad communicator arguments (world <= 8, id from vx355_comm_get_unique_id)";
    return kErrInvalid;
  }
  auto c = new Comm();
  c->world = world;
  c->rank = rank;
  c->name.assign(id.internal, strnlen(id.internal, sizeof(id.internal)));
  c->bytes = Comm::segmentBytes(world);
  (void)hipGetDevice(&c->device);
  const double deadline = seconds() + timeoutSeconds();
  int fd = shm_open(c->name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
  const bool creator = fd >= 0;
  if (creator) {
    if (ftruncate(fd, static_cast<off_t>(c->bytes)) != 0) {
      tError = "shm transport: cannot size the segment (" + std::to_string(c->bytes >> 20) + " MiB in /dev/shm)";
      close(fd);
      shm_unlink(c->name.c_str());
      delete c;
      return kErrInternal;
    }
  } else {
    for (;;) {
      fd = shm_open(c->name.c_str(), O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && static_cast<size_t>(st.st_size) >= c->bytes) {
        break;
      }
      if (fd >= 0) {
        close(fd);
      }
      if (seconds() > deadline) {
        tError = "shm transport: the segment of rank 0 never appeared";
        delete c;
        return kErrInternal;
      }
      usleep(1000);
    }
  }
  void* p = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    tError = "shm transport: mmap failed";
    if (creator) {
      shm_unlink(c->name.c_str());
    }
    delete c;
    return kErrInternal;
  }
  c->base = static_cast<char*>(p);
  Header* h = c->header();
  if (creator) {
    // (a fresh segment is zero filled: counters start at 0)
    h->world = static_cast<uint32_t>(world);
    h->magic.store(kMagic, std::memory_order_release);
  }
  while (h->magic.load(std::memory_order_acquire) != kMagic) {
    if (seconds() > deadline) {
      tError = "shm transport: the segment was never initialised";
      munmap(c->base, c->bytes);
      delete c;
      return kErrInternal;
    }
    usleep(1000);
  }
  if (h->world != static_cast<uint32_t>(world)) {
    tError = "shm transport: ranks disagree about the world size";
    munmap(c->base, c->bytes);
    delete c;
    return kErrInvalid;
  }
  h->joined.fetch_add(1);
  while (h->joined.load() < static_cast<uint32_t>(world)) {
    if (seconds() > deadline) {
      tError = "shm transport: not every rank joined";
      h->aborted.store(1);
      munmap(c->base, c->bytes);
      delete c;
      return kErrInternal;
    }
    usleep(1000);
  }
  if (creator) {
    shm_unlink(c->name.c_str());  // everybody has it mapped: the name can go
  }
  *out = c;
  return kNcclSuccess;
}

// ncclCommInitAll: ONE process creates all 'world' communicators (the shape SURVEY.md 8(e) names: one
// process, one Driver thread per rank). Every rank's CommInitRank waits for the others to join, so
// they run on helper threads side by side; rank i's communicator reports devices[i].
int CommInitAll(ncclComm_t* comms, int world, const int* devices) {
  if (!comms || world < 1 || world > kMaxWorld) {
    tError = "shm transport: bad communicator arguments (world <= 8)";
    return kErrInvalid;
  }
  ncclUniqueId id;
  const int rc = GetUniqueId(&id);
  if (rc != kNcclSuccess) {
    return rc;
  }
  std::vector<int> status(static_cast<size_t>(world), kNcclSuccess);
  std::vector<std::string> errors(static_cast<size_t>(world));
  std::vector<std::thread> helpers;
  for (int r = 0; r < world; ++r) {
    helpers.emplace_back([&, r] {
      if (devices) {
        (void)hipSetDevice(devices[r]);
      }
      status[r] = CommInitRank(&comms[r], world, id, r);
      errors[r] = tError;
    });
  }
  for (auto& t : helpers) {
    t.join();
  }
  for (int r = 0; r < world; ++r) {
    if (status[r] != kNcclSuccess) {
      tError = errors[r];
      for (int q = 0; q < world; ++q) {
        if (status[q] == kNcclSuccess) {
          CommDestroy(comms[q]);
          comms[q] = nullptr;
        }
      }
      return status[r];
    }
  }
  return kNcclSuccess;
}

int CommDestroy(ncclComm_t comm) {
  Comm* c = static_cast<Comm*>(comm);
  if (c) {
    c->header()->left.fetch_add(1);
    munmap(c->base, c->bytes);
    delete c;
  }
  return kNcclSuccess;
}

int Send(const void* buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(true, buf, count, dtype, peer, comm, stream);
}

int Recv(void* buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
  return post(false, buf, count, dtype, peer, comm, stream);
}

int AllGather(const void* send, void* recv, size_t count, int dtype, ncclComm_t comm, hipStream_t stream) {
  Comm* c = static_cast<Comm*>(comm);
  const size_t bytes = count * dtypeBytes(dtype);
  if (!c || dtypeBytes(dtype) == 0) {
    tError = "bad all-gather argument";
    return kErrInvalid;
  }
  ++tGroupDepth;
  int rc = kNcclSuccess;
  for (int peer = 0; peer < c->world && rc == kNcclSuccess; ++peer) {
    rc = post(true, send, bytes, kNcclUint8, peer, comm, stream);
    if (rc == kNcclSuccess) {
      rc = post(false, static_cast<char*>(recv) + static_cast<size_t>(peer) * bytes, bytes, kNcclUint8, peer, comm, stream);
    }
  }
  --tGroupDepth;
  if (rc != kNcclSuccess) {
    tOps.clear();
    return rc;
  }
  return tGroupDepth > 0 ? kNcclSuccess : progressAll();
}

int GroupStart() {
  ++tGroupDepth;
  return kNcclSuccess;
}

int GroupEnd() {
  if (tGroupDepth <= 0) {
    tError = "ncclGroupEnd without ncclGroupStart";
    return kErrInvalid;
  }
  if (--tGroupDepth > 0) {
    return kNcclSuccess;
  }
  return progressAll();
}

int CommCount(ncclComm_t comm, int* count) {
  *count = static_cast<Comm*>(comm)->world;
  return kNcclSuccess;
}
int CommUserRank(ncclComm_t comm, int* rank) {
  *rank = static_cast<Comm*>(comm)->rank;
  return kNcclSuccess;
}
int CommCuDevice(ncclComm_t comm, int* device) {
  *device = static_cast<Comm*>(comm)->device;
  return kNcclSuccess;
}
const char* GetErrorString(int rc) {
  (void)rc;
  return tError.empty() ? "shm transport error" : tError.c_str();
}

}  // namespace shmx
}  // namespace vx
