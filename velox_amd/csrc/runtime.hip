// libvx355 runtime: device binding, library stream, pinned mailbox, device
// memory helpers, batch staging and the HIP-event profiler.
#include "common.h"

#include <chrono>
#include <cstdlib>

#include <algorithm>
#include <cstdio>

namespace vx {

// (internal linkage is not needed: these names live in namespace vx and are used by the
// extern "C" entry points at the end of this file)
thread_local std::string tlsLastError;
thread_local Runtime* tlsCurrent = nullptr;  // context of the entry point running on this thread
thread_local int tlsDevice = -1;             // vx355_set_device

constexpr int kMaxDevices = 64;
std::mutex gInitMutex;
DeviceState* gDevices[kMaxDevices] = {};     // never deleted: buffers may outlive vx355_shutdown
std::atomic<int> gDefaultDevice{-1};
std::atomic<uint64_t> gNextContextId{1};

// The thread's HIP device is never cached: the embedding host and librccl
// (ncclCommInitAll / ncclCommInitRank) change it behind the library's back, and a
// stale binding would make hipMalloc / hipStreamCreate / launches land on another GPU.
// hipSetDevice costs ~50 ns when the device is already current.
void bindHipDevice(int device) { HIP_OK(hipSetDevice(device)); }

DeviceState* deviceState(int device) {
  if (device < 0) {
    device = tlsDevice >= 0 ? tlsDevice : gDefaultDevice.load();
  }
  if (device < 0 || device >= kMaxDevices || !gDevices[device] || !gDevices[device]->alive) {
    VX_THROW(VX355_EINVAL, "vx355_init has not been called");
  }
  return gDevices[device];
}

Runtime* newContext(DeviceState* ds, bool isDefault) {
  bindHipDevice(ds->device);
  auto ctx = std::make_unique<Runtime>(ds);
  ctx->device = ds->device;
  ctx->numCUs = ds->numCUs;
  ctx->ldsPerBlock = ds->ldsPerBlock;
  ctx->isDefault = isDefault;
  ctx->id = gNextContextId.fetch_add(1);
  HIP_OK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&ctx->mail.host), Mailbox::kWords * 8, hipHostMallocMapped));
  HIP_OK(hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->mail.dev), ctx->mail.host, 0));
  std::memset(ctx->mail.host, 0, Mailbox::kWords * 8);
  ctx->initialized = true;
  {
    std::lock_guard<std::mutex> lock(ds->memMutex);
    ds->contexts[ctx->id] = ctx.get();
  }
  return ctx.release();
}

void setLastError(const std::string& m) { tlsLastError = m; }

void hipFail(hipError_t e, const char* what, const char* file, int line) {
  char buf[512];
  snprintf(buf, sizeof(buf), "HIP error %d (%s) at %s:%d: %s", static_cast<int>(e),
           hipGetErrorString(e), file, line, what);
  (void)hipGetLastError();  // clear sticky non-fatal errors
  VX_THROW(e == hipErrorOutOfMemory ? VX355_ENOMEM : VX355_EINTERNAL, buf);
}

Runtime* Runtime::tryGet() {
  if (tlsCurrent) {
    return tlsCurrent;
  }
  const int device = tlsDevice >= 0 ? tlsDevice : gDefaultDevice.load();
  if (device < 0 || !gDevices[device] || !gDevices[device]->alive) {
    return nullptr;
  }
  return gDevices[device]->defaultCtx;
}

Runtime& Runtime::get() {
  Runtime* rt = tryGet();
  if (!rt) {
    VX_THROW(VX355_EINVAL, "vx355_init has not been called");
  }
  return *rt;
}

Runtime* Runtime::defaultContext(int device) { return deviceState(device)->defaultCtx; }

Runtime* Runtime::createContext() {
  // A handle created inside another handle's entry point lives on that handle's device.
  DeviceState* ds = tlsCurrent ? tlsCurrent->ds : deviceState(-1);
  {
    std::lock_guard<std::mutex> lock(ds->memMutex);
    if (!ds->idleContexts.empty()) {
      Runtime* ctx = ds->idleContexts.back();
      ds->idleContexts.pop_back();
      ctx->resetGpuStats();
      return ctx;  // idle: its last entry point drained the stream
    }
  }
  return newContext(ds, false);
}

void freeContext(Runtime* ctx) {
  (void)hipStreamSynchronize(ctx->stream);
  (void)hipHostFree(ctx->mail.host);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

void Runtime::destroyContext(Runtime* ctx) {
  if (!ctx) {
    return;
  }
  (void)hipStreamSynchronize(ctx->stream);
  DeviceState* ds = ctx->ds;
  {
    std::lock_guard<std::mutex> lock(ds->memMutex);
    if (ds->alive && !ctx->isDefault && ds->idleContexts.size() < 64) {
      ds->idleContexts.push_back(ctx);  // stays registered in ds->contexts
      return;
    }
    ds->contexts.erase(ctx->id);
  }
  freeContext(ctx);
}

ContextScope::ContextScope(Runtime* ctx) : ctx_(ctx), prev_(tlsCurrent) {
  if (!ctx_) {
    // Handle-less entry point. Nested inside another entry point (the Python
    // harness never does that, the library itself may): stay in its context.
    ctx_ = tlsCurrent ? tlsCurrent : deviceState(-1)->defaultCtx;
  }
  outer_ = ctx_ != prev_;
  if (!outer_) {
    return;
  }
  if (ctx_->isDefault) {
    ctx_->callMutex.lock();
    locked_ = true;
  }
  bindHipDevice(ctx_->device);
  tlsCurrent = ctx_;
  enteredNanos_ = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

ContextScope::~ContextScope() {
  if (!outer_) {
    return;
  }
  // Everything the entry point queued is complete when it returns.
  (void)hipStreamSynchronize(ctx_->stream);
  ctx_->busyNanos += static_cast<uint64_t>(
      std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() -
      enteredNanos_);
  ctx_->doneCalls.store(ctx_->currentCall);
  ++ctx_->currentCall;
  tlsCurrent = prev_;
  if (prev_) {
    (void)hipSetDevice(prev_->device);
  }
  if (locked_) {
    ctx_->callMutex.unlock();
  }
}

hipEvent_t Runtime::newEvent() {
  {
    std::lock_guard<std::mutex> lock(ds->profMutex);
    if (!ds->freeEvents.empty()) {
      hipEvent_t e = ds->freeEvents.back();
      ds->freeEvents.pop_back();
      return e;
    }
  }
  hipEvent_t e;
  HIP_OK(hipEventCreate(&e));
  return e;
}

void Runtime::profBegin(const char* name) {
  hipEvent_t a = newEvent();
  HIP_OK(hipEventRecord(a, stream));
  std::lock_guard<std::mutex> lock(ds->profMutex);
  auto& entry = ds->prof[name];
  entry.open[id] = a;
}

void Runtime::profEnd(const char* name) {
  hipEvent_t b = newEvent();
  HIP_OK(hipEventRecord(b, stream));
  std::lock_guard<std::mutex> lock(ds->profMutex);
  auto& entry = ds->prof[name];
  auto it = entry.open.find(id);
  if (it != entry.open.end()) {
    entry.events.emplace_back(it->second, b);
    entry.open.erase(it);
    ++entry.launches;
  } else {
    ds->freeEvents.push_back(b);
  }
}

static bool logAlloc() {
  static const bool on = std::getenv("VX355_LOG_ALLOC") != nullptr;
  return on;
}

void* DeviceState::allocBlock(size_t bytes, size_t* actual) {
  // Size classes: powers of two up to 1 MiB, then multiples of 1 MiB.
  size_t want = bytes <= (1u << 20) ? static_cast<size_t>(nextPow2(std::max<size_t>(bytes, 256)))
                                    : ((bytes + (1u << 20) - 1) >> 20) << 20;
  Runtime* cur = Runtime::tryGet();
  {
    std::unique_lock<std::mutex> lock(memMutex);
    if (memoryLimit != 0 && liveBytes + want > memoryLimit) {
      // MemoryPool::allocate past its capacity (the reference then reclaims / spills, or fails the
      // query with "Exceeded memory pool capacity"): here the operator's entry point fails with
      // VX355_ENOMEM, its handle stays destroyable and everything it held goes back on destroy.
      VX_THROW(VX355_ENOMEM, "memory limit of " + std::to_string(memoryLimit) + " bytes exceeded: " + std::to_string(want) +
                                 " bytes asked with " + std::to_string(liveBytes) + " in use (vx355_set_memory_limit)");
    }
    auto it = freeBlocks.lower_bound(want);
    if (it != freeBlocks.end() && it->first <= want + want / 4 &&
        (memoryLimit == 0 || liveBytes + it->first <= memoryLimit)) {
      const CachedBlock b = it->second;
      *actual = it->first;
      cachedBytes -= it->first;
      liveBytes += it->first;
      peakLiveBytes = std::max(peakLiveBytes, liveBytes);
      freeBlocks.erase(it);
      // Released by another context whose call has not returned yet: its stream
      // may still touch the block. Same context: ordered on the same stream.
      if (b.ownerCtx != 0 && (!cur || b.ownerCtx != cur->id)) {
        auto owner = contexts.find(b.ownerCtx);
        if (owner != contexts.end() && owner->second->doneCalls.load() < b.ownerCall) {
          hipStream_t s = owner->second->stream;
          lock.unlock();
          (void)hipStreamSynchronize(s);
        }
      }
      if (logAlloc()) {
        fprintf(stderr, "vx355 alloc: cached %p %zu (asked %zu)\n", b.p, *actual, want);
      }
      return b.p;
    }
  }
  void* p = nullptr;
  bindHipDevice(device);
  hipError_t e = hipMalloc(&p, want);
  if (logAlloc()) {
    fprintf(stderr, "vx355 alloc: fresh %p %zu\n", p, want);
  }
  if (e == hipErrorOutOfMemory) {
    (void)hipGetLastError();
    trimCache();
    e = hipMalloc(&p, want);
  }
  if (e != hipSuccess) {
    hipFail(e, "hipMalloc", __FILE__, __LINE__);
  }
  *actual = want;
  {
    std::lock_guard<std::mutex> lock(memMutex);
    liveBytes += want;
    peakLiveBytes = std::max(peakLiveBytes, liveBytes);
  }
  return p;
}

void DeviceState::freeBlock(void* p, size_t bytes) {
  if (logAlloc() && p) {
    fprintf(stderr, "vx355 alloc: release %p %zu\n", p, bytes);
  }
  if (!p) {
    return;
  }
  Runtime* cur = Runtime::tryGet();
  if (cur && cur->ds != this) {
    cur = nullptr;  // freed from a context on another GPU: nothing of ours is in flight on it
  }
  // The block just released is the one most likely to be asked for again (the next operator of the
  // same plan): when the cache is full, the blocks that have sat in it the longest make room. (A
  // process that runs one workload after another - bench.py's secondary blocks - otherwise fills
  // the cache with the first workloads' sizes and pays hipMalloc + hipFree of tens of GB per step
  // for the later ones: config 4 with sparse keys took 432 ms instead of 62.)
  std::vector<void*> evicted;
  bool keep = false;
  {
    std::lock_guard<std::mutex> lock(memMutex);
    liveBytes -= std::min(liveBytes, bytes);
    if (alive && bytes <= cacheLimit) {
      while (cachedBytes + bytes > cacheLimit && !freeBlocks.empty()) {
        auto oldest = freeBlocks.begin();
        for (auto it = freeBlocks.begin(); it != freeBlocks.end(); ++it) {
          if (it->second.seq < oldest->second.seq) {
            oldest = it;
          }
        }
        evicted.push_back(oldest->second.p);
        cachedBytes -= oldest->first;
        freeBlocks.erase(oldest);
      }
      // Work already queued on the releasing context's stream may still touch the
      // block: a later user on the same stream is ordered behind it, any other
      // context waits for that call to finish (allocBlock).
      freeBlocks.emplace(bytes, CachedBlock{p, cur ? cur->id : 0, cur ? cur->currentCall : 0, ++cacheSeq});
      cachedBytes += bytes;
      keep = true;
    }
  }
  for (void* old : evicted) {
    (void)hipFree(old);  // (hipFree waits for the device: nothing can still be using the block)
  }
  if (keep) {
    return;
  }
  if (cur) {
    (void)hipStreamSynchronize(cur->stream);
  }
  (void)hipFree(p);
}

void DeviceState::trimCache() {
  std::multimap<size_t, CachedBlock> blocks;
  std::multimap<size_t, void*> pinned;
  std::vector<hipStream_t> streams;
  {
    std::lock_guard<std::mutex> lock(memMutex);
    blocks.swap(freeBlocks);
    pinned.swap(freePinned);
    cachedBytes = 0;
    cachedPinned = 0;
    for (auto& kv : contexts) {
      streams.push_back(kv.second->stream);
    }
  }
  for (hipStream_t s : streams) {
    (void)hipStreamSynchronize(s);
  }
  for (auto& kv : blocks) {
    (void)hipFree(kv.second.p);
  }
  for (auto& kv : pinned) {
    (void)hipHostFree(kv.second);
  }
}

void* DevBuf::ensure(size_t bytes, bool preserve, size_t preserveBytes) {
  if (bytes <= cap_) {
    return p_;
  }
  auto& rt = Runtime::get();
  if (p_ && ds_ != rt.ds) {
    VX_THROW(VX355_EINTERNAL, "device buffer used from a context on another GPU");
  }
  size_t newCap = std::max<size_t>(bytes, cap_ + cap_ / 2);
  void* np = rt.allocBlock(newCap, &newCap);
  if (preserve && p_ && preserveBytes) {
    HIP_OK(hipMemcpyAsync(np, p_, std::min(preserveBytes, cap_), hipMemcpyDeviceToDevice,
                          rt.stream));
  }
  if (p_) {
    ds_->freeBlock(p_, cap_);
  }
  ds_ = rt.ds;
  p_ = np;
  cap_ = newCap;
  return p_;
}

void DevBuf::release() {
  if (p_) {
    ds_->freeBlock(p_, cap_);
    p_ = nullptr;
    cap_ = 0;
  }
}

int kindWidth(int32_t kind) {
  switch (kind) {
    case VX355_BOOLEAN:
      return 0;
    case VX355_TINYINT:
      return 1;
    case VX355_SMALLINT:
      return 2;
    case VX355_INTEGER:
    case VX355_REAL:
      return 4;
    case VX355_BIGINT:
    case VX355_DOUBLE:
      return 8;
    case VX355_VARCHAR:
    case VX355_VARBINARY:
    case VX355_TIMESTAMP:
      return 16;
    default:
      return -1;
  }
}

int streamGrid(int64_t items, int block, int perThread) {
  int64_t blocks = ceilDiv(items, static_cast<int64_t>(block) * perThread);
  int64_t cap = static_cast<int64_t>(Runtime::get().numCUs) * 8;
  return static_cast<int>(std::max<int64_t>(1, std::min(blocks, cap)));
}

int gpuStatsOf(const Runtime* ctx, vx355_gpu_stats* out) {
  if (!ctx || !out) {
    setLastError("NULL argument");
    return VX355_EINVAL;
  }
  out->busy_nanos = static_cast<int64_t>(ctx->busyNanos.load());
  out->h2d_bytes = static_cast<int64_t>(ctx->h2dBytes.load());
  out->d2h_bytes = static_cast<int64_t>(ctx->d2hBytes.load());
  out->input_bytes = static_cast<int64_t>(ctx->inputBytes.load());
  out->launches = static_cast<int64_t>(ctx->launchCount - ctx->launchesAtReset);
  out->reserved = 0;
  return VX355_OK;
}

void copyOut(void* dst, int32_t dstMem, const void* devSrc, size_t bytes) {
  if (!bytes) {
    return;
  }
  auto& rt = Runtime::get();
  HIP_OK(hipMemcpyAsync(dst, devSrc, bytes,
                        dstMem == VX355_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                        rt.stream));
  if (dstMem == VX355_MEM_HOST) {
    rt.d2hBytes += bytes;
    rt.sync();
  }
}

void copyOutAsync(void* dst, int32_t dstMem, const void* devSrc, size_t bytes) {
  if (!bytes) {
    return;
  }
  auto& rt = Runtime::get();
  HIP_OK(hipMemcpyAsync(dst, devSrc, bytes,
                        dstMem == VX355_MEM_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice,
                        rt.stream));
  if (dstMem == VX355_MEM_HOST) {
    rt.d2hBytes += bytes;
  }
}

void copyIn(void* devDst, const void* src, int32_t srcMem, size_t bytes) {
  if (!bytes) {
    return;
  }
  auto& rt = Runtime::get();
  HIP_OK(hipMemcpyAsync(devDst, src, bytes,
                        srcMem == VX355_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice,
                        rt.stream));
  if (srcMem == VX355_MEM_HOST) {
    rt.h2dBytes += bytes;
  }
}

namespace {
// One wave per string: desc = {source pointer, size, offset in 'out'} per string.
__global__ __launch_bounds__(64) void k_gather_strings(const uint64_t* desc, int64_t n, char* out) {
  for (int64_t i = blockIdx.x; i < n; i += gridDim.x) {
    const char* src = reinterpret_cast<const char*>(desc[i * 3]);
    const uint64_t size = desc[i * 3 + 1];
    char* dst = out + desc[i * 3 + 2];
    for (uint64_t b = threadIdx.x; b < size; b += 64) {
      dst[b] = src[b];
    }
  }
}
}  // namespace

void fetchLongStrings(char* views, int32_t n, std::vector<std::vector<char>>& keep) {
  std::vector<std::pair<int32_t, uint32_t>> longRows;  // row, size
  size_t total = 0;
  for (int32_t r = 0; r < n; ++r) {
    uint32_t size;
    std::memcpy(&size, views + static_cast<size_t>(r) * 16, 4);
    if (size > 12) {
      longRows.emplace_back(r, size);
      total += size;
    }
  }
  if (longRows.empty()) {
    return;
  }
  keep.emplace_back(total);
  char* dst = keep.back().data();
  const size_t m = longRows.size();
  std::vector<uint64_t> desc(m * 3);  // source pointer, size, offset
  size_t at = 0;
  for (size_t i = 0; i < m; ++i) {
    uint64_t src;
    std::memcpy(&src, views + static_cast<size_t>(longRows[i].first) * 16 + 8, 8);
    desc[i * 3] = src;
    desc[i * 3 + 1] = longRows[i].second;
    desc[i * 3 + 2] = at;
    char* hostPtr = dst + at;
    std::memcpy(views + static_cast<size_t>(longRows[i].first) * 16 + 8, &hostPtr, 8);
    at += longRows[i].second;
  }
  DevBuf dDesc, dBlob;
  uint64_t* devDesc = static_cast<uint64_t*>(dDesc.ensure(desc.size() * 8 + 64));
  char* devBlob = static_cast<char*>(dBlob.ensure(total + 64));
  copyIn(devDesc, desc.data(), VX355_MEM_HOST, desc.size() * 8);
  VX_LAUNCH("k_gather_strings", k_gather_strings, static_cast<int>(std::min<size_t>(m, 65535)), 64, 0, devDesc,
            static_cast<int64_t>(m), devBlob);
  copyOut(dst, VX355_MEM_HOST, devBlob, total);
}

char* DeviceState::allocPinned(size_t bytes, size_t* actual) {
  const size_t want = static_cast<size_t>(nextPow2(std::max<size_t>(bytes, 64 << 10)));
  {
    std::lock_guard<std::mutex> lock(memMutex);
    auto it = freePinned.lower_bound(want);
    if (it != freePinned.end() && it->first <= want * 2) {
      char* p = static_cast<char*>(it->second);
      *actual = it->first;
      cachedPinned -= it->first;
      freePinned.erase(it);
      return p;
    }
  }
  char* p = nullptr;
  bindHipDevice(device);
  HIP_OK(hipHostMalloc(reinterpret_cast<void**>(&p), want, hipHostMallocDefault));
  *actual = want;
  return p;
}

void DeviceState::releasePinned(char* p, size_t bytes) {
  if (!p) {
    return;
  }
  {
    std::lock_guard<std::mutex> lock(memMutex);
    if (alive && cachedPinned + bytes <= pinnedLimit) {
      freePinned.emplace(bytes, p);
      cachedPinned += bytes;
      return;
    }
  }
  (void)hipHostFree(p);
}

PinnedBuf::~PinnedBuf() {
  // Copies out of this block were synchronised by the consumer before it returned.
  if (p_) {
    ds_->releasePinned(p_, cap_);
  }
}

char* PinnedBuf::extend(size_t bytes) {
  if (size_ + bytes > cap_) {
    auto& rt = Runtime::get();
    size_t cap = 0;
    char* np = rt.allocPinned(std::max<size_t>(size_ + bytes, cap_ * 2), &cap);
    if (size_) {
      std::memcpy(np, p_, size_);
    }
    if (p_) {
      ds_->releasePinned(p_, cap_);
    }
    ds_ = rt.ds;
    p_ = np;
    cap_ = cap;
  }
  char* tail = p_ + size_;
  size_ += bytes;
  return tail;
}

bool HostCoalescer::append(const vx355_batch* batch, const std::vector<int32_t>& usedCols) {
  const int64_t n = batch->num_rows;
  if (thresholdRows <= 0 || n <= 0 || n >= thresholdRows) {
    return false;
  }
  for (int32_t c : usedCols) {
    if (c < 0) {
      continue;
    }
    if (c >= batch->num_cols) {
      return false;
    }
    const vx355_column& col = batch->cols[c];
    if (col.mem != VX355_MEM_HOST || kindWidth(col.type_kind) < 0) {
      return false;
    }
    if (isString(col.type_kind)) {
      // Non-inline strings point into buffers that die with the batch.
      const int64_t count =
          col.encoding == VX355_FLAT ? n : (col.encoding == VX355_CONSTANT ? 1 : col.base_size);
      const char* v = static_cast<const char*>(col.values);
      for (int64_t i = 0; i < count; ++i) {
        uint32_t size;
        std::memcpy(&size, v + i * 16, 4);
        if (size > 12) {
          return false;
        }
      }
    }
  }
  if (pending_.size() < static_cast<size_t>(batch->num_cols)) {
    pending_.resize(batch->num_cols);
  }
  const int64_t before = pendingRows_;
  for (int32_t c : usedCols) {
    if (c < 0) {
      continue;
    }
    auto& pc = pending_[c];
    const vx355_column& col = batch->cols[c];
    if (pc.kind >= 0 && static_cast<int64_t>(pc.valid.size()) == before + n) {
      continue;  // column listed twice: already appended
    }
    if (pc.kind < 0) {
      pc.kind = col.type_kind;
    } else if (pc.kind != col.type_kind) {
      VX_THROW(VX355_EINVAL, "column type changed between batches");
    }
    const int w = kindWidth(col.type_kind) == 0 ? 1 : kindWidth(col.type_kind);
    char* dst = pc.values.extend(static_cast<size_t>(n) * w);
    pc.valid.resize(static_cast<size_t>(before + n), 1);
    uint8_t* valid = pc.valid.data() + before;
    const char* src = static_cast<const char*>(col.values);
    const bool isBool = col.type_kind == VX355_BOOLEAN;
    if (col.encoding == VX355_FLAT && !col.nulls && !isBool) {
      std::memcpy(dst, src, static_cast<size_t>(n) * w);  // the common case: one copy
      continue;
    }
    for (int64_t r = 0; r < n; ++r) {
      const int64_t nullBit = col.encoding == VX355_CONSTANT ? 0 : r;
      const bool ok = !col.nulls || bitAt(col.nulls, nullBit);
      valid[r] = ok ? 1 : 0;
      if (!ok) {
        pc.anyNull = true;
        std::memset(dst + r * w, 0, w);
        continue;
      }
      const int64_t i =
          col.encoding == VX355_FLAT ? r : (col.encoding == VX355_CONSTANT ? 0 : col.indices[r]);
      if (isBool) {
        dst[r] = bitAt(static_cast<const uint64_t*>(col.values), i) ? 1 : 0;
      } else {
        std::memcpy(dst + r * w, src + i * w, w);
      }
    }
  }
  pendingRows_ += n;
  return true;
}

vx355_batch HostCoalescer::makeBatch(std::vector<vx355_column>* cols,
                                     std::vector<std::vector<uint64_t>>* bitmaps) {
  const int64_t n = pendingRows_;
  cols->assign(pending_.size(), vx355_column{});
  bitmaps->reserve(pending_.size() * 2);
  auto packBits = [&](const uint8_t* bytes) -> const uint64_t* {
    std::vector<uint64_t> words(static_cast<size_t>(ceilDiv(n, 64)), 0);
    for (int64_t i = 0; i < n; ++i) {
      if (bytes[i]) {
        words[i >> 6] |= 1ULL << (i & 63);
      }
    }
    bitmaps->push_back(std::move(words));
    return bitmaps->back().data();
  };
  for (size_t c = 0; c < pending_.size(); ++c) {
    auto& pc = pending_[c];
    vx355_column& col = (*cols)[c];
    col.encoding = VX355_FLAT;
    col.mem = VX355_MEM_HOST;
    if (pc.kind >= 0) {
      col.type_kind = pc.kind;
      col.values = pc.kind == VX355_BOOLEAN
          ? static_cast<const void*>(packBits(reinterpret_cast<const uint8_t*>(pc.values.data())))
          : static_cast<const void*>(pc.values.data());
      col.nulls = pc.anyNull ? packBits(pc.valid.data()) : nullptr;
    }
  }
  return vx355_batch{static_cast<int32_t>(n), static_cast<int32_t>(cols->size()), cols->data()};
}

void HostCoalescer::clear() {
  for (auto& pc : pending_) {
    pc.values.clear();
    pc.valid.clear();
    pc.anyNull = false;
  }
  pendingRows_ = 0;
}

namespace {
thread_local int tlsInlineStringsVerified = 0;
}
InlineStringsVerified::InlineStringsVerified() { ++tlsInlineStringsVerified; }
InlineStringsVerified::~InlineStringsVerified() { --tlsInlineStringsVerified; }
bool InlineStringsVerified::active() { return tlsInlineStringsVerified > 0; }

const void* DeviceBatch::stage(const void* src, size_t bytes, int32_t mem) {
  if (!src || mem == VX355_MEM_DEVICE) {
    return src;
  }
  auto buf = std::make_unique<DevBuf>();
  // Readers may touch up to 16 bytes past the end with vector loads.
  void* d = buf->ensure(bytes + 64);
  copyIn(d, src, VX355_MEM_HOST, bytes);
  staging_.push_back(std::move(buf));
  return d;
}

void DeviceBatch::load(const vx355_batch* batch, const std::vector<int32_t>& usedCols) {
  VX_CHECK_ARG(batch != nullptr, "batch is NULL");
  VX_CHECK_ARG(batch->num_rows >= 0 && batch->num_cols >= 0, "negative batch size");
  numRows_ = batch->num_rows;
  views_.assign(batch->num_cols, ColView{});
  used_.assign(batch->num_cols, 0);
  staging_.clear();
  hostTmp_.clear();
  for (int32_t c : usedCols) {
    if (c < 0) {
      continue;
    }
    VX_CHECK_ARG(c < batch->num_cols, "column index out of range");
    if (used_[c]) {
      continue;
    }
    used_[c] = 1;
    const vx355_column& col = batch->cols[c];
    const int width = kindWidth(col.type_kind);
    if (width < 0) {
      VX_THROW(VX355_EUNSUPPORTED, "unsupported type kind " + std::to_string(col.type_kind));
    }
    VX_CHECK_ARG(col.encoding >= VX355_FLAT && col.encoding <= VX355_DICTIONARY, "bad encoding");
    int64_t numValues = col.encoding == VX355_FLAT
        ? numRows_
        : (col.encoding == VX355_CONSTANT ? 1 : col.base_size);
    VX_CHECK_ARG(numValues >= 0, "negative base_size");
    ColView v;
    v.kind = col.type_kind;
    v.enc = col.encoding;
    size_t valueBytes = width == 0 ? static_cast<size_t>(ceilDiv(numValues, 64)) * 8
                                   : static_cast<size_t>(numValues) * width;
    if (numValues > 0) {
      VX_CHECK_ARG(col.values != nullptr, "column values is NULL");
    }
    if (col.mem == VX355_MEM_HOST && isString(col.type_kind) && numValues > 0 && !InlineStringsVerified::active()) {
      // Rewrite pointers of non-inline strings into a device blob.
      std::vector<char> views(valueBytes);
      std::memcpy(views.data(), col.values, valueBytes);
      std::vector<char> blob;
      std::vector<std::pair<int64_t, size_t>> fix;  // view index -> blob offset
      for (int64_t i = 0; i < numValues; ++i) {
        uint32_t size;
        std::memcpy(&size, views.data() + i * 16, 4);
        if (size > 12) {
          const char* p;
          std::memcpy(&p, views.data() + i * 16 + 8, 8);
          fix.emplace_back(i, blob.size());
          blob.insert(blob.end(), p, p + size);
        }
      }
      if (!fix.empty()) {
        auto buf = std::make_unique<DevBuf>();
        char* dblob = static_cast<char*>(buf->ensure(blob.size() + 64));
        copyIn(dblob, blob.data(), VX355_MEM_HOST, blob.size());
        for (auto& f : fix) {
          const char* p = dblob + f.second;
          std::memcpy(views.data() + f.first * 16 + 8, &p, 8);
        }
        hostTmp_.push_back(std::move(blob));
        staging_.push_back(std::move(buf));
      }
      hostTmp_.push_back(std::move(views));
      v.values = stage(hostTmp_.back().data(), valueBytes, VX355_MEM_HOST);
    } else {
      v.values = stage(col.values, valueBytes, col.mem);
    }
    int64_t nullBits = col.encoding == VX355_CONSTANT ? 1 : numRows_;
    v.nulls = static_cast<const uint64_t*>(
        stage(col.nulls, static_cast<size_t>(ceilDiv(nullBits, 64)) * 8, col.mem));
    if (col.encoding == VX355_DICTIONARY) {
      if (numRows_ > 0) {
        VX_CHECK_ARG(col.indices != nullptr, "dictionary column without indices");
      }
      v.indices = static_cast<const int32_t*>(
          stage(col.indices, static_cast<size_t>(numRows_) * 4, col.mem));
    }
    if (Runtime* rt = Runtime::tryGet()) {
      rt->inputBytes += valueBytes + (col.nulls ? static_cast<size_t>(ceilDiv(nullBits, 64)) * 8 : 0) +
          (col.encoding == VX355_DICTIONARY ? static_cast<size_t>(numRows_) * 4 : 0);
    }
    views_[c] = v;
  }
}

}  // namespace vx

using vx::Runtime;
using vx::DeviceState;

extern "C" {

int vx355_abi_version(void) { return VX355_ABI_VERSION; }

const char* vx355_last_error(void) { return vx::tlsLastError.c_str(); }

int vx355_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int vx355_init(int device) {
  try {
    std::lock_guard<std::mutex> lock(vx::gInitMutex);
    int n = 0;
    HIP_OK(hipGetDeviceCount(&n));
    if (device < 0 || device >= n || device >= vx::kMaxDevices) {
      VX_THROW(VX355_EINVAL, "vx355_init: no such device");
    }
    DeviceState* ds = vx::gDevices[device];
    if (!ds) {
      ds = new DeviceState();
      ds->device = device;
      vx::gDevices[device] = ds;
    }
    if (!ds->alive) {
      vx::bindHipDevice(device);
      hipDeviceProp_t prop;
      HIP_OK(hipGetDeviceProperties(&prop, device));
      ds->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
      ds->ldsPerBlock = prop.sharedMemPerBlock;
      ds->totalMem = prop.totalGlobalMem;
      // Scratch blocks released by operators are kept for reuse (hipMalloc/hipFree of
      // multi-GB blocks cost ~100 ms): up to 40 % of the device memory (one 10^9-row radix
      // aggregation over an open-addressing table releases ~75 GB of record buffers), or
      // VX355_CACHE_LIMIT_GB.
      ds->cacheLimit = static_cast<size_t>(prop.totalGlobalMem / 5 * 2);
      if (const char* e = std::getenv("VX355_CACHE_LIMIT_GB")) {
        ds->cacheLimit = static_cast<size_t>(std::strtoull(e, nullptr, 10)) << 30;
      }
      ds->alive = true;
      ds->defaultCtx = vx::newContext(ds, true);
    }
    // The first device initialised is the process default; the calling thread is
    // bound to the one it just initialised (vx355_set_device changes that).
    int expected = -1;
    vx::gDefaultDevice.compare_exchange_strong(expected, device);
    vx::tlsDevice = device;
    vx::bindHipDevice(device);
  VX_API_CATCH
}

int vx355_set_device(int device) {
  try {
    (void)vx::deviceState(device < 0 ? vx::kMaxDevices : device);  // throws unless initialised
    vx::tlsDevice = device;
    vx::bindHipDevice(device);
  VX_API_CATCH
}

int vx355_current_device(void) {
  Runtime* rt = Runtime::tryGet();
  return rt ? rt->device : -1;
}

void vx355_shutdown(void) {
  std::lock_guard<std::mutex> lock(vx::gInitMutex);
  for (int d = 0; d < vx::kMaxDevices; ++d) {
    DeviceState* ds = vx::gDevices[d];
    if (!ds || !ds->alive) {
      continue;
    }
    (void)hipSetDevice(d);
    ds->trimCache();
    {
      std::lock_guard<std::mutex> plock(ds->profMutex);
      for (auto& kv : ds->prof) {
        for (auto& ev : kv.second.events) {
          (void)hipEventDestroy(ev.first);
          (void)hipEventDestroy(ev.second);
        }
        for (auto& ev : kv.second.open) {
          (void)hipEventDestroy(ev.second);
        }
      }
      ds->prof.clear();
      for (auto e : ds->freeEvents) {
        (void)hipEventDestroy(e);
      }
      ds->freeEvents.clear();
    }
    Runtime* def = ds->defaultCtx;
    ds->defaultCtx = nullptr;
    std::vector<Runtime*> idle;
    {
      std::lock_guard<std::mutex> mlock(ds->memMutex);
      ds->alive = false;  // blocks released from now on go straight back to the driver
      idle.swap(ds->idleContexts);
      for (Runtime* c : idle) {
        ds->contexts.erase(c->id);
      }
    }
    for (Runtime* c : idle) {
      vx::freeContext(c);
    }
    Runtime::destroyContext(def);
  }
  vx::gDefaultDevice.store(-1);
  vx::tlsDevice = -1;
}

int vx355_set_memory_limit(int64_t bytes) {
  try {
    VX_CHECK_ARG(bytes >= 0, "a limit is >= 0 (0 = none)");
    DeviceState* ds = Runtime::get().ds;
    std::lock_guard<std::mutex> lock(ds->memMutex);
    ds->memoryLimit = static_cast<size_t>(bytes);
  VX_API_CATCH
}

int vx355_memory_usage(int64_t* in_use, int64_t* peak, int64_t* cached) {
  try {
    DeviceState* ds = Runtime::get().ds;
    std::lock_guard<std::mutex> lock(ds->memMutex);
    if (in_use) {
      *in_use = static_cast<int64_t>(ds->liveBytes);
    }
    if (peak) {
      *peak = static_cast<int64_t>(ds->peakLiveBytes);
      ds->peakLiveBytes = ds->liveBytes;
    }
    if (cached) {
      *cached = static_cast<int64_t>(ds->cachedBytes);
    }
  VX_API_CATCH
}

void* vx355_device_malloc(size_t bytes) {
  Runtime* rt = Runtime::tryGet();
  if (!rt) {
    vx::setLastError("vx355_init has not been called");
    return nullptr;
  }
  (void)hipSetDevice(rt->device);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    vx::setLastError(std::string("hipMalloc failed: ") + hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

void vx355_device_free(void* p) {
  if (p) {
    // Every entry point drains its stream before it returns: nothing of ours is in flight.
    (void)hipFree(p);
  }
}

int vx355_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  if (bytes) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, rt.stream));
    rt.sync();
  }
  VX_API_END
}

int vx355_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  if (bytes) {
    HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, rt.stream));
    rt.sync();
  }
  VX_API_END
}

int vx355_memset_d(void* dst, int value, size_t bytes) {
  VX_API_BEGIN
  auto& rt = Runtime::get();
  if (bytes) {
    HIP_OK(hipMemsetAsync(dst, value, bytes, rt.stream));
    rt.sync();
  }
  VX_API_END
}

int vx355_synchronize(void) {
  try {
    // Every context of the calling thread's device (operator handles included).
    DeviceState* ds = vx::deviceState(-1);
    std::vector<hipStream_t> streams;
    {
      std::lock_guard<std::mutex> lock(ds->memMutex);
      for (auto& kv : ds->contexts) {
        streams.push_back(kv.second->stream);
      }
    }
    for (hipStream_t s : streams) {
      HIP_OK(hipStreamSynchronize(s));
    }
  VX_API_CATCH
}

int vx355_stream_wait_event(void* stream, void* event) {
  try {
    VX_CHECK_ARG(stream && event, "NULL argument");
    HIP_OK(hipStreamWaitEvent(static_cast<hipStream_t>(stream), static_cast<hipEvent_t>(event), 0));
  VX_API_CATCH
}

void* vx355_default_stream(void) {
  Runtime* rt = Runtime::tryGet();
  return rt ? static_cast<void*>(rt->stream) : nullptr;
}

int vx355_profile_enable(int on) {
  try {
    DeviceState* ds = vx::deviceState(-1);
    (void)vx355_synchronize();
    std::lock_guard<std::mutex> lock(ds->profMutex);
    ds->profile = on != 0;
  VX_API_CATCH
}

namespace {
// Caller holds ds->profMutex and has synchronised the device's streams.
void drainProfile(DeviceState* ds) {
  for (auto& kv : ds->prof) {
    for (auto& ev : kv.second.events) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) {
        kv.second.doneMs += ms;
      }
      ds->freeEvents.push_back(ev.second);
      ds->freeEvents.push_back(ev.first);
    }
    kv.second.events.clear();
  }
}
}  // namespace

int vx355_profile_reset(void) {
  try {
    DeviceState* ds = vx::deviceState(-1);
    (void)vx355_synchronize();
    std::lock_guard<std::mutex> lock(ds->profMutex);
    drainProfile(ds);
    ds->prof.clear();
  VX_API_CATCH
}

int vx355_profile_get(const char* kernel, double* total_ms, int64_t* launches) {
  try {
    DeviceState* ds = vx::deviceState(-1);
    VX_CHECK_ARG(kernel && total_ms && launches, "NULL argument");
    (void)vx355_synchronize();
    std::lock_guard<std::mutex> lock(ds->profMutex);
    drainProfile(ds);
    auto it = ds->prof.find(kernel);
    if (it == ds->prof.end()) {
      *total_ms = 0;
      *launches = 0;
    } else {
      *total_ms = it->second.doneMs;
      *launches = it->second.launches;
    }
  VX_API_CATCH
}

int vx355_profile_names(char* buf, size_t cap) {
  try {
    DeviceState* ds = vx::deviceState(-1);
    VX_CHECK_ARG(buf && cap > 0, "NULL buffer");
    std::lock_guard<std::mutex> lock(ds->profMutex);
    std::string s;
    for (auto& kv : ds->prof) {
      if (!s.empty()) {
        s += "\n";
      }
      s += kv.first;
    }
    if (s.size() + 1 > cap) {
      s.resize(cap - 1);
    }
    std::memcpy(buf, s.c_str(), s.size() + 1);
  VX_API_CATCH
}

}  // extern "C"
