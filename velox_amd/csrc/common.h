// Host-side plumbing of libvx355: status/exception mapping, the runtime
// singleton (device, stream, pinned mailbox), growable HBM buffers, staging of
// vx355_batch descriptors into device-resident column views, and the
// HIP-event profiler behind vx355_profile_*.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vx355.h"
#include "device_utils.h"

namespace vx {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

void setLastError(const std::string& m);

// Asynchronous boundary (async.hip): one worker thread per handle runs its *_add_input_async
// batches in order; every other entry point of the handle drains the queue first (VX_ASYNC_DRAIN).
struct AsyncQueue;
AsyncQueue* asyncCreate();
// 'done' (optional) runs on the worker thread right BEFORE the task's ticket is reported complete - also when the
// task was skipped behind a failed one - with the status the queue holds then: once poll / wait report the ticket,
// the callback has returned and its argument may be freed.
int64_t asyncSubmit(AsyncQueue* q, std::function<int(std::string*)> task, std::function<void(int)> done = nullptr);
// The open chunk of the parallel ingest goes into the queue now (what is submitted next runs behind it).
void asyncSealIngest(AsyncQueue* q);
struct DeviceState;
// A batch for the handle: through the parallel ingest (async.hip) when it qualifies, else as an
// ordinary task; 'call' is how the handle takes a batch (on the queue's worker thread).
int64_t asyncSubmitBatch(AsyncQueue* q, DeviceState* ds, const vx355_batch* batch, const std::vector<int32_t>& usedCols,
                         std::function<int(const vx355_batch*)> call);
void asyncPoll(AsyncQueue* q, int64_t* submitted, int64_t* completed);
int asyncWait(AsyncQueue* q);
void asyncQuiesce(AsyncQueue* q);
int asyncFailed(AsyncQueue* q);
void asyncDestroy(AsyncQueue* q);
std::function<int(std::string*)> asyncBatchTask(const vx355_batch* batch, std::function<int(const vx355_batch*)> call);
#define VX_ASYNC_DRAIN(h)                                  \
  if ((h) != nullptr && (h)->aq != nullptr) {              \
    const int asyncStatus__ = ::vx::asyncWait((h)->aq);    \
    if (asyncStatus__ != VX355_OK) {                       \
      return asyncStatus__;                                \
    }                                                      \
  }

#define VX_THROW(status, msg) throw ::vx::Error((status), (msg))
#define VX_CHECK_ARG(cond, msg)         \
  do {                                  \
    if (!(cond)) {                      \
      VX_THROW(VX355_EINVAL, (msg));    \
    }                                   \
  } while (0)

void hipFail(hipError_t e, const char* what, const char* file, int line);
#define HIP_OK(expr)                                   \
  do {                                                 \
    hipError_t e__ = (expr);                           \
    if (e__ != hipSuccess) {                           \
      ::vx::hipFail(e__, #expr, __FILE__, __LINE__);   \
    }                                                  \
  } while (0)

// Every extern "C" entry point is wrapped: no exception crosses the ABI
// (include/vx355.h "Errors"). There is no library-wide lock: an entry point
// runs in an execution context (struct Runtime below: one HIP stream + one
// pinned mailbox on one GPU). Operator handles own their context
// (VX_API_BEGIN_CTX), handle-less calls use the default context of the calling
// thread's device and serialise on its mutex (VX_API_BEGIN).
#define VX_API_CATCH                                 \
  return VX355_OK;                                   \
  }                                                  \
  catch (const ::vx::Error& e) {                     \
    ::vx::setLastError(e.what());                    \
    return e.status;                                 \
  }                                                  \
  catch (const std::bad_alloc&) {                    \
    ::vx::setLastError("host out of memory");        \
    return VX355_ENOMEM;                             \
  }                                                  \
  catch (const std::exception& e) {                  \
    ::vx::setLastError(e.what());                    \
    return VX355_EINTERNAL;                          \
  }
#define VX_API_BEGIN \
  try {              \
    ::vx::ContextScope ctxScope__(nullptr);
#define VX_API_BEGIN_CTX(ctx) \
  try {                       \
    ::vx::ContextScope ctxScope__(ctx);
// Default context of one device (join-table functions: the table knows its device).
#define VX_API_BEGIN_DEV(device) \
  try {                          \
    ::vx::ContextScope ctxScope__(::vx::Runtime::defaultContext(device));
#define VX_API_END VX_API_CATCH
#define VX_CTX_OF(h) ((h) ? (h)->ctx : nullptr)

// Mailbox: pinned host memory the kernels write small results into (counts,
// flags, stats) so the host reads them after a stream sync without a D2H copy.
struct Mailbox {
  static constexpr int kWords = 512;
  uint64_t* host = nullptr;  // pinned, device-mapped
  uint64_t* dev = nullptr;   // device alias of host
};

struct ProfileEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> events;
  std::map<uint64_t, hipEvent_t> open;  // begin event of the launch in flight, per context
  double doneMs = 0;
  int64_t launches = 0;
};

// Per-GPU state shared by every context on that device: the HBM block cache,
// the pinned-block cache and the profiler, each behind its own mutex.
struct Runtime;
struct CachedBlock {
  void* p;
  uint64_t ownerCtx;   // context whose stream may still touch the block (0 = nobody)
  uint64_t ownerCall;  // that context's call number when the block was released
  uint64_t seq = 0;    // order of release: the cache evicts the blocks that have waited longest
};
struct DeviceState {
  int device = -1;
  bool alive = false;
  int numCUs = 256;
  size_t ldsPerBlock = 65536;
  size_t totalMem = 0;
  Runtime* defaultCtx = nullptr;

  std::mutex memMutex;
  // HBM block cache: operators are created and destroyed per query; their
  // tables and scratch buffers are recycled instead of going through
  // hipMalloc/hipFree (which synchronise the device).
  std::multimap<size_t, CachedBlock> freeBlocks;
  size_t cachedBytes = 0;
  uint64_t cacheSeq = 0;
  size_t cacheLimit = 16ULL << 30;
  // Bytes of blocks handed out and not yet released, and the cap on them (0 = none): what a Velox
  // MemoryPool's capacity is to the reference's operators (vx355_set_memory_limit).
  size_t liveBytes = 0;
  size_t peakLiveBytes = 0;
  size_t memoryLimit = 0;
  // Pinned host blocks (power-of-two sizes >= 64 KB): hipHostMalloc costs ~0.3 ms a
  // call, so released blocks are kept (up to pinnedLimit bytes) for the next operator.
  std::multimap<size_t, void*> freePinned;
  size_t cachedPinned = 0;
  size_t pinnedLimit = 2ULL << 30;
  // live contexts by id: a cached block released by a context whose call is still in
  // flight is only handed to another context after that stream has drained
  std::map<uint64_t, Runtime*> contexts;
  // Contexts of destroyed handles, kept for the next operator: creating a stream and a
  // pinned mailbox costs ~0.5 ms, operators are created per query.
  std::vector<Runtime*> idleContexts;

  std::mutex profMutex;
  bool profile = false;
  std::map<std::string, ProfileEntry> prof;
  std::vector<hipEvent_t> freeEvents;

  void* allocBlock(size_t bytes, size_t* actual);
  void freeBlock(void* p, size_t bytes);
  void trimCache();
  char* allocPinned(size_t bytes, size_t* actual);
  void releasePinned(char* p, size_t bytes);
};

// An execution context: one HIP stream and one pinned mailbox on one GPU. Every
// operator handle owns one (handles are single threaded, exec/Driver.cpp:538, so
// a context is never used by two threads at once); every device has a default
// one for handle-less entry points, guarded by callMutex.
struct Runtime {
  DeviceState* ds = nullptr;
  bool initialized = false;
  int device = -1;
  hipStream_t stream = nullptr;
  int numCUs = 256;
  size_t ldsPerBlock = 65536;
  Mailbox mail;
  uint64_t id = 0;
  uint64_t currentCall = 1;          // number of the API call in flight (or the next one)
  std::atomic<uint64_t> doneCalls{0};  // calls that have returned (their stream work is complete)
  std::recursive_mutex callMutex;    // default contexts only
  bool isDefault = false;
  uint64_t launchCount = 0;  // kernels launched through VX_LAUNCH in this context (see resetCounters in agg.hip)
  // What the context's operator cost the GPU side so far (vx355_*_get_gpu_stats; reset when the context
  // is handed to a new handle): nanoseconds between entering an entry point and its stream being
  // drained, bytes copied host -> HBM and HBM -> host, bytes of input columns handed to kernels.
  std::atomic<uint64_t> busyNanos{0}, h2dBytes{0}, d2hBytes{0}, inputBytes{0};
  uint64_t launchesAtReset = 0;
  void resetGpuStats() {
    busyNanos = 0;
    h2dBytes = 0;
    d2hBytes = 0;
    inputBytes = 0;
    launchesAtReset = launchCount;
  }
  bool& profile;                     // = ds->profile

  explicit Runtime(DeviceState* d) : ds(d), profile(d->profile) {}

  // The context of the calling thread: set by the entry point (ContextScope).
  static Runtime& get();
  static Runtime* tryGet();
  static Runtime* defaultContext(int device);  // device < 0: the calling thread's device
  static Runtime* createContext();             // on the calling thread's device
  static void destroyContext(Runtime* ctx);

  void requireInit() const {
    if (!initialized) {
      VX_THROW(VX355_EINVAL, "vx355_init has not been called");
    }
  }
  void sync() { HIP_OK(hipStreamSynchronize(stream)); }
  void* allocBlock(size_t bytes, size_t* actual) { return ds->allocBlock(bytes, actual); }
  char* allocPinned(size_t bytes, size_t* actual) { return ds->allocPinned(bytes, actual); }
  void releasePinned(char* p, size_t bytes) { ds->releasePinned(p, bytes); }
  hipEvent_t newEvent();
  void profBegin(const char* name);
  void profEnd(const char* name);
};

// Makes 'ctx' (nullptr: the default context of the thread's device, locked) the
// calling thread's context for the duration of an entry point. On exit the
// context's stream is drained: when an entry point returns, everything it
// queued has completed (outputs are usable, released blocks are reusable).
class ContextScope {
 public:
  explicit ContextScope(Runtime* ctx);
  ~ContextScope();
  ContextScope(const ContextScope&) = delete;
  ContextScope& operator=(const ContextScope&) = delete;

 private:
  Runtime* ctx_;
  Runtime* prev_;
  bool locked_ = false;
  bool outer_ = false;
  int64_t enteredNanos_ = 0;
};

// Launch on the library stream, bracketed by events when profiling is on.
#define VX_LAUNCH(name, kernel, grid, block, shmem, ...)                              \
  do {                                                                                \
    auto& rt__ = ::vx::Runtime::get();                                                \
    ++rt__.launchCount;                                                               \
    if (rt__.profile) rt__.profBegin(name);                                           \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), rt__.stream,         \
                       __VA_ARGS__);                                                  \
    if (rt__.profile) rt__.profEnd(name);                                             \
    HIP_OK(hipGetLastError());                                                        \
  } while (0)

// Growable device buffer (never shrinks; contents preserved on growth when
// asked). HBM is plentiful (288 GB): grow by doubling.
class DevBuf {
 public:
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p_(o.p_), cap_(o.cap_), ds_(o.ds_) { o.p_ = nullptr; o.cap_ = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept {
    if (this != &o) {
      release();
      p_ = o.p_;
      cap_ = o.cap_;
      ds_ = o.ds_;
      o.p_ = nullptr;
      o.cap_ = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); }
  void* ensure(size_t bytes, bool preserve = false, size_t preserveBytes = 0);
  template <typename T>
  T* as() const { return static_cast<T*>(p_); }
  void* ptr() const { return p_; }
  size_t capacity() const { return cap_; }
  void release();

 private:
  void* p_ = nullptr;
  size_t cap_ = 0;
  DeviceState* ds_ = nullptr;  // the GPU the block lives on
};

// While alive on a thread: host StringView columns handed to DeviceBatch::load hold inline strings
// only (the parallel ingest's copiers looked at every view), so load stages them as plain 16-byte
// values instead of scanning a million views per chunk for pointers to rewrite.
struct InlineStringsVerified {
  InlineStringsVerified();
  ~InlineStringsVerified();
  static bool active();
};

int kindWidth(int32_t kind);  // bytes per value; 0 for bit-packed BOOLEAN; -1 unknown
inline bool isIntLike(int32_t k) { return k >= VX355_BOOLEAN && k <= VX355_BIGINT; }
inline bool isString(int32_t k) { return k == VX355_VARCHAR || k == VX355_VARBINARY; }

// A vx355_batch made device-resident: host columns are copied into staging
// buffers owned by this object (H2D on the library stream); device columns are
// aliased. Non-inline strings of host columns are copied into a device blob
// and their StringView pointers rewritten.
class DeviceBatch {
 public:
  void load(const vx355_batch* batch, const std::vector<int32_t>& usedCols);
  const ColView& col(int32_t i) const { return views_.at(i); }
  int32_t numRows() const { return numRows_; }
  int32_t numCols() const { return static_cast<int32_t>(views_.size()); }
  bool used(int32_t i) const { return used_.at(i); }

 private:
  const void* stage(const void* src, size_t bytes, int32_t mem);
  std::vector<ColView> views_;
  std::vector<char> used_;
  std::vector<std::unique_ptr<DevBuf>> staging_;
  std::vector<std::vector<char>> hostTmp_;
  int32_t numRows_ = 0;
};

// Coalescing of small host batches. Velox hands operators 1 K - 10 K row vectors
// (core/QueryConfig.h:489 preferred_output_batch_rows); a launch wants >= 10^5
// rows. Rows of eligible batches (host memory, inline strings) are appended to
// host-side flat column buffers — dictionary and constant encodings are
// flattened on the way — and handed to the operator's normal path in one piece,
// the role CudfBatchConcat plays for the cuDF backend
// (experimental/cudf/CudfConfig.h:114-125).
// Growable pinned host buffer (hipHostMalloc): H2D copies from it are real
// DMA transfers at PCIe speed instead of the driver's bounce-buffer path.
class PinnedBuf {
 public:
  PinnedBuf() = default;
  PinnedBuf(const PinnedBuf&) = delete;
  PinnedBuf& operator=(const PinnedBuf&) = delete;
  PinnedBuf(PinnedBuf&& o) noexcept : p_(o.p_), cap_(o.cap_), size_(o.size_), ds_(o.ds_) { o.p_ = nullptr; o.cap_ = o.size_ = 0; }
  ~PinnedBuf();
  // Extends the used size by 'bytes' (contents preserved) and returns the new tail.
  char* extend(size_t bytes);
  char* data() const { return p_; }
  size_t size() const { return size_; }
  void clear() { size_ = 0; }

 private:
  char* p_ = nullptr;
  size_t cap_ = 0;
  size_t size_ = 0;
  DeviceState* ds_ = nullptr;
};

class HostCoalescer {
 public:
  int64_t thresholdRows = 1 << 18;  // <= 0 disables coalescing
  int64_t pendingRows() const { return pendingRows_; }
  // Appends the used columns of 'batch'; false = not eligible, nothing appended.
  bool append(const vx355_batch* batch, const std::vector<int32_t>& usedCols);
  // Calls consume(flat batch) on the pending rows (if any) and clears them.
  template <typename F>
  void flush(F&& consume) {
    if (pendingRows_ == 0) {
      return;
    }
    std::vector<vx355_column> cols;
    std::vector<std::vector<uint64_t>> bitmaps;
    vx355_batch flat = makeBatch(&cols, &bitmaps);
    pendingRows_ = 0;  // before the call: the consumer must not see pending rows
    try {
      consume(&flat);
    } catch (...) {
      clear();
      throw;
    }
    clear();
  }

 private:
  struct PendingCol {
    int32_t kind = -1;
    PinnedBuf values;            // flat values (BOOLEAN: one byte per row)
    std::vector<uint8_t> valid;  // one byte per row
    bool anyNull = false;
  };
  vx355_batch makeBatch(std::vector<vx355_column>* cols, std::vector<std::vector<uint64_t>>* bitmaps);
  void clear();
  std::vector<PendingCol> pending_;
  int64_t pendingRows_ = 0;
};

// Output columns of a PARTIAL / INTERMEDIATE aggregation that hold the sum half of an avg's
// (sum, count) pair (agg.hip); the count follows in the next column.
std::vector<int32_t> aggPartialAvgColumns(const vx355_agg* h);

// Block codecs of compressed PrestoPages (codec.hip; host work). kind = vx355_compression_kind.
// codecUncompress throws VX355_EUSER unless 'src' decodes to exactly dstLen bytes.
const char* codecName(int32_t kind);   // nullptr: no folly codec for the kind
void codecUncompress(int32_t kind, const unsigned char* src, size_t n, unsigned char* dst, size_t dstLen);
void codecCompress(int32_t kind, const unsigned char* src, size_t n, std::vector<unsigned char>& out);

// vx355_*_get_gpu_stats: the counters of one execution context.
int gpuStatsOf(const Runtime* ctx, vx355_gpu_stats* out);

// Exclusive scan of n u32 cells into n + 1 u64 offsets (last = total), on the library stream.
void scanU32ToU64(const uint32_t* in, int64_t n, uint64_t* out, DevBuf& scratch);

// Copies caller-visible results out of device scratch.
void copyOut(void* dst, int32_t dstMem, const void* devSrc, size_t bytes);
// Same without the stream synchronisation: the caller syncs once after a batch of copies.
void copyOutAsync(void* dst, int32_t dstMem, const void* devSrc, size_t bytes);
void copyIn(void* devDst, const void* src, int32_t srcMem, size_t bytes);

// Host output column of n 16-byte StringViews whose non-inline entries (size > 12) still hold DEVICE
// pointers: copies those strings to the host (one gather kernel + one copy) into a buffer appended
// to 'keep' and re-points the views. The caller keeps 'keep' alive until its next output call.
void fetchLongStrings(char* hostViews, int32_t n, std::vector<std::vector<char>>& keep);

inline int64_t ceilDiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
inline uint64_t nextPow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) {
    p <<= 1;
  }
  return p;
}
// Grid for a streaming kernel: enough blocks to fill 256 CUs several times
// over, capped so grid-stride loops amortise launch cost.
int streamGrid(int64_t items, int block, int perThread = 1);

}  // namespace vx
