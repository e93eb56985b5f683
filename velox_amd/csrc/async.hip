// Asynchronous boundary: the *_add_input_async entry points hand the batch to the handle's queue
// and return; vx355_*_poll tells the shim how many batches are outstanding - what a Velox operator
// needs for isBlocked(ContinueFuture*) / needsInput() (exec/Operator.h:280-299): the Driver thread
// is free while the staging copies, the H2D transfers and the kernels of the batch run. Batches
// take effect in submission order through ONE worker per handle (the operator state machines are
// single-threaded, like the reference's); every other entry point of the handle first waits for
// the queue to drain, so the synchronous semantics of the rest of the ABI hold.
//
// Parallel ingest. Velox hands an operator 1 K - 10 K-row vectors in pageable host memory; what
// the GPU wants is a few large transfers from pinned memory. One thread copying the vectors into a
// pinned staging buffer moves ~13 GB/s - a fifth of PCIe Gen5 (160 MB in 12 ms). So
// the copies are spread over a pool of copier threads: the submitting thread only assigns the
// batch a place (chunk, row offset) in a ring of pinned chunk buffers, copiers fill the chunks
// concurrently, and the handle's worker takes a chunk when all its copies have landed: one
// flat host batch of up to 2^20 rows in pinned memory -> one DMA transfer per column and one launch.
// While the worker uploads and aggregates chunk i, the copiers fill chunk i + 1 ... i + 3.
// A batch's ticket completes when its CHUNK has been processed (its buffers are then no longer
// referenced). Eligible: host-resident FLAT columns without null bitmaps, fixed width (inline
// strings included: a copier that meets a string longer than 12 bytes marks the chunk and the worker
// feeds the chunk's batches one by one through the ordinary path instead); everything else takes
// the ordinary path, in order.
#include "common.h"

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace vx {

namespace {

// A vx355_batch whose descriptor arrays belong to the queue (the buffers they point to stay the
// caller's until the batch's ticket completes).
struct OwnedBatch {
  vx355_batch batch;
  std::vector<vx355_column> cols;
  explicit OwnedBatch(const vx355_batch* b) : batch(*b), cols(b->cols, b->cols + b->num_cols) { batch.cols = cols.data(); }
};

// Copier threads shared by every handle of the process (started on first use).
class CopierPool {
 public:
  static CopierPool& get() {
    static CopierPool* pool = new CopierPool();  // never destroyed: threads may outlive static destructors
    return *pool;
  }
  void submit(std::function<void()> task) {
    {
      std::lock_guard<std::mutex> lock(m_);
      tasks_.push_back(std::move(task));
    }
    wake_.notify_one();
  }
  int threads() const { return static_cast<int>(workers_.size()); }

 private:
  CopierPool() {
    int n = 8;
    if (const char* e = std::getenv("VX355_INGEST_THREADS")) {
      n = std::atoi(e);
    }
    const int hw = static_cast<int>(std::thread::hardware_concurrency());
    n = std::max(1, std::min(n, hw > 2 ? hw / 2 : 1));
    for (int i = 0; i < n; ++i) {
      workers_.emplace_back([this] { run(); });
      workers_.back().detach();
    }
  }
  void run() {
    std::unique_lock<std::mutex> lock(m_);
    for (;;) {
      wake_.wait(lock, [&] { return !tasks_.empty(); });
      auto task = std::move(tasks_.front());
      tasks_.pop_front();
      lock.unlock();
      task();
      lock.lock();
    }
  }
  std::mutex m_;
  std::condition_variable wake_;
  std::deque<std::function<void()>> tasks_;
  std::vector<std::thread> workers_;
};

constexpr int kIngestChunks = 4;

constexpr size_t kCopyGroup = 16;  // batches per copier task: one wake-up of a copier per 16 vectors

struct IngestChunk {
  std::vector<char*> data;   // per layout column: pinned
  std::vector<size_t> cap;
  int64_t rows = 0;          // rows assigned so far (submitter side)
  int64_t tickets = 0;
  // alive until the chunk has been processed
  std::vector<std::unique_ptr<OwnedBatch>> batches;
  std::vector<int64_t> offsets;   // row offset of batches[i] in the chunk
  size_t handedOut = 0;           // batches[0 .. handedOut) have a copier task
  std::mutex m;
  std::condition_variable copied;
  int64_t pending = 0;       // copy tasks in flight
  bool fallback = false;     // a copier met something the flat chunk cannot hold
  bool busy = false;
};

struct Ingest {
  DeviceState* ds = nullptr;
  int32_t batchCols = 0;           // num_cols of the batches this layout was made for
  std::vector<int32_t> cols;       // batch column indices that are copied
  std::vector<int32_t> kinds, widths;
  int64_t chunkRows = 0;
  IngestChunk chunk[kIngestChunks];
  int open = -1;
  std::mutex m;
  std::condition_variable freed;

  ~Ingest() {
    for (auto& c : chunk) {
      for (size_t i = 0; i < c.data.size(); ++i) {
        ds->releasePinned(c.data[i], c.cap[i]);
      }
    }
  }
};

}  // namespace

struct AsyncTask {
  std::function<int(std::string*)> fn;
  int64_t tickets;
  std::function<void()> cleanup;  // runs after fn - and instead of it when the task is skipped
  std::function<void(int)> done;  // runs after the tickets have been reported complete (asyncSubmit)
};

struct AsyncQueue {
  std::mutex m;
  std::condition_variable wake, idle;
  std::deque<AsyncTask> tasks;
  std::thread worker;
  bool stop = false;
  int64_t submitted = 0, completed = 0;
  int firstError = VX355_OK;
  std::string errorText;
  std::unique_ptr<Ingest> ingest;
  std::function<int(const vx355_batch*)> feed;  // how the handle takes a batch (parallel ingest)
  int64_t submittedAtLastPoll = -1;

  void run() {
    std::unique_lock<std::mutex> lock(m);
    for (;;) {
      wake.wait(lock, [&] { return stop || !tasks.empty(); });
      if (tasks.empty()) {
        return;  // stop
      }
      AsyncTask task = std::move(tasks.front());
      tasks.pop_front();
      const bool skip = firstError != VX355_OK;  // a failed batch poisons the ones behind it
      lock.unlock();
      int status = VX355_OK;
      std::string text;
      if (!skip) {
        status = task.fn(&text);
      }
      if (task.cleanup) {
        task.cleanup();
      }
      task.fn = nullptr;  // (releases what the task kept alive before the tickets are reported done)
      task.cleanup = nullptr;
      lock.lock();
      if (status != VX355_OK && firstError == VX355_OK) {
        firstError = status;
        errorText = text;
      }
      // The callback BEFORE the tickets count as completed: a caller that sees completed >= ticket (poll,
      // wait) may free what the callback touches - its argument, the page - right away.
      if (task.done) {
        const int seen = firstError;
        lock.unlock();
        task.done(seen);
        task.done = nullptr;
        lock.lock();
      }
      completed += task.tickets;
      if (completed == submitted) {
        idle.notify_all();
      }
    }
  }
};

AsyncQueue* asyncCreate() {
  auto* q = new AsyncQueue();
  q->worker = std::thread([q] { q->run(); });
  return q;
}

namespace {

void enqueue(AsyncQueue* q, std::function<int(std::string*)> fn, int64_t tickets, std::function<void()> cleanup) {
  std::lock_guard<std::mutex> lock(q->m);
  q->tasks.push_back(AsyncTask{std::move(fn), tickets, std::move(cleanup), nullptr});
  q->wake.notify_one();
}

int64_t reserveTicket(AsyncQueue* q) {
  std::lock_guard<std::mutex> lock(q->m);
  return ++q->submitted;
}

// Copies a group of batches into the chunk's pinned columns (a copier thread).
struct CopyItem {
  const OwnedBatch* batch;
  int64_t offset;
};
void copyBatches(Ingest* ing, IngestChunk* c, const std::vector<CopyItem>& items) {
  bool bad = false;
  for (const CopyItem& item : items) {
    const OwnedBatch& owned = *item.batch;
    const int64_t n = owned.batch.num_rows;
    for (size_t i = 0; i < ing->cols.size(); ++i) {
      const vx355_column& col = owned.cols[ing->cols[i]];
      const int w = ing->widths[i];
      const char* src = static_cast<const char*>(col.values);
      if (isString(ing->kinds[i])) {
        // inline StringViews are plain 16-byte values; a longer string points into the caller's buffers
        for (int64_t r = 0; r < n && !bad; ++r) {
          uint32_t size;
          std::memcpy(&size, src + r * 16, 4);
          bad = size > 12;
        }
      }
      std::memcpy(c->data[i] + item.offset * w, src, static_cast<size_t>(n) * w);
    }
  }
  std::lock_guard<std::mutex> lock(c->m);
  c->fallback = c->fallback || bad;
  if (--c->pending == 0) {
    c->copied.notify_all();
  }
}

// Gives the batches appended since the last call to the copier pool (all of them when 'all', else
// whole groups of kCopyGroup). Submitting thread only.
void handOutCopies(Ingest* ing, IngestChunk* c, bool all) {
  while (c->handedOut < c->batches.size() && (all || c->batches.size() - c->handedOut >= kCopyGroup)) {
    const size_t begin = c->handedOut;
    const size_t end = std::min(c->batches.size(), begin + kCopyGroup);
    c->handedOut = end;
    std::vector<CopyItem> items;
    items.reserve(end - begin);
    for (size_t b = begin; b < end; ++b) {
      items.push_back(CopyItem{c->batches[b].get(), c->offsets[b]});
    }
    {
      std::lock_guard<std::mutex> lock(c->m);
      ++c->pending;
    }
    CopierPool::get().submit([ing, c, items = std::move(items)] { copyBatches(ing, c, items); });
  }
}

// Hands the open chunk (if any) to the worker: its task waits for the chunk's copies, feeds the
// operator and gives the chunk back to the ring. Caller holds ing.m.
void sealOpenChunk(AsyncQueue* q, Ingest& ing, const std::function<int(const vx355_batch*)>& call) {
  if (ing.open < 0) {
    return;
  }
  IngestChunk* c = &ing.chunk[ing.open];
  ing.open = -1;
  Ingest* ingp = &ing;
  handOutCopies(ingp, c, true);
  enqueue(
      q,
      [c, ingp, call](std::string* text) -> int {
        {
          std::unique_lock<std::mutex> lock(c->m);
          c->copied.wait(lock, [&] { return c->pending == 0; });
        }
        int status = VX355_OK;
        if (!c->fallback) {
          std::vector<vx355_column> cols(static_cast<size_t>(ingp->batchCols), vx355_column{});
          for (size_t i = 0; i < ingp->cols.size(); ++i) {
            vx355_column& col = cols[ingp->cols[i]];
            col.type_kind = ingp->kinds[i];
            col.encoding = VX355_FLAT;
            col.mem = VX355_MEM_HOST;
            col.values = c->data[i];
          }
          vx355_batch flat{static_cast<int32_t>(c->rows), ingp->batchCols, cols.data()};
          InlineStringsVerified verified;  // (the copiers looked at every view: DeviceBatch::load need not)
          status = call(&flat);
        } else {
          for (auto& b : c->batches) {
            status = call(&b->batch);
            if (status != VX355_OK) {
              break;
            }
          }
        }
        if (status != VX355_OK) {
          *text = vx355_last_error();
        }
        return status;
      },
      c->tickets,
      [c, ingp] {
        // (also when the task is skipped behind a failure: the chunk's copies must have landed
        // before its buffers and the callers' batches are let go)
        {
          std::unique_lock<std::mutex> lock(c->m);
          c->copied.wait(lock, [&] { return c->pending == 0; });
        }
        {
          std::lock_guard<std::mutex> lock(ingp->m);
          c->rows = 0;
          c->tickets = 0;
          c->batches.clear();
          c->offsets.clear();
          c->handedOut = 0;
          c->fallback = false;
          c->busy = false;
        }
        ingp->freed.notify_all();
      });
}

bool layoutMatches(const Ingest& ing, const vx355_batch* batch) {
  if (batch->num_cols != ing.batchCols) {
    return false;
  }
  for (size_t i = 0; i < ing.cols.size(); ++i) {
    if (batch->cols[ing.cols[i]].type_kind != ing.kinds[i]) {
      return false;
    }
  }
  return true;
}

// Host-resident FLAT columns without nulls, fixed width; not a batch the ordinary path should take
// whole (>= a chunk) and not an empty one.
bool eligible(const vx355_batch* batch, const std::vector<int32_t>& usedCols, int64_t maxRows) {
  if (batch->num_rows <= 0 || (maxRows > 0 && batch->num_rows > maxRows / 4)) {
    return false;
  }
  bool any = false;
  for (int32_t c : usedCols) {
    if (c < 0) {
      continue;
    }
    if (c >= batch->num_cols) {
      return false;
    }
    const vx355_column& col = batch->cols[c];
    if (col.mem != VX355_MEM_HOST || col.encoding != VX355_FLAT || col.nulls || kindWidth(col.type_kind) <= 0 ||
        !col.values) {
      return false;
    }
    any = true;
  }
  return any;
}

}  // namespace

int64_t asyncSubmit(AsyncQueue* q, std::function<int(std::string*)> task, std::function<void(int)> done) {
  std::lock_guard<std::mutex> lock(q->m);
  q->tasks.push_back(AsyncTask{std::move(task), 1, nullptr, std::move(done)});
  const int64_t ticket = ++q->submitted;
  q->wake.notify_one();
  return ticket;
}

std::function<int(std::string*)> asyncBatchTask(const vx355_batch* batch, std::function<int(const vx355_batch*)> call) {
  auto owned = std::make_shared<OwnedBatch>(batch);
  return [owned, call](std::string* text) {
    const int status = call(&owned->batch);
    if (status != VX355_OK) {
      *text = vx355_last_error();
    }
    return status;
  };
}

void asyncSealIngest(AsyncQueue* q) {
  if (!q->ingest) {
    return;
  }
  std::lock_guard<std::mutex> lock(q->ingest->m);
  sealOpenChunk(q, *q->ingest, q->feed);
}

int64_t asyncSubmitBatch(AsyncQueue* q, DeviceState* ds, const vx355_batch* batch, const std::vector<int32_t>& usedCols,
                         std::function<int(const vx355_batch*)> call) {
  static const bool enabled = [] {
    const char* e = std::getenv("VX355_INGEST_PARALLEL");
    return !e || std::atoi(e) != 0;
  }();
  if (!q->feed) {
    q->feed = call;  // (the same callable for every batch of a handle)
  }
  Ingest* ing = q->ingest.get();
  const bool take = enabled && ds != nullptr && eligible(batch, usedCols, ing ? ing->chunkRows : 0) &&
      (!ing || layoutMatches(*ing, batch));
  if (!take) {
    asyncSealIngest(q);  // order: what was assigned to the open chunk comes first
    return asyncSubmit(q, asyncBatchTask(batch, std::move(call)));
  }
  if (!ing) {
    q->ingest = std::make_unique<Ingest>();
    ing = q->ingest.get();
    ing->ds = ds;
    ing->batchCols = batch->num_cols;
    size_t rowBytes = 0;
    std::vector<char> seen(static_cast<size_t>(batch->num_cols), 0);
    for (int32_t c : usedCols) {
      if (c < 0 || seen[c]) {
        continue;
      }
      seen[c] = 1;
      ing->cols.push_back(c);
      ing->kinds.push_back(batch->cols[c].type_kind);
      ing->widths.push_back(kindWidth(batch->cols[c].type_kind));
      rowBytes += static_cast<size_t>(ing->widths.back());
    }
    // chunks of up to 2^20 rows and 64 MB: large enough for full-rate DMA and launches, small
    // enough that four of them are a modest pinned footprint
    ing->chunkRows = std::max<int64_t>(1 << 16, std::min<int64_t>(1 << 20, (64LL << 20) / static_cast<int64_t>(rowBytes)));
    if (const char* e = std::getenv("VX355_INGEST_CHUNK_ROWS")) {
      ing->chunkRows = std::max<int64_t>(1024, std::strtoll(e, nullptr, 10));
    }
    if (batch->num_rows > ing->chunkRows / 4) {
      // (a first batch this large: the ordinary path takes such batches whole)
      q->ingest.reset();
      return asyncSubmit(q, asyncBatchTask(batch, std::move(call)));
    }
  }
  const int64_t n = batch->num_rows;
  IngestChunk* c = nullptr;
  int64_t offset = 0;
  {
    std::unique_lock<std::mutex> lock(ing->m);
    if (ing->open >= 0 && ing->chunk[ing->open].rows + n > ing->chunkRows) {
      sealOpenChunk(q, *ing, q->feed);
    }
    if (ing->open < 0) {
      int pick = -1;
      ing->freed.wait(lock, [&] {
        for (int i = 0; i < kIngestChunks; ++i) {
          if (!ing->chunk[i].busy) {
            pick = i;
            return true;
          }
        }
        return false;  // ring full: the GPU side is the bottleneck, the submitter waits for a chunk
      });
      IngestChunk& fresh = ing->chunk[pick];
      fresh.busy = true;
      if (fresh.data.empty()) {
        for (size_t i = 0; i < ing->cols.size(); ++i) {
          size_t cap = 0;
          char* p = ing->ds->allocPinned(static_cast<size_t>(ing->chunkRows) * ing->widths[i] + 64, &cap);
          fresh.data.push_back(p);
          fresh.cap.push_back(cap);
        }
      }
      ing->open = pick;
    }
    c = &ing->chunk[ing->open];
    offset = c->rows;
    c->rows += n;
    ++c->tickets;
    // Still under ing->m: a poll / get_stats / wait / destroy from another thread may seal this chunk
    // (sealOpenChunk) the moment the lock is dropped, and a sealed chunk must hold every batch whose
    // rows and ticket it counts. (Copier tasks carry pointers to the batches themselves; the worker
    // clears the list after pending == 0. handOutCopies only takes c->m, as it does under seal.)
    c->batches.push_back(std::make_unique<OwnedBatch>(batch));
    c->offsets.push_back(offset);
    handOutCopies(ing, c, false);
  }
  return reserveTicket(q);
}

void asyncPoll(AsyncQueue* q, int64_t* submitted, int64_t* completed) {
  bool quiet = false;
  {
    std::lock_guard<std::mutex> lock(q->m);
    if (submitted) {
      *submitted = q->submitted;
    }
    if (completed) {
      *completed = q->completed;
    }
    // Nothing new since the last poll and tickets outstanding: a caller that only polls must still
    // see the batches of the open chunk complete one day.
    quiet = q->submittedAtLastPoll == q->submitted && q->completed < q->submitted;
    q->submittedAtLastPoll = q->submitted;
  }
  if (quiet) {
    asyncSealIngest(q);
  }
}

// Waits until every submitted batch has been processed and reports the first failure (status +
// message on the calling thread). The failure STAYS: the batches queued behind the failed one were
// skipped, so the operator's state is short of input - every later wait, and every entry point
// that drains the queue (add_input, no_more_input, get_output ...), keeps failing with it until the
// handle is destroyed. A shim that only checks no_more_input / get_output cannot miss it.
int asyncWait(AsyncQueue* q) {
  asyncSealIngest(q);
  std::unique_lock<std::mutex> lock(q->m);
  q->idle.wait(lock, [&] { return q->completed == q->submitted; });
  const int status = q->firstError;
  if (status != VX355_OK) {
    setLastError(q->errorText);
  }
  return status;
}

// Waits for the queue without reporting anything (inspection entry points: get_stats).
void asyncQuiesce(AsyncQueue* q) {
  asyncSealIngest(q);
  std::unique_lock<std::mutex> lock(q->m);
  q->idle.wait(lock, [&] { return q->completed == q->submitted; });
}

// The sticky failure of the queue without waiting (VX355_OK = none so far).
int asyncFailed(AsyncQueue* q) {
  std::lock_guard<std::mutex> lock(q->m);
  if (q->firstError != VX355_OK) {
    setLastError(q->errorText);
  }
  return q->firstError;
}

void asyncDestroy(AsyncQueue* q) {
  if (!q) {
    return;
  }
  asyncSealIngest(q);
  {
    std::unique_lock<std::mutex> lock(q->m);
    q->idle.wait(lock, [&] { return q->completed == q->submitted; });
    q->stop = true;
    q->wake.notify_all();
  }
  q->worker.join();
  delete q;
}

}  // namespace vx
