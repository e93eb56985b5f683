// Asynchronous boundary: the *_add_input_async entry points hand the batch to a worker thread of
// the handle and return; vx355_*_poll tells the shim how many batches are outstanding - what a
// Velox operator needs for isBlocked(ContinueFuture*) / needsInput() (exec/Operator.h:280-299):
// the Driver thread is free while the staging copies, the H2D transfers and the kernels of the
// batch run. Batches are processed in submission order by ONE worker per handle (the operator
// state machines are single-threaded, like the reference's); every other entry point of the handle
// first waits for the queue to drain, so the synchronous semantics of the rest of the ABI hold.
#include "common.h"

#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>

namespace vx {

struct AsyncQueue {
  std::mutex m;
  std::condition_variable wake, idle;
  std::deque<std::function<int(std::string*)>> tasks;
  std::thread worker;
  bool stop = false;
  int64_t submitted = 0, completed = 0;
  int firstError = VX355_OK;
  std::string errorText;

  void run() {
    std::unique_lock<std::mutex> lock(m);
    for (;;) {
      wake.wait(lock, [&] { return stop || !tasks.empty(); });
      if (tasks.empty()) {
        return;  // stop
      }
      auto task = std::move(tasks.front());
      tasks.pop_front();
      const bool skip = firstError != VX355_OK;  // a failed batch poisons the ones behind it
      lock.unlock();
      int status = VX355_OK;
      std::string text;
      if (!skip) {
        status = task(&text);
      }
      lock.lock();
      if (status != VX355_OK && firstError == VX355_OK) {
        firstError = status;
        errorText = text;
      }
      ++completed;
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

int64_t asyncSubmit(AsyncQueue* q, std::function<int(std::string*)> task) {
  std::lock_guard<std::mutex> lock(q->m);
  q->tasks.push_back(std::move(task));
  const int64_t ticket = ++q->submitted;
  q->wake.notify_one();
  return ticket;
}

void asyncPoll(AsyncQueue* q, int64_t* submitted, int64_t* completed) {
  std::lock_guard<std::mutex> lock(q->m);
  if (submitted) {
    *submitted = q->submitted;
  }
  if (completed) {
    *completed = q->completed;
  }
}

// Waits until every submitted batch has been processed and reports the first failure (status +
// message on the calling thread). The failure STAYS: the batches queued behind the failed one were
// skipped, so the operator's state is short of input - every later wait, and every entry point
// that drains the queue (add_input, no_more_input, get_output ...), keeps failing with it until the
// handle is destroyed. A shim that only checks no_more_input / get_output cannot miss it.
int asyncWait(AsyncQueue* q) {
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
  {
    std::unique_lock<std::mutex> lock(q->m);
    q->idle.wait(lock, [&] { return q->completed == q->submitted; });
    q->stop = true;
    q->wake.notify_all();
  }
  q->worker.join();
  delete q;
}

// A vx355_batch whose descriptor arrays belong to the task (the buffers they point to stay the
// caller's until the batch's ticket completes).
struct OwnedBatch {
  vx355_batch batch;
  std::vector<vx355_column> cols;
  explicit OwnedBatch(const vx355_batch* b) : batch(*b), cols(b->cols, b->cols + b->num_cols) { batch.cols = cols.data(); }
};

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

}  // namespace vx
