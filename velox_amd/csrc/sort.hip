// Radix sort of (first input row, group row) pairs: the one place the library
// leans on rocPRIM. It runs once per aggregation, at output time, to put the
// groups into first-seen order (GroupingSet.cpp:828-839); it is not on the
// per-row path.
#include <cstring>

#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace vx {

void sortPairsU64U32(uint64_t* keys, uint32_t* vals, uint64_t* keysTmp, uint32_t* valsTmp,
                     size_t n, DevBuf& tmp, bool* resultInTmp, int endBit) {
  auto& rt = Runtime::get();
  *resultInTmp = false;
  if (n <= 1) {
    return;
  }
  size_t bytes = 0;
  HIP_OK(rocprim::radix_sort_pairs(nullptr, bytes, keys, keysTmp, vals, valsTmp, n, 0, endBit, rt.stream));
  void* scratch = tmp.ensure(bytes + 64);
  HIP_OK(rocprim::radix_sort_pairs(scratch, bytes, keys, keysTmp, vals, valsTmp, n, 0, endBit, rt.stream));
  *resultInTmp = true;
}

// Ascending sort of n u64 keys: in -> out (both n entries, distinct buffers).
void sortKeysU64(const uint64_t* in, uint64_t* out, size_t n, DevBuf& tmp) {
  auto& rt = Runtime::get();
  if (n == 0) {
    return;
  }
  size_t bytes = 0;
  HIP_OK(rocprim::radix_sort_keys(nullptr, bytes, in, out, n, 0, 64, rt.stream));
  void* scratch = tmp.ensure(bytes + 64);
  HIP_OK(rocprim::radix_sort_keys(scratch, bytes, in, out, n, 0, 64, rt.stream));
}

}  // namespace vx
