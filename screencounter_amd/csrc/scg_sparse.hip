// scg_sparse.hip -- combination spaces too large for a dense histogram: sort + run-length encode of the per-read
// combination stream on the device.
//
// countComboBarcodes with two pools of 50 000 barcodes has 2.5 x 10^9 possible combinations; the reference never
// materialises that space: its handlers collect one (first, second) tuple per matching read, radix-sort the tuples
// (kaori/utils.hpp:173-198, sort_combinations) and run-length encode them (src/utils.h:14-45, count_combinations).  The
// counting kernels do the same above the dense limit: instead of adding into cell first * n1 + second they store the
// read's combination as a 64-bit key (first << 32 | second, ~0 for "none") in a stream (ScgCounters::unit_pair); this
// file turns a batch's stream into (distinct key, count) runs -- rocPRIM's radix sort and run-length encode, the
// library routines for exactly this -- and the host merges the runs of the batches and devices (scg_api.cpp).
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "scg_launch.h"

namespace scg {

size_t sort_rle_scratch_bytes(size_t n) {
    size_t a = 0, b = 0;
    uint64_t* k = nullptr;
    uint32_t* c = nullptr;
    if (rocprim::radix_sort_keys(nullptr, a, k, k, n, 0, 64, nullptr) != hipSuccess) return 0;
    if (rocprim::run_length_encode(nullptr, b, k, static_cast<unsigned int>(n), k, c, c, nullptr) != hipSuccess) return 0;
    return (a > b ? a : b) + 256;
}

hipError_t launch_sort_rle(const uint64_t* d_keys, uint64_t* d_sorted, size_t n, uint64_t* d_unique, uint32_t* d_counts, uint32_t* d_runs,
                           void* d_scratch, size_t scratch_bytes, hipStream_t stream) {
    if (n == 0) return hipMemsetAsync(d_runs, 0, sizeof(uint32_t), stream);
    if (n > 0xFFFFFFFFull) return hipErrorInvalidValue;
    size_t a = scratch_bytes, b = scratch_bytes;
    hipError_t e = rocprim::radix_sort_keys(d_scratch, a, d_keys, d_sorted, n, 0, 64, stream);
    if (e != hipSuccess) return e;
    return rocprim::run_length_encode(d_scratch, b, d_sorted, static_cast<unsigned int>(n), d_unique, d_counts, d_runs, stream);
}

} // namespace scg
