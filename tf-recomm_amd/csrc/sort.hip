// sort.hip - stable key sort of batch positions by row id (the integer half of the
// deterministic segmented scatter).  Restates tf.unique + the batch-order walk of
// unsorted_segment_sum [TF1-lib] as "stable sort by id, then walk each run in order".
// rocPRIM's LSD radix sort is stable; only the low `end_bit` bits (enough for the table's
// row count) are sorted.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "svd_kernels.h"

namespace tfr {

size_t sort_temp_bytes(int64_t n, int end_bit) {
    size_t bytes = 0;
    const unsigned int* kin = nullptr;
    unsigned int* kout = nullptr;
    const int32_t* vin = nullptr;
    int32_t* vout = nullptr;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, (size_t)n, 0u,
                                             (unsigned)end_bit, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    return bytes ? bytes : 16;
}

hipError_t sort_pairs(void* temp, size_t temp_bytes, const int32_t* keys_in, int32_t* keys_out,
                      const int32_t* vals_in, int32_t* vals_out, int64_t n, int end_bit,
                      hipStream_t s) {
    return rocprim::radix_sort_pairs(temp, temp_bytes, reinterpret_cast<const unsigned int*>(keys_in),
                                     reinterpret_cast<unsigned int*>(keys_out), vals_in, vals_out,
                                     (size_t)n, 0u, (unsigned)end_bit, s);
}

}  // namespace tfr
