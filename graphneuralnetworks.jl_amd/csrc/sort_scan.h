// sort_scan.h — hand-written device sort / scan primitives of the graph-preparation code (sort_scan.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace gnnmp {
// Stable LSD radix sort on key bits [begin_bit, end_bit), 8 bits per pass; inputs are not modified; temporaries are allocated
// and freed inside (the stream is synchronised before they are freed).  n < 2^32.
int radix_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, size_t n,
                         int begin_bit, int end_bit, hipStream_t stream);
int radix_sort_keys_u64(const uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit, int end_bit, hipStream_t stream);
// out[i] = in[0] + ... + in[i-1]  (in == out allowed).  ws: the block-sum scratch, exclusive_scan_workspace(n) elements from the
// caller's pool — then the scan is three launches and nothing else; ws = nullptr: allocated, synchronised on and freed inside (the
// once-per-graph plan build; the per-mini-batch entry points of graphprep.hip pass their pooled scratch: ADVICE r2).
size_t exclusive_scan_workspace(size_t n);
int exclusive_scan_i64(const int64_t *in, int64_t *out, size_t n, hipStream_t stream, int64_t *ws = nullptr);
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream, uint32_t *ws = nullptr);
}  // namespace gnnmp
