// gather_probe.hip — how fast can the chip fetch random ROWS of a row-major fp32 matrix, with nothing else to do?
// Each lane group of G lanes (16 bytes per lane: a row of 4 G floats) walks a list of random row ids and adds the rows into
// registers, U row loads in flight; one 16-byte store per lane at the end.  No index broadcast, no order, no epilogue: this is
// the ceiling the propagate / attention kernels are measured against (profiles/README.md).
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int U>
__global__ void __launch_bounds__(256) gather_probe_kernel(const float *__restrict__ x, const int32_t *__restrict__ ids,
                                                           int64_t n_ids, int log2g, int row_floats, int per_group,
                                                           float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int G = 1 << log2g;
    const int lig = lane & (G - 1);
    const int64_t group = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    const int64_t beg = group * per_group;
    if (beg >= n_ids) return;
    const int64_t end = beg + per_group < n_ids ? beg + per_group : n_ids;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool active = 4 * lig < row_floats;
    for (int64_t p = beg; p < end; p += U) {
        int c[U];
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) c[u] = ids[p + u < end ? p + u : end - 1];
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (active) v[u] = *reinterpret_cast<const float4 *>(x + (int64_t)c[u] * row_floats + 4 * lig);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (active && p + u < end) {
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
            }
        }
    }
    if (active) *reinterpret_cast<float4 *>(out + group * row_floats + 4 * lig) = acc;
}

extern "C" int gather_probe(const float *x, const int32_t *ids, int64_t n_ids, int log2g, int row_floats, int per_group, int U,
                            float *out, void *stream) {
    const int64_t groups = (n_ids + per_group - 1) / per_group;
    const int64_t threads = groups << log2g;
    const unsigned blocks = (unsigned)((threads + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    switch (U) {
        case 4: gather_probe_kernel<4><<<blocks, 256, 0, s>>>(x, ids, n_ids, log2g, row_floats, per_group, out); break;
        case 8: gather_probe_kernel<8><<<blocks, 256, 0, s>>>(x, ids, n_ids, log2g, row_floats, per_group, out); break;
        case 16: gather_probe_kernel<16><<<blocks, 256, 0, s>>>(x, ids, n_ids, log2g, row_floats, per_group, out); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Variant: PERSISTENT lane groups.  A lane group walks lists g, g + stride, g + 2 stride, ... of `per_group` rows each (what a
// row kernel whose waves loop over destinations would do); with PREFETCH the first batch of ids of the next list is loaded
// before the current list's last batch is consumed.  Tells how much of the 26-vs-208 gap is wave turnover and how much is the
// dependent ids -> rows chain at the start of every list.
template <int U, bool PREFETCH>
__global__ void __launch_bounds__(256) gather_probe_persistent_kernel(const float *__restrict__ x, const int32_t *__restrict__ ids,
                                                                      int64_t n_ids, int log2g, int row_floats, int per_group,
                                                                      float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int G = 1 << log2g;
    const int lig = lane & (G - 1);
    const int64_t group0 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> log2g;
    const int64_t stride = ((int64_t)gridDim.x * blockDim.x) >> log2g;
    const int64_t n_groups = (n_ids + per_group - 1) / per_group;
    const bool active = 4 * lig < row_floats;
    int cn[U];
    if (PREFETCH && group0 < n_groups) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t p = group0 * per_group + u;
            cn[u] = ids[p < n_ids ? p : n_ids - 1];
        }
    }
    for (int64_t group = group0; group < n_groups; group += stride) {
        const int64_t beg = group * per_group;
        const int64_t end = beg + per_group < n_ids ? beg + per_group : n_ids;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t p = beg; p < end; p += U) {
            int c[U];
            float4 v[U];
            if (PREFETCH && p == beg) {
#pragma unroll
                for (int u = 0; u < U; ++u) c[u] = cn[u];
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) c[u] = ids[p + u < end ? p + u : end - 1];
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (active) v[u] = *reinterpret_cast<const float4 *>(x + (int64_t)c[u] * row_floats + 4 * lig);
            if (PREFETCH && p + U >= end && group + stride < n_groups) {   // last batch of this list: the next list's first ids
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int64_t q = (group + stride) * per_group + u;
                    cn[u] = ids[q < n_ids ? q : n_ids - 1];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (active && p + u < end) {
                    acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                }
            }
        }
        if (active) *reinterpret_cast<float4 *>(out + group * row_floats + 4 * lig) = acc;
    }
}

extern "C" int gather_probe_persistent(const float *x, const int32_t *ids, int64_t n_ids, int log2g, int row_floats, int per_group,
                                       int prefetch, int blocks, float *out, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (prefetch)
        gather_probe_persistent_kernel<8, true><<<blocks, 256, 0, s>>>(x, ids, n_ids, log2g, row_floats, per_group, out);
    else
        gather_probe_persistent_kernel<8, false><<<blocks, 256, 0, s>>>(x, ids, n_ids, log2g, row_floats, per_group, out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
