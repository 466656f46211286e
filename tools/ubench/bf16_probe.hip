// bf16_probe.hip — issue rate of the bf16 MFMAs behind msplit.h: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 from 1 / 2 / 3 waves
// per SIMD, as dependent chains on ONE accumulator (what six plane products of one column block are) or round-robin over NACC
// independent accumulators.  Registers only: no memory, no LDS.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/ubench/bf16_probe.hip -o tools/ubench/libbf16_probe.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int CHAIN>
__global__ void __launch_bounds__(1024) probe32(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u + lane, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int i = 0; i < CHAIN; ++i)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int CHAIN>
__global__ void __launch_bounds__(1024) probe16(float *out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x4 acc[NACC];
#pragma unroll
    for (int c = 0; c < NACC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 a = {0x3f803f80u + lane, 0x3f803f80u, 0x3f003f00u, 0x3e803e80u}, b = {0x3f803f80u, 0x3f003f00u + lane, 0x3f803f80u, 0x3f803f80u};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int i = 0; i < CHAIN; ++i)
                acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[c], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NACC; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    if (s == 12345.678f) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// shape 32 | 16; nacc accumulators, each hit `chain` times in a row; returns MFMAs per wave per iteration
extern "C" int bf16_probe(int shape, int nacc, int chain, int waves, int iters, int blocks, float *out, hipStream_t stream) {
    dim3 grid(blocks), blk(64 * waves);
#define P32(N, C) if (shape == 32 && nacc == N && chain == C) { probe32<N, C><<<grid, blk, 0, stream>>>(out, iters); return N * C; }
#define P16(N, C) if (shape == 16 && nacc == N && chain == C) { probe16<N, C><<<grid, blk, 0, stream>>>(out, iters); return N * C; }
    P32(1, 6) P32(2, 6) P32(4, 6) P32(4, 1) P32(2, 1) P32(6, 1)
    P16(1, 6) P16(2, 6) P16(4, 6) P16(8, 6) P16(8, 1) P16(4, 1) P16(2, 1)
    return -1;
}
