// sort_scan.hip — the two primitives graph preparation needs, hand-written for gfx950 (round 1 called rocPRIM for them):
//   * a STABLE least-significant-digit radix sort (8-bit digits) of 32-bit keys with 32-bit payloads — the destination sort
//     behind the plan (dst, edge position) and the row order — and of bare 64-bit keys (sort_edge_index's packed (s, t) pairs);
//   * an exclusive prefix sum of 32- or 64-bit integers (the digit-count tables of the sort, rowptr-like offset arrays).
// Off the timed path (a plan is built once per graph), but part of the drop-in all the same.
//
// One pass of the sort = three launches over wave-private tiles of 2 048 consecutive elements:
//   rs_hist    every wave counts its tile's digits in a private 256-entry LDS table (ds_add), writes them DIGIT-MAJOR
//              (counts[digit][tile]) so that one flat exclusive scan yields every (digit, tile) pair's first output slot;
//   scan       reduce-then-scan over 4 096-element chunks (wave shuffles + one LDS hop);
//   (passes whose digit is the same in every key are dropped beforehand: rs_varying_bits)
//   rs_scatter every wave walks its tile again 64 elements at a time IN ORDER: the lanes holding equal digits are found with 8
//              ballots (one per digit bit), a lane's rank among them is a popcount of the lower lanes — no sorting network, no
//              atomics — and the digit's running slot lives in an LDS table.  The four wave tiles of a block are first put into
//              digit order inside LDS and then streamed out as contiguous runs.  Element order inside a tile and tile order
//              inside a digit are both preserved, so every pass is stable and so is the whole sort.
#include <algorithm>

#include "common.h"
#include "sort_scan.h"

namespace gnnmp {

constexpr int RS_WT = 2048;        // elements per wave tile
constexpr int RS_WPB = 4;          // waves per block
constexpr int SC_CHUNK = 4096;     // elements per scan block (256 threads x 16)

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <class K>
__global__ void __launch_bounds__(64 * RS_WPB) rs_hist(const K *__restrict__ keys, size_t n, int shift, uint32_t *__restrict__ counts,
                                                       size_t n_tiles) {
    __shared__ uint32_t h[RS_WPB][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const size_t tile = (size_t)blockIdx.x * RS_WPB + w;
    if (tile >= n_tiles) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) h[w][lane + 64 * k] = 0;
    wave_lds_sync();
    const size_t base = tile * RS_WT;
    // the whole tile's keys first (32 independent loads per lane in flight), then the LDS counting: with the load inside the
    // counting loop every step paid a memory round trip (182 us per pass of 62 M 64-bit keys)
    K kreg[RS_WT / 64];
#pragma unroll
    for (int s = 0; s < RS_WT / 64; ++s) {
        const size_t i = base + (size_t)s * 64 + lane;
        kreg[s] = i < n ? keys[i] : (K)0;
    }
#pragma unroll
    for (int s = 0; s < RS_WT / 64; ++s) {
        const size_t i = base + (size_t)s * 64 + lane;
        if (i < n) atomicAdd(&h[w][(int)((kreg[s] >> shift) & 255)], 1u);
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) counts[(size_t)(lane + 64 * k) * n_tiles + tile] = h[w][lane + 64 * k];
}

// Scatter of one pass, block-cooperative: the four wave tiles of a block (8 192 consecutive elements) are first brought into
// digit order INSIDE LDS, then streamed out — elements of one digit leave as one contiguous run (32 elements = 128 bytes on
// average) instead of 64 separate 4-byte stores per wave step (the first version: 2.2 ms per pass of 64 M pairs, bound by store
// requests).
//   1. every wave learns its tile's digit counts from the scanned table itself (count = next offset - this offset: the table is
//      digit-major, so a (digit, tile)'s successor is (digit, tile + 1) or the next digit's first tile);
//   2. a 1 024-entry exclusive scan in LDS (digit-major, wave-minor) gives every (digit, wave) its first slot in the block tile;
//   3. each wave walks its 2 048 elements in order — equal digits among the 64 lanes by 8 ballots, rank by popcount, running
//      offsets in the wave's LDS table — and writes (key, payload) to their block-tile slots in LDS;
//   4. after one barrier all 256 threads stream the staged tile in slot order: slot p of digit d goes to
//      offsets[d][first tile of the block] + (p - first slot of d): stable across waves, tiles and blocks.
template <class K, bool PAIRS>
__global__ void __launch_bounds__(64 * RS_WPB) rs_scatter(const K *__restrict__ keys, const uint32_t *__restrict__ vals, size_t n,
                                                          int shift, const uint32_t *__restrict__ offsets, size_t n_tiles,
                                                          K *__restrict__ keys_out, uint32_t *__restrict__ vals_out) {
    constexpr int BT = RS_WT * RS_WPB;                       // elements per block tile
    extern __shared__ __align__(16) unsigned char rs_lds[];
    K *stage_k = reinterpret_cast<K *>(rs_lds);
    uint32_t *stage_v = reinterpret_cast<uint32_t *>(stage_k + BT);
    uint32_t *pos = stage_v + (PAIRS ? BT : 0);              // [256][RS_WPB] counts, then exclusive scan: first slot of (digit, wave)
    uint32_t *gstart = pos + 256 * RS_WPB;                   // [256] global first slot of digit d for this block
    __shared__ uint32_t wsum[RS_WPB];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const size_t tile0 = (size_t)blockIdx.x * RS_WPB;
    const size_t m = 256 * n_tiles;
    // 1. counts of (digit, wave) from the scanned table; global start of every digit for this block
    for (int e = tid; e < 256 * RS_WPB; e += 64 * RS_WPB) {
        const int d = e / RS_WPB, ww = e - d * RS_WPB;
        const size_t tile = tile0 + ww;
        uint32_t c = 0;
        if (tile < n_tiles) {
            const size_t idx = (size_t)d * n_tiles + tile;
            const uint32_t lo = offsets[idx], hi = idx + 1 < m ? offsets[idx + 1] : (uint32_t)n;
            c = hi - lo;
            if (ww == 0) gstart[d] = lo;
        }
        pos[e] = c;
    }
    __syncthreads();
    // 2. exclusive scan of the 1 024 counts (thread t owns entries 4 t .. 4 t + 3 = the four waves of digit t)
    {
        uint32_t c[RS_WPB], tot = 0;
#pragma unroll
        for (int k = 0; k < RS_WPB; ++k) { c[k] = pos[tid * RS_WPB + k]; tot += c[k]; }
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t pre = inc - tot;
        for (int k = 0; k < w; ++k) pre += wsum[k];
#pragma unroll
        for (int k = 0; k < RS_WPB; ++k) { pos[tid * RS_WPB + k] = pre; pre += c[k]; }
    }
    __syncthreads();
    // 3. stage this wave's tile in digit order
    const size_t tile = tile0 + w;
    if (tile < n_tiles) {
        const size_t base = tile * RS_WT;
        const uint64_t lower = (1ull << lane) - 1ull;
        // the tile's keys (and payloads) into registers first: 32 (64) independent loads per lane in flight.  With the load inside
        // the ranking loop every one of its 32 steps waited for memory between two LDS barriers (575 us per pass of 62 M keys).
        K kreg[RS_WT / 64];
        uint32_t vreg[PAIRS ? RS_WT / 64 : 1];
#pragma unroll
        for (int s = 0; s < RS_WT / 64; ++s) {
            const size_t i = base + (size_t)s * 64 + lane;
            kreg[s] = i < n ? keys[i] : (K)0;
            if (PAIRS) vreg[PAIRS ? s : 0] = i < n ? vals[i] : 0u;
        }
#pragma unroll
        for (int s = 0; s < RS_WT / 64; ++s) {
            const size_t i = base + (size_t)s * 64 + lane;
            const bool valid = i < n;
            const K key = kreg[s];
            const uint32_t val = PAIRS ? vreg[PAIRS ? s : 0] : 0u;
            const int d = (int)((key >> shift) & 255);
            uint64_t same = __ballot(valid);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint64_t mb = __ballot(valid && ((d >> b) & 1));
                same &= ((d >> b) & 1) ? mb : ~mb;
            }
            const int rank = __popcll(same & lower);
            const uint32_t first = pos[d * RS_WPB + w];
            wave_lds_sync();                               // every lane has read its digit's slot before a leader moves it
            if (valid && rank == 0) pos[d * RS_WPB + w] = first + (uint32_t)__popcll(same);
            wave_lds_sync();
            if (valid) {
                stage_k[first + rank] = key;
                if (PAIRS) stage_v[first + rank] = val;
            }
        }
    }
    __syncthreads();
    // 4. stream the block tile out in slot order.  After step 3 pos[d][w] = END of (d, w) = start of its successor, so the first
    //    slot of digit d in the block tile is pos[d - 1][RS_WPB - 1] (0 for d = 0).
    const size_t bt_base = tile0 * RS_WT;
    const int valid_n = (int)(n > bt_base ? (n - bt_base < (size_t)BT ? n - bt_base : (size_t)BT) : 0);
    for (int p = tid; p < valid_n; p += 64 * RS_WPB) {
        const K key = stage_k[p];
        const int d = (int)((key >> shift) & 255);
        const uint32_t dfirst = d == 0 ? 0u : pos[(d - 1) * RS_WPB + (RS_WPB - 1)];
        const size_t dst = (size_t)gstart[d] + (uint32_t)p - dfirst;
        keys_out[dst] = key;
        if (PAIRS) vals_out[dst] = stage_v[p];
    }
}

// ---- exclusive scan: out[i] = sum of in[0 .. i-1] ---------------------------------------------------------------------
template <class T>
__device__ __forceinline__ T wave_inclusive_scan(T v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const T u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    return v;
}

template <class T>
__global__ void __launch_bounds__(256) scan_reduce(const T *__restrict__ in, size_t n, T *__restrict__ block_sums) {
    __shared__ T ws[4];
    const size_t base = (size_t)blockIdx.x * SC_CHUNK;
    T s = 0;
    for (int k = 0; k < SC_CHUNK / 256; ++k) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < n) s += in[i];
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) ws[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// one block: exclusive scan of the block sums in place (any length: the block walks it in strides with a running carry)
template <class T>
__global__ void __launch_bounds__(1024) scan_block_sums(T *__restrict__ sums, size_t m) {
    __shared__ T ws[16];
    __shared__ T carry_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (size_t base = 0; base < m; base += 1024) {
        const size_t i = base + threadIdx.x;
        const T v = i < m ? sums[i] : (T)0;
        const T inc = wave_inclusive_scan(v, lane);
        if (lane == 63) ws[w] = inc;
        __syncthreads();
        T wave_off = 0;
        for (int k = 0; k < w; ++k) wave_off += ws[k];
        const T carry = carry_s;
        if (i < m) sums[i] = carry + wave_off + inc - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + wave_off + inc;
        __syncthreads();
    }
}

template <class T>
// (in and out may be the SAME array — the radix passes scan their digit table in place — so neither carries __restrict__: every
// thread reads its 16 elements into registers before it stores any, and without the qualifier the compiler has to keep it so)
__global__ void __launch_bounds__(256) scan_apply(const T *in, size_t n, const T *__restrict__ block_offsets, T *out) {
    __shared__ T ws[4];
    const size_t base = (size_t)blockIdx.x * SC_CHUNK;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // thread t owns the 16 consecutive elements base + 16 t .. + 15: sequential local scan, then a block scan of the thread totals
    T v[SC_CHUNK / 256];
    T tot = 0;
#pragma unroll
    for (int k = 0; k < SC_CHUNK / 256; ++k) {
        const size_t i = base + (size_t)threadIdx.x * (SC_CHUNK / 256) + k;
        v[k] = i < n ? in[i] : (T)0;
        tot += v[k];
    }
    const T inc = wave_inclusive_scan(tot, lane);
    if (lane == 63) ws[w] = inc;
    __syncthreads();
    T pre = block_offsets[blockIdx.x] + inc - tot;
    for (int k = 0; k < w; ++k) pre += ws[k];
#pragma unroll
    for (int k = 0; k < SC_CHUNK / 256; ++k) {
        const size_t i = base + (size_t)threadIdx.x * (SC_CHUNK / 256) + k;
        if (i < n) out[i] = pre;
        pre += v[k];
    }
}

// enqueue the three scan launches; sums: workspace of scan_blocks(n) entries
static inline size_t scan_blocks(size_t n) { return (n + SC_CHUNK - 1) / SC_CHUNK; }
template <class T>
static void exclusive_scan_enqueue(const T *in, T *out, size_t n, T *sums, hipStream_t stream) {
    const size_t nb = scan_blocks(n);
    scan_reduce<T><<<(unsigned)nb, 256, 0, stream>>>(in, n, sums);
    scan_block_sums<T><<<1, 1024, 0, stream>>>(sums, nb);
    scan_apply<T><<<(unsigned)nb, 256, 0, stream>>>(in, n, sums, out);
}
template <class T>
static int exclusive_scan_impl(const T *in, T *out, size_t n, hipStream_t stream, T *ws) {
    if (n == 0) return GNNMP_OK;
    if (ws) {   // the caller's (pooled) block-sum scratch: three launches, no allocation, no synchronisation
        exclusive_scan_enqueue<T>(in, out, n, ws, stream);
        hipError_t e0 = hipGetLastError();
        if (e0 != hipSuccess) return hip_fail(e0, "exclusive_scan");
        return GNNMP_OK;
    }
    T *sums = nullptr;
    GNNMP_HIP(hipMalloc((void **)&sums, sizeof(T) * scan_blocks(n)));
    exclusive_scan_enqueue<T>(in, out, n, sums, stream);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);     // the block sums are freed below
    (void)hipFree(sums);
    if (e != hipSuccess) return hip_fail(e, "exclusive_scan");
    return GNNMP_OK;
}

size_t exclusive_scan_workspace(size_t n) { return scan_blocks(n); }
int exclusive_scan_i64(const int64_t *in, int64_t *out, size_t n, hipStream_t stream, int64_t *ws) {
    return exclusive_scan_impl<int64_t>(in, out, n, stream, ws);
}
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, size_t n, hipStream_t stream, uint32_t *ws) {
    return exclusive_scan_impl<uint32_t>(in, out, n, stream, ws);
}

// ---- the sort ---------------------------------------------------------------------------------------------------------
// OR over all keys of (key XOR key[0]): the bits that are not the same in every key.  A pass whose eight bits are all constant
// moves nothing and is skipped — known before the first pass, from ONE read of the keys and one host round trip (the first
// version found out per pass, from that pass's scanned histogram: a histogram, a scan and a stream synchronisation for every
// dead byte of a packed (s, t) pair or of a 22-bit destination).
template <class K>
__global__ void __launch_bounds__(256) rs_varying_bits(const K *__restrict__ keys, size_t n, unsigned long long *__restrict__ out) {
    const K k0 = keys[0];
    unsigned long long v = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        v |= (unsigned long long)(keys[i] ^ k0);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v |= __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicOr(out, v);
}

template <class K, bool PAIRS>
static int radix_sort_impl(const K *keys_in, K *keys_out, const uint32_t *vals_in, uint32_t *vals_out, size_t n, int begin_bit,
                           int end_bit, hipStream_t stream) {
    if (n == 0) return GNNMP_OK;
    if (n >= ((size_t)1 << 32)) return fail(GNNMP_EUNSUPPORTED, "radix sort: %zu elements exceed the 32-bit offset tables", n);
    const int passes = std::max(1, (end_bit - begin_bit + 7) / 8);
    const size_t n_tiles = (n + RS_WT - 1) / RS_WT;
    const size_t m = 256 * n_tiles;
    K *ktmp = nullptr;
    uint32_t *vtmp = nullptr, *counts = nullptr, *sums = nullptr;
    unsigned long long *dvary = nullptr, hvary = 0;
    int rc = GNNMP_OK;
    // ONE allocation for every temporary (hipMalloc / hipFree cost 0.1 - 0.5 ms each at these sizes: five pairs of them were a
    // third of the products plan build): [ktmp | vtmp | counts | sums | dvary], each piece 256-byte aligned
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t b_k = up(sizeof(K) * n), b_v = PAIRS ? up(sizeof(uint32_t) * n) : 0, b_c = up(sizeof(uint32_t) * m),
                 b_s = up(sizeof(uint32_t) * scan_blocks(m));
    unsigned char *arena = nullptr;
    hipError_t e = hipMalloc((void **)&arena, b_k + b_v + b_c + b_s + 256);
    if (e != hipSuccess) {
        rc = hip_fail(e, "hipMalloc(radix sort temporaries)");
    } else {
        ktmp = reinterpret_cast<K *>(arena);
        vtmp = PAIRS ? reinterpret_cast<uint32_t *>(arena + b_k) : nullptr;
        counts = reinterpret_cast<uint32_t *>(arena + b_k + b_v);
        sums = reinterpret_cast<uint32_t *>(arena + b_k + b_v + b_c);
        dvary = reinterpret_cast<unsigned long long *>(arena + b_k + b_v + b_c + b_s);
    }
    const unsigned blocks = (unsigned)((n_tiles + RS_WPB - 1) / RS_WPB);
    const size_t scatter_lds = (size_t)RS_WT * RS_WPB * (sizeof(K) + (PAIRS ? sizeof(uint32_t) : 0)) + (256 * RS_WPB + 256) * sizeof(uint32_t);
    {
        hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(&rs_scatter<K, PAIRS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                            (int)scatter_lds);
        if (ea != hipSuccess && rc == GNNMP_OK) rc = hip_fail(ea, "hipFuncSetAttribute(rs_scatter)");
    }
    if (rc == GNNMP_OK) {
        e = hipMemsetAsync(dvary, 0, sizeof(unsigned long long), stream);
        if (e == hipSuccess) {
            const unsigned vb = (unsigned)std::min<size_t>((n + 255) / 256, (size_t)device_cus() * 8);
            rs_varying_bits<K><<<vb, 256, 0, stream>>>(keys_in, n, dvary);
            e = hipMemcpyAsync(&hvary, dvary, sizeof(hvary), hipMemcpyDeviceToHost, stream);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(stream);
        if (e != hipSuccess) rc = hip_fail(e, "radix sort: varying bits");
    }
    const K *ksrc = keys_in;
    const uint32_t *vsrc = vals_in;
    for (int p = 0; p < passes && rc == GNNMP_OK; ++p) {
        const int shift = begin_bit + 8 * p;
        if (((hvary >> shift) & 255ull) == 0) continue;    // every key has the same digit here: nothing moves
        rs_hist<K><<<blocks, 64 * RS_WPB, 0, stream>>>(ksrc, n, shift, counts, n_tiles);
        exclusive_scan_enqueue<uint32_t>(counts, counts, m, sums, stream);
        // destination: whichever of (out, tmp) does not hold the current source
        K *kdst = (ksrc == keys_out) ? ktmp : keys_out;
        uint32_t *vdst = (ksrc == keys_out) ? vtmp : vals_out;
        rs_scatter<K, PAIRS><<<blocks, 64 * RS_WPB, scatter_lds, stream>>>(ksrc, vsrc, n, shift, counts, n_tiles, kdst, vdst);
        e = hipGetLastError();
        if (e != hipSuccess) rc = hip_fail(e, "radix sort pass");
        ksrc = kdst;
        vsrc = vdst;
    }
    if (rc == GNNMP_OK && ksrc != keys_out) {              // every pass was trivial, or the last one landed in the temporary
        e = hipMemcpyAsync(keys_out, ksrc, sizeof(K) * n, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess && PAIRS) e = hipMemcpyAsync(vals_out, vsrc, sizeof(uint32_t) * n, hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) rc = hip_fail(e, "radix sort copy-out");
    }
    if (rc == GNNMP_OK) {
        e = hipStreamSynchronize(stream);                    // temporaries are freed below
        if (e != hipSuccess) rc = hip_fail(e, "radix sort");
    }
    if (arena) (void)hipFree(arena);
    return rc;
}

int radix_sort_pairs_u32(const uint32_t *keys_in, uint32_t *keys_out, const uint32_t *vals_in, uint32_t *vals_out, size_t n,
                         int begin_bit, int end_bit, hipStream_t stream) {
    return radix_sort_impl<uint32_t, true>(keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, stream);
}
int radix_sort_keys_u64(const uint64_t *keys_in, uint64_t *keys_out, size_t n, int begin_bit, int end_bit, hipStream_t stream) {
    return radix_sort_impl<uint64_t, false>(keys_in, keys_out, nullptr, nullptr, n, begin_bit, end_bit, stream);
}

}  // namespace gnnmp
