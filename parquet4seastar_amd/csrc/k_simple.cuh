// BYTE_STREAM_SPLIT, PLAIN fixed-width and PLAIN BOOLEAN kernels.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

__device__ __forceinline__ uint32_t ld_u32_unaligned(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// 4x4 byte transpose: r[k] holds byte k of values 0..3 (one dword per plane);
// o[j] receives value j (bytes from planes 0..3).
__device__ __forceinline__ void transpose4(const uint32_t r[4], uint32_t o[4]) {
    uint32_t t0 = __byte_perm(r[0], r[1], 0x5140);
    uint32_t t1 = __byte_perm(r[0], r[1], 0x7362);
    uint32_t t2 = __byte_perm(r[2], r[3], 0x5140);
    uint32_t t3 = __byte_perm(r[2], r[3], 0x7362);
    o[0] = __byte_perm(t0, t2, 0x5410);
    o[1] = __byte_perm(t0, t2, 0x7632);
    o[2] = __byte_perm(t1, t3, 0x5410);
    o[3] = __byte_perm(t1, t3, 0x7632);
}

// 16 bytes that are read or written once: non-temporal, so they do not displace what the other kernels keep in L2
typedef uint32_t v4u32 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ld_stream16(const uint8_t *p) {
    const v4u32 v = __builtin_nontemporal_load(reinterpret_cast<const v4u32 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_stream16(uint8_t *p, const uint4 &x) {
    v4u32 v;
    v.x = x.x; v.y = x.y; v.z = x.z; v.w = x.w;
    __builtin_nontemporal_store(v, reinterpret_cast<v4u32 *>(p));
}

constexpr int BSS_THREADS = 256;
constexpr int BSS_VPT = 4;     // values per thread and group: one dword from each plane
constexpr int BSS_GROUPS = 2;  // groups per thread and piece: every load of a piece is issued before its first store
constexpr int BSS_PIECE = BSS_THREADS * BSS_VPT * BSS_GROUPS;      // values per piece (host side: same for both widths)

// BYTE_STREAM_SPLIT (ref: src/encoding.cc:364-394): out byte k of value i is in[k*N + i], N = page_size / K.
// A page is cut into pieces of BSS_PIECE values and every piece is one short-lived workgroup (blockIdx.x = piece,
// blockIdx.y / z = page; pieces step by gridDim.x when a page has more of them than the host guessed): the chip streams fastest
// with many small workgroups, each lane holding a few loads in flight, rather than with loops inside long ones.
// Each lane reads one dword from each of the K planes (coalesced 256 B per plane and wave) and writes K*4
// contiguous bytes (`v_perm` 4x4 byte transposes, no LDS: both sides are already coalesced).
template <int K>
__global__ __launch_bounds__(BSS_THREADS) void k_byte_stream_split(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, p4s_page_status *__restrict__ status, uint32_t n_pages) {
    const uint32_t page = blockIdx.z * gridDim.y + blockIdx.y;      // (no division on the way to the first load)
    const uint32_t sub = blockIdx.x, fan = gridDim.x;
    if (page >= n_pages) return;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    if (pv.size % K != 0) {
        if (sub == 0 && threadIdx.x == 0)
            write_status(&status[page], P4S_E_BSS_SIZE, 0, 0, 0, 0, 0);
        return;
    }
    const uint64_t N = pv.size / K;
    const uint64_t n = pd.num_values < N ? pd.num_values : N;
    if (sub == 0 && threadIdx.x == 0)
        write_status(&status[page], P4S_E_NONE, n, 0, 0, pv.size, 0);
    uint8_t *out = d_out + pd.out_offset * K;
    const bool out16 = (((uintptr_t)out) & 15u) == 0;
    // one group of BSS_VPT values: a dword from each plane in, K*4 contiguous bytes out
    auto load4 = [&](uint64_t i, uint32_t (&r)[K]) {
#pragma unroll
        for (int k = 0; k < K; ++k) r[k] = ld_u32_unaligned(pv.base + (uint64_t)k * N + i);
    };
    auto store4 = [&](uint64_t i, const uint32_t (&r)[K]) {
        if constexpr (K == 4) {
            uint32_t o[4];
            transpose4(r, o);
            if (out16) {
                st_stream16(out + i * 4, make_uint4(o[0], o[1], o[2], o[3]));
            } else {
                __builtin_memcpy(out + i * 4, o, 16);
            }
        } else {
            uint32_t lo[4], hi[4];
            transpose4(r, lo);
            transpose4(r + 4, hi);
            uint4 a = make_uint4(lo[0], hi[0], lo[1], hi[1]);
            uint4 b = make_uint4(lo[2], hi[2], lo[3], hi[3]);
            if (out16) {
                st_stream16(out + i * 8, a);
                st_stream16(out + i * 8 + 16, b);
            } else {
                __builtin_memcpy(out + i * 8, &a, 16);
                __builtin_memcpy(out + i * 8 + 16, &b, 16);
            }
        }
    };
    for (uint64_t p0 = (uint64_t)sub * BSS_PIECE; p0 < n; p0 += (uint64_t)fan * BSS_PIECE) {
        const uint64_t i0 = p0 + (uint64_t)threadIdx.x * BSS_VPT;
        if (p0 + BSS_PIECE <= n) {
            uint32_t r[BSS_GROUPS][K];
#pragma unroll
            for (int g = 0; g < BSS_GROUPS; ++g) load4(i0 + (uint64_t)g * BSS_THREADS * BSS_VPT, r[g]);
#pragma unroll
            for (int g = 0; g < BSS_GROUPS; ++g) store4(i0 + (uint64_t)g * BSS_THREADS * BSS_VPT, r[g]);
        } else {
            // the page's last, partial piece
            for (int g = 0; g < BSS_GROUPS; ++g) {
                const uint64_t i = i0 + (uint64_t)g * BSS_THREADS * BSS_VPT;
                if (i + BSS_VPT <= n) {
                    uint32_t r[K];
                    load4(i, r);
                    store4(i, r);
                } else {
                    for (uint64_t v = i; v < n; ++v)
                        for (int k = 0; k < K; ++k) out[v * K + k] = pv.base[(uint64_t)k * N + v];
                }
            }
        }
    }
}

constexpr int COPY_THREADS = 256;
constexpr int COPY_UNROLL = 2;                                     // 16-byte loads a lane holds in flight
constexpr int COPY_PIECE = COPY_THREADS * 16 * COPY_UNROLL;        // bytes per piece

// PLAIN fixed width (ref: src/encoding.cc:415-424): floor(size/elem) values, capped by n.  Pieces of COPY_PIECE
// bytes, one short-lived workgroup each (see k_byte_stream_split), non-temporal 16-byte loads and stores.
__global__ __launch_bounds__(COPY_THREADS) void k_plain_fixed(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, p4s_page_status *__restrict__ status, uint32_t elem, uint32_t n_pages) {
    const uint32_t page = blockIdx.z * gridDim.y + blockIdx.y;
    const uint32_t sub = blockIdx.x, fan = gridDim.x;
    if (page >= n_pages) return;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    uint64_t n = elem == 8 ? pv.size >> 3 : (elem == 4 ? pv.size >> 2 : pv.size / elem);
    if (pd.num_values < n) n = pd.num_values;
    const uint64_t bytes = n * elem;
    if (sub == 0 && threadIdx.x == 0)
        write_status(&status[page], P4S_E_NONE, n, 0, 0, bytes, 0);
    const uint8_t *src = pv.base;
    uint8_t *dst = d_out + pd.out_offset * elem;
    const uint32_t tid = threadIdx.x;
    const bool al16 = ((((uintptr_t)src) | ((uintptr_t)dst)) & 15u) == 0;
    const bool al4 = ((((uintptr_t)src) | ((uintptr_t)dst)) & 3u) == 0;
    for (uint64_t p0 = (uint64_t)sub * COPY_PIECE; p0 < bytes; p0 += (uint64_t)fan * COPY_PIECE) {
        const uint64_t end = p0 + COPY_PIECE < bytes ? p0 + COPY_PIECE : bytes;
        if (al16 && end - p0 == COPY_PIECE) {
            uint4 v[COPY_UNROLL];
#pragma unroll
            for (int u = 0; u < COPY_UNROLL; ++u) v[u] = ld_stream16(src + p0 + ((uint64_t)u * COPY_THREADS + tid) * 16);
#pragma unroll
            for (int u = 0; u < COPY_UNROLL; ++u) st_stream16(dst + p0 + ((uint64_t)u * COPY_THREADS + tid) * 16, v[u]);
        } else if (al16) {
            const uint64_t n16 = (end - p0) / 16;
            for (uint64_t i = tid; i < n16; i += COPY_THREADS)
                reinterpret_cast<uint4 *>(dst + p0)[i] = reinterpret_cast<const uint4 *>(src + p0)[i];
            for (uint64_t i = p0 + n16 * 16 + tid; i < end; i += COPY_THREADS) dst[i] = src[i];
        } else if (al4) {
            const uint64_t n4 = (end - p0) / 4;
            for (uint64_t i = tid; i < n4; i += COPY_THREADS)
                reinterpret_cast<uint32_t *>(dst + p0)[i] = reinterpret_cast<const uint32_t *>(src + p0)[i];
            for (uint64_t i = p0 + n4 * 4 + tid; i < end; i += COPY_THREADS) dst[i] = src[i];
        } else {
            for (uint64_t i = p0 + tid; i < end; i += COPY_THREADS) dst[i] = src[i];
        }
    }
}

// PLAIN BOOLEAN (ref: src/encoding.cc:426-428 -> BitReader::GetBatch(1, out, n)):
// value i is bit (i & 7) of byte i >> 3; n is clamped to the bits available.
__global__ __launch_bounds__(COPY_THREADS) void k_plain_boolean(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, p4s_page_status *__restrict__ status, uint32_t fan) {
    const uint32_t page = blockIdx.x / fan;
    const uint32_t sub = blockIdx.x % fan;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    uint64_t n = pv.size * 8;
    if (pd.num_values < n) n = pd.num_values;
    if (sub == 0 && threadIdx.x == 0)
        write_status(&status[page], P4S_E_NONE, n, 0, 0, (n + 7) / 8, 0);
    uint8_t *out = d_out + pd.out_offset;
    const bool out4 = (((uintptr_t)out) & 3u) == 0;
    const uint64_t step = (uint64_t)fan * COPY_THREADS * 8;
    for (uint64_t i = ((uint64_t)sub * COPY_THREADS + threadIdx.x) * 8; i < n; i += step) {
        uint32_t b = pv.base[i >> 3];
        uint32_t lo = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
        b >>= 4;
        uint32_t hi = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
        if (i + 8 <= n && out4) {
            reinterpret_cast<uint32_t *>(out + i)[0] = lo;
            reinterpret_cast<uint32_t *>(out + i)[1] = hi;
        } else {
            uint64_t both = ((uint64_t)hi << 32) | lo;
            for (int j = 0; j < 8 && i + j < n; ++j) out[i + j] = (uint8_t)(both >> (8 * j));
        }
    }
}

}  // namespace p4s
