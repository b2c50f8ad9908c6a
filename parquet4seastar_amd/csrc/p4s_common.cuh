// Device-side helpers shared by the gfx950 page-decode kernels: wave64 DPP scans,
// bit-field extraction from a dword stream, bounds-checked byte reads and the
// per-page status record.  Everything here is wave64 / CDNA4 specific.
#pragma once

// Experiment hooks.  The kernels carry compile-time switches (P4S_EXP_*) that take work out of them or read cycle
// counters inside them, for measurements; a library built with one of them does not decode correctly or reports
// timings in place of status fields.  They only compile together with -DP4S_EXPERIMENTS (tools/build_variant.py adds
// it), so that no build flag carried over by accident changes the product library.
#if !defined(P4S_EXPERIMENTS) && (defined(P4S_EXP_TIMING) || defined(P4S_EXP_TIMING_WALK) || defined(P4S_EXP_TIMING_WALK2) || \
    defined(P4S_EXP_TIMING_DECODE) || defined(P4S_EXP_FILLONLY) || defined(P4S_EXP_NOSTORE) || defined(P4S_EXP_NOPRIO) || \
    defined(P4S_EXP_NOSHIFT) || defined(P4S_EXP_NOGENERAL) || defined(P4S_EXP_NODECODE) || defined(P4S_EXP_DUMMYDECODE) || \
    defined(P4S_EXP_ALIGNSTORE) || defined(P4S_EXP_RLE_NOEXPAND) || defined(P4S_EXP_SNAPPY_PARSEONLY) || \
    defined(P4S_EXP_SNAPPY_NOPAIR) || defined(P4S_EXP_SNAPPY_NOLIT) || defined(P4S_EXP_SNAPPY_NOCOPY) || defined(P4S_EXP_SNAPPY_NOBATCH))
#error "P4S_EXP_* switches are for experiment builds: add -DP4S_EXPERIMENTS (tools/build_variant.py does)"
#endif

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/p4s_gpu.h"

namespace p4s {

constexpr int WAVE = 64;
constexpr int32_t E_OUT_CAPACITY = 23;   // P4S_E_OUT_CAPACITY: an output slot of the caller is too small

// ---------------------------------------------------------------------------
// DPP wave64 inclusive scans.  row_shr:n moves data n lanes up inside a row of
// 16 lanes (source out of the row -> 0 because old = 0 and bound_ctrl is off);
// row_bcast:15 adds lane 15 of each row to the next row, row_bcast:31 adds lane
// 31 to rows 2 and 3.  Six steps give the full 64-lane inclusive prefix sum.
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xF, false);
}

__device__ __forceinline__ uint32_t wave_iscan(uint32_t v) {
    v += dpp_mov<0x111, 0xF>(v);
    v += dpp_mov<0x112, 0xF>(v);
    v += dpp_mov<0x114, 0xF>(v);
    v += dpp_mov<0x118, 0xF>(v);
    v += dpp_mov<0x142, 0xA>(v);
    v += dpp_mov<0x143, 0xC>(v);
    return v;
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_mov64(uint64_t x) {
    uint32_t lo = dpp_mov<CTRL, ROW_MASK>((uint32_t)x);
    uint32_t hi = dpp_mov<CTRL, ROW_MASK>((uint32_t)(x >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_iscan(uint64_t v) {
    v += dpp_mov64<0x111, 0xF>(v);
    v += dpp_mov64<0x112, 0xF>(v);
    v += dpp_mov64<0x114, 0xF>(v);
    v += dpp_mov64<0x118, 0xF>(v);
    v += dpp_mov64<0x142, 0xA>(v);
    v += dpp_mov64<0x143, 0xC>(v);
    return v;
}

// inclusive max-scan (run-index propagation in the hybrid expander)
__device__ __forceinline__ uint32_t wave_imaxscan(uint32_t v) {
    uint32_t t;
    t = dpp_mov<0x111, 0xF>(v); v = t > v ? t : v;
    t = dpp_mov<0x112, 0xF>(v); v = t > v ? t : v;
    t = dpp_mov<0x114, 0xF>(v); v = t > v ? t : v;
    t = dpp_mov<0x118, 0xF>(v); v = t > v ? t : v;
    t = dpp_mov<0x142, 0xA>(v); v = t > v ? t : v;
    t = dpp_mov<0x143, 0xC>(v); v = t > v ? t : v;
    return v;
}

// The value of the lane before (lane 0 gets 0): DPP wavefront shift right by one.
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return dpp_mov<0x138, 0xF>(v); }
__device__ __forceinline__ uint64_t wave_shr1(uint64_t v) { return dpp_mov64<0x138, 0xF>(v); }

// N independent inclusive sum scans, written step by step across the N so that their latencies overlap
template <int N>
__device__ __forceinline__ void wave_iscan_n(uint32_t (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x111, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x112, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x114, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x118, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x142, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov<0x143, 0xC>(v[i]);
}
template <int N>
__device__ __forceinline__ void wave_iscan_n(uint64_t (&v)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x111, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x112, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x114, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x118, 0xF>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x142, 0xA>(v[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] += dpp_mov64<0x143, 0xC>(v[i]);
}

// N independent inclusive max-scans, written step by step across the N so that their latencies overlap
template <int N>
__device__ __forceinline__ void wave_imaxscan_n(uint32_t (&v)[N]) {
    uint32_t t;
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x111, 0xF>(v[i]); v[i] = t > v[i] ? t : v[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x112, 0xF>(v[i]); v[i] = t > v[i] ? t : v[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x114, 0xF>(v[i]); v[i] = t > v[i] ? t : v[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x118, 0xF>(v[i]); v[i] = t > v[i] ? t : v[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x142, 0xA>(v[i]); v[i] = t > v[i] ? t : v[i]; }
#pragma unroll
    for (int i = 0; i < N; ++i) { t = dpp_mov<0x143, 0xC>(v[i]); v[i] = t > v[i] ? t : v[i]; }
}

// A workgroup barrier for waves that hand each other data through LDS only: it waits for this wave's LDS operations
// but, unlike __syncthreads(), not for its global loads and stores (every store in flight would have to be acknowledged
// by memory first, thousands of cycles for a wave that streams its output).  What the waves exchange through global
// memory is NOT ordered by it.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// The same value, but opaque to the optimiser: what is computed from it inside a loop stays inside the loop
// instead of being hoisted and held in a register (or spilled) for the whole kernel.
__device__ __forceinline__ uint32_t opaque(uint32_t v) {
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ uint32_t bcast_u32(uint32_t v, int src_lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, src_lane);
}
__device__ __forceinline__ uint64_t bcast_u64(uint64_t v, int src_lane) {
    uint32_t lo = bcast_u32((uint32_t)v, src_lane), hi = bcast_u32((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t first_u32(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
}
__device__ __forceinline__ uint64_t first_u64(uint64_t v) {
    uint32_t lo = first_u32((uint32_t)v), hi = first_u32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t bcast_value(uint32_t v, int src_lane) { return bcast_u32(v, src_lane); }
__device__ __forceinline__ uint64_t bcast_value(uint64_t v, int src_lane) { return bcast_u64(v, src_lane); }
// A value every lane of the wave holds, moved to scalar registers.
__device__ __forceinline__ uint32_t uniform_value(uint32_t v) { return first_u32(v); }
__device__ __forceinline__ uint64_t uniform_value(uint64_t v) { return first_u64(v); }

// ---------------------------------------------------------------------------
// Bit fields.  `words` is a dword view of an LSB-first bit string; `bit` is the
// position of the field's first bit.  Reads up to two (32-bit result) or three
// (64-bit result) consecutive dwords: the caller keeps that much slack mapped.
// ---------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mask32(uint32_t v, uint32_t w) {
    return w >= 32 ? v : (v & ((1u << w) - 1u));
}
__device__ __forceinline__ uint64_t mask64(uint64_t v, uint32_t w) {
    return w >= 64 ? v : (v & ((1ull << w) - 1ull));
}

template <typename WordPtr>
__device__ __forceinline__ uint32_t field32(WordPtr words, uint32_t bit, uint32_t w) {
    uint32_t i = bit >> 5, sh = bit & 31u;
    uint32_t d0 = words[i], d1 = words[i + 1];
    return mask32(__funnelshift_r(d0, d1, sh), w);
}

template <typename WordPtr>
__device__ __forceinline__ uint64_t field64(WordPtr words, uint32_t bit, uint32_t w) {
    uint32_t i = bit >> 5, sh = bit & 31u;
    uint32_t d0 = words[i], d1 = words[i + 1], d2 = words[i + 2];
    uint32_t lo = __funnelshift_r(d0, d1, sh);
    uint32_t hi = __funnelshift_r(d1, d2, sh);
    return mask64(((uint64_t)hi << 32) | lo, w);
}

// ---------------------------------------------------------------------------
// Debug build (-DP4S_DEBUG_BOUNDS): the global accesses of the delta and string kernels are checked against the
// buffers of the call (set by the host before the launch); an access outside all of them is counted, its address
// and site number are kept, and it is skipped instead of executed.  P4S_LD / P4S_ST are plain accesses otherwise.
// ---------------------------------------------------------------------------
#ifdef P4S_DEBUG_BOUNDS
struct DbgBounds {
    uint64_t lo[6], hi[6];
    unsigned long long hits;
    unsigned long long rec[16][4];      // address, site | size << 32, workgroup, thread of the first 16
};
__device__ DbgBounds g_dbg;
__device__ __forceinline__ bool dbg_ok(const void *p, uint32_t n, uint32_t site) {
#ifdef __HIP_DEVICE_COMPILE__
    if (__builtin_amdgcn_is_shared((const __attribute__((address_space(0))) void *)p)) return true;
#endif
    const uint64_t a = (uint64_t)(uintptr_t)p;
    for (int r = 0; r < 6; ++r)
        if (a >= g_dbg.lo[r] && a + n <= g_dbg.hi[r]) return true;
    const unsigned long long k = atomicAdd(&g_dbg.hits, 1ull);
    if (k < 16) {
        g_dbg.rec[k][0] = a; g_dbg.rec[k][1] = site | ((unsigned long long)n << 32);
        g_dbg.rec[k][2] = blockIdx.x; g_dbg.rec[k][3] = threadIdx.x;
    }
    return false;
}
template <typename T>
__device__ __forceinline__ T dbg_ld(const T *p, uint32_t site) {
    if (dbg_ok(p, sizeof(T), site)) return *p;
    T z;
    __builtin_memset(&z, 0, sizeof(T));
    return z;
}
template <typename T, typename V>
__device__ __forceinline__ void dbg_st(T *p, V v, uint32_t site) {
    if (dbg_ok(p, sizeof(T), site)) *p = (T)v;
}
#define P4S_LD(ptr, site) ::p4s::dbg_ld((ptr), (site))
#define P4S_ST(ptr, val, site) ::p4s::dbg_st((ptr), (val), (site))
#define P4S_INB(ptr, n, site) ::p4s::dbg_ok((ptr), (n), (site))
#else
#define P4S_LD(ptr, site) (*(ptr))
#define P4S_ST(ptr, val, site) (*(ptr) = (val))
#define P4S_INB(ptr, n, site) true
#endif

// ---------------------------------------------------------------------------
// A page as the kernels see it: a byte range of the input allocation.
// ---------------------------------------------------------------------------
struct PageView {
    const uint8_t *base;   // first byte of the page
    uint64_t size;         // bytes in the page
    uint64_t room;         // bytes readable from base before the input allocation ends (>= size)
};

__device__ __forceinline__ PageView page_view(const uint8_t *d_in, uint64_t in_bytes, const p4s_page_desc &pd) {
    PageView pv;
    uint64_t off = pd.in_offset < in_bytes ? pd.in_offset : in_bytes;
    pv.base = d_in + off;
    pv.room = in_bytes - off;
    pv.size = pd.in_size < pv.room ? pd.in_size : pv.room;
    return pv;
}

__device__ __forceinline__ void write_status(p4s_page_status *st, int32_t code, uint64_t values,
                                             uint64_t aux0, uint64_t aux1, uint64_t consumed,
                                             uint64_t out_bytes) {
    st->code = code;
    st->reserved = 0;
    st->values_decoded = values;
    st->aux0 = aux0;
    st->aux1 = aux1;
    st->bytes_consumed = consumed;
    st->out_bytes = out_bytes;
}

// ---------------------------------------------------------------------------
// Sliding window over a page's bytes, staged in LDS by the whole workgroup.
// The window starts at a page offset `lo` chosen so that the global address is
// 16-byte aligned (lo may be up to 15 bytes before the position asked for, and
// negative when the page itself starts unaligned; bytes before the page belong
// to the same allocation because d_in is 16-byte aligned).  Bytes past the end
// of the input allocation read as zero.
// ---------------------------------------------------------------------------
template <int WIN_BYTES, int THREADS>
__device__ __forceinline__ int64_t window_load(uint8_t *lds, const PageView &pv, uint64_t pos, uint32_t tid = threadIdx.x) {
    const uint8_t *p = pv.base + pos;
    uint32_t mis = (uint32_t)((uintptr_t)p & 15u);
    int64_t lo = (int64_t)pos - (int64_t)mis;
    const uint8_t *g = pv.base + lo;
    int64_t room = (int64_t)pv.room - lo;      // bytes readable from g
    constexpr int CHUNKS = WIN_BYTES / 16;
    for (int c = (int)tid; c < CHUNKS; c += THREADS) {
        int64_t o = (int64_t)c * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (o + 16 <= room) {
            v = P4S_LD(reinterpret_cast<const uint4 *>(g + o), 1);
        } else if (o < room) {
            uint8_t tmp[16];
            for (int i = 0; i < 16; ++i) tmp[i] = (o + i < room) ? P4S_LD(g + o + i, 2) : 0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        *reinterpret_cast<uint4 *>(lds + o) = v;
    }
    return lo;
}

// The same window in two steps, so that the global loads can be in flight while the workgroup does other work:
// window_fetch issues the loads into registers (and returns the window's page offset), window_stage writes them
// to LDS.  WIN_BYTES / 16 pieces over THREADS threads.
template <int WIN_BYTES, int THREADS>
struct WindowRegs {
    static constexpr int PER_THREAD = (WIN_BYTES / 16 + THREADS - 1) / THREADS;
    uint4 v[PER_THREAD];
};

template <int WIN_BYTES, int THREADS>
__device__ __forceinline__ int64_t window_fetch(const PageView &pv, uint64_t pos, uint32_t tid, WindowRegs<WIN_BYTES, THREADS> &r) {
    const uint8_t *p = pv.base + pos;
    const uint32_t mis = (uint32_t)((uintptr_t)p & 15u);
    const int64_t lo = (int64_t)pos - (int64_t)mis;
    const uint8_t *g = pv.base + lo;
    const int64_t room = (int64_t)pv.room - lo;      // bytes readable from g
    constexpr int CHUNKS = WIN_BYTES / 16;
#pragma unroll
    for (int i = 0; i < WindowRegs<WIN_BYTES, THREADS>::PER_THREAD; ++i) {
        const int c = (int)tid + i * THREADS;
        const int64_t o = (int64_t)c * 16;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (c < CHUNKS) {
            if (o + 16 <= room) {
                v = P4S_LD(reinterpret_cast<const uint4 *>(g + o), 3);
            } else if (o < room) {
                // the last, partial piece of the input: byte loads, gathered without an indexed array
                uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
#pragma unroll
                for (int b = 0; b < 16; ++b) {
                    const uint32_t x = (o + b < room) ? (uint32_t)P4S_LD(g + o + b, 4) << (8 * (b & 3)) : 0u;
                    if (b < 4) w0 |= x; else if (b < 8) w1 |= x; else if (b < 12) w2 |= x; else w3 |= x;
                }
                v = make_uint4(w0, w1, w2, w3);
            }
        }
        r.v[i] = v;
    }
    return lo;
}

template <int WIN_BYTES, int THREADS>
__device__ __forceinline__ void window_stage(uint8_t *lds, uint32_t tid, const WindowRegs<WIN_BYTES, THREADS> &r) {
    constexpr int CHUNKS = WIN_BYTES / 16;
#pragma unroll
    for (int i = 0; i < WindowRegs<WIN_BYTES, THREADS>::PER_THREAD; ++i) {
        const int c = (int)tid + i * THREADS;
        if (c < CHUNKS) *reinterpret_cast<uint4 *>(lds + (int64_t)c * 16) = r.v[i];
    }
}

// ULEB128 at byte offset `pos` of the page, read with plain global loads
// (bit_stream_utils.hh:424-441 of the reference: at most max_bytes bytes).
// Returns the number of bytes consumed, 0 when the page ends first or the
// terminator is missing.
__device__ __forceinline__ uint32_t uleb_global(const PageView &pv, uint64_t pos, uint32_t max_bytes, uint64_t *out) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < max_bytes; ++i) {
        if (pos + i >= pv.size) return 0;
        uint8_t b = P4S_LD(pv.base + pos + i, 5);
        v |= (uint64_t)(b & 0x7F) << (7 * i);
        if (!(b & 0x80)) { *out = v; return i + 1; }
    }
    return 0;
}

__device__ __forceinline__ int64_t zigzag64(uint64_t u) { return (int64_t)((u >> 1) ^ (0 - (u & 1))); }

}  // namespace p4s
