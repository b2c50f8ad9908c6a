// DELTA_BINARY_PACKED encoder (ref: delta_binary_packed_encoder<T>, src/encoding.cc:862-996): the page one
// put_batch(values, n) + flush() writes.  One 256-thread workgroup per page, one block of 256 deltas per round
// (a thread per delta).
//
// A round: delta = value - previous value in unsigned_type (:893-898); the block's smallest delta as a signed value
// (:900-904, butterfly minimum per wave + four LDS words); delta - minimum; the width of a miniblock of 32 values is the
// bit width of its largest value = of the OR of its values (:905-914; LDS atomic OR, eight words); the block's
// bytes [zigzag ULEB128 minimum | 8 width bytes | 4 * width bytes per used miniblock] (:922-935) are put together in
// LDS (every thread ORs its field in) and leave with coalesced stores.  Blocks follow each other in the page, so
// the rounds of a page run in order and the position is a running sum; pages are independent.
// The reference writes the never-initialised delta slots behind a page's last value as padding of that value's
// miniblock (:889,:932-934); here those bits are 0.
#pragma once
#include "p4s_common.cuh"
#include "k_strings.cuh"      // E_OUT_CAPACITY
#include <type_traits>

namespace p4s {

constexpr int ENC_THREADS = 256;                 // = the reference's BLOCK_VALUES (:871)
constexpr int ENC_MINIBLOCKS = 8;                // :872
constexpr int ENC_STAGE_WORDS = (10 + 8 + 256 * 8) / 4 + 4;   // a block of 64-bit fields, in dwords (+ slack for a field's third dword)

struct EncShared {
    uint32_t stage[ENC_STAGE_WORDS];
    unsigned long long mb_or[ENC_MINIBLOCKS];
    long long wave_min[4];
};

// Smallest of a wavefront's values (six butterfly steps), in every lane.
__device__ __forceinline__ long long enc_wave_min(long long m) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long o = __shfl_xor(m, off);
        m = o < m ? o : m;
    }
    return m;
}

// ULEB128 of v into bytes (at most 10); returns the length.
__device__ __forceinline__ uint32_t enc_uleb(uint64_t v, uint8_t *b) {
    uint32_t n = 0;
    while (v & ~0x7Full) { b[n++] = (uint8_t)((v & 0x7F) | 0x80); v >>= 7; }
    b[n++] = (uint8_t)v;
    return n;
}

template <typename T>      // uint32_t or uint64_t: the reference's unsigned_type
__global__ __launch_bounds__(ENC_THREADS) void k_delta_encode(
        const T *__restrict__ d_values, uint64_t n_values, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, uint64_t out_bytes, p4s_page_status *__restrict__ status) {
    using S = typename std::make_signed<T>::type;
    __shared__ EncShared sh;
    const uint32_t page = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const p4s_page_desc pd = pages[page];
    // the page's values and its output slot, clamped to the buffers of the call
    uint64_t n = pd.num_values;
    if (pd.in_offset >= n_values) n = 0; else if (n > n_values - pd.in_offset) n = n_values - pd.in_offset;
    const T *v = d_values + pd.in_offset;
    uint64_t cap = pd.param0;
    if (pd.out_offset >= out_bytes) cap = 0; else if (cap > out_bytes - pd.out_offset) cap = out_bytes - pd.out_offset;
    uint8_t *out = d_out + pd.out_offset;

    // ---- flush(): block size, miniblocks per block, total, first value (:977-982)
    uint8_t hdr[2 + 1 + 10 + 10];
    uint32_t hl = 0;
    hdr[hl++] = 0x80; hdr[hl++] = 0x02;              // 256
    hdr[hl++] = ENC_MINIBLOCKS;
    hl += enc_uleb(n, hdr + hl);
    const int64_t first = n ? (int64_t)(S)v[0] : 0;
    hl += enc_uleb(((uint64_t)first << 1) ^ (uint64_t)(first >> 63), hdr + hl);
    uint64_t pos = hl;                               // bytes of the page so far (uniform)
    bool full = pos > cap;
    if (!full && tid < hl) out[tid] = hdr[tid];

    for (uint64_t b0 = 1; b0 < n; b0 += ENC_THREADS) {
        const uint64_t left = n - b0;
        const uint32_t cnt = left < ENC_THREADS ? (uint32_t)left : (uint32_t)ENC_THREADS;
        const bool have = tid < cnt;
        const T delta = have ? (T)(v[b0 + tid] - v[b0 + tid - 1]) : (T)0;
        // ---- smallest delta of the block, as signed_type
        long long m = have ? (long long)(S)delta : 0x7FFFFFFFFFFFFFFFll;
        m = enc_wave_min(m);
        for (uint32_t i = tid; i < ENC_STAGE_WORDS; i += ENC_THREADS) sh.stage[i] = 0;
        if (tid < ENC_MINIBLOCKS) sh.mb_or[tid] = 0;
        if (lane == 0) sh.wave_min[wave] = m;
        __syncthreads();
        long long mn = sh.wave_min[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) mn = sh.wave_min[i] < mn ? sh.wave_min[i] : mn;
        const T adj = have ? (T)(delta - (T)mn) : (T)0;
        // ---- widths: the OR of a miniblock's values has the bit width of its largest one
        if (adj) atomicOr(&sh.mb_or[tid >> 5], (unsigned long long)adj);
        __syncthreads();
        uint32_t w[ENC_MINIBLOCKS];
#pragma unroll
        for (int i = 0; i < ENC_MINIBLOCKS; ++i) {
            const unsigned long long o = sh.mb_or[i];
            w[i] = o ? 64u - (uint32_t)__builtin_clzll(o) : 0u;
        }
        // ---- the block: minimum, widths, then whole miniblocks while they hold a value (:922-935)
        uint8_t mb[10];
        const uint32_t ml = enc_uleb(((uint64_t)mn << 1) ^ (uint64_t)(mn >> 63), mb);
        const uint32_t used = (cnt + 31u) >> 5;
        uint32_t body0 = ml + ENC_MINIBLOCKS, before = 0, total = 0;          // bytes in front of this thread's miniblock
#pragma unroll
        for (int i = 0; i < ENC_MINIBLOCKS; ++i) {
            if ((uint32_t)i < (tid >> 5)) before += 4u * w[i];
            if ((uint32_t)i < used) total += 4u * w[i];
        }
        const uint32_t block_bytes = body0 + total;
        if (!full && pos + block_bytes > cap) full = true;
        if (!full) {
            if (tid < ml) atomicOr(&sh.stage[tid >> 2], (uint32_t)mb[tid] << (8u * (tid & 3u)));
            if (tid < ENC_MINIBLOCKS) {
                const uint32_t at = ml + tid;
                if (w[tid]) atomicOr(&sh.stage[at >> 2], w[tid] << (8u * (at & 3u)));
            }
            const uint32_t wm = w[tid >> 5];
            if (have && adj) {
                const uint64_t bit = (uint64_t)(body0 + before) * 8u + (uint64_t)(tid & 31u) * wm;
                const uint32_t d = (uint32_t)(bit >> 5), s = (uint32_t)(bit & 31u);
                const uint64_t x = (uint64_t)adj;
                atomicOr(&sh.stage[d], (uint32_t)(x << s));
                const uint64_t hi = s ? (x >> (32u - s)) : (x >> 32);         // the bits above the first dword
                if ((uint32_t)hi) atomicOr(&sh.stage[d + 1], (uint32_t)hi);
                if ((uint32_t)(hi >> 32)) atomicOr(&sh.stage[d + 2], (uint32_t)(hi >> 32));
            }
        }
        __syncthreads();
        if (!full) {
            const uint8_t *src = reinterpret_cast<const uint8_t *>(sh.stage);
            uint8_t *dst = out + pos;
            const uint32_t mis = (uint32_t)((uintptr_t)dst & 3u);
            const uint32_t headb = mis ? (4u - mis < block_bytes ? 4u - mis : block_bytes) : 0u;   // bytes up to a dword boundary of the output
            if (tid < headb) dst[tid] = src[tid];
            const uint32_t nd = (block_bytes - headb) >> 2;
            for (uint32_t i = tid; i < nd; i += ENC_THREADS) {
                uint32_t x;
                __builtin_memcpy(&x, src + headb + 4u * i, 4);                // (LDS reads need no alignment)
                reinterpret_cast<uint32_t *>(dst + headb)[i] = x;
            }
            const uint32_t tail0 = headb + 4u * nd;
            if (tid < block_bytes - tail0) dst[tail0 + tid] = src[tail0 + tid];
        }
        pos += block_bytes;
        __syncthreads();             // the staging words are cleared again at the top
    }
    if (tid == 0) {
        if (full) write_status(&status[page], E_OUT_CAPACITY, 0, pos, 0, 0, 0);
        else write_status(&status[page], P4S_E_NONE, n, 0, 0, n * sizeof(T), pos);
    }
}

}  // namespace p4s
