// DELTA_BINARY_PACKED encoder (ref: delta_binary_packed_encoder<T>, src/encoding.cc:862-996): the page one
// put_batch(values, n) + flush() writes.  One 256-thread workgroup per page; a round is four blocks of 256 deltas,
// one block per wavefront, four consecutive deltas per lane (a miniblock of 32 values is eight lanes).
//
// A wavefront's block: delta = value - previous value in unsigned_type (:893-898); the block's smallest delta as a
// signed value (:900-904, DPP butterfly minimum over the lanes); delta - minimum; the width of a miniblock is the bit
// width of its largest value = of the OR of its values (:905-914; three DPP steps inside the eight lanes); a
// prefix sum over the miniblock sizes places every lane; the block's bytes [zigzag ULEB128 minimum | 8 width bytes |
// 4 * width bytes per used miniblock] (:922-935) are put together in the wavefront's own LDS staging area (every
// lane ORs its fields in) and leave with dword stores.  Blocks follow each other in the page: the four block sizes
// of a round are exchanged through LDS (one barrier per round) and the position is a running sum; pages are
// independent.
// The reference writes the never-initialised delta slots behind a page's last value as padding of that value's
// miniblock (:889,:932-934); here those bits are 0.
#pragma once
#include "p4s_common.cuh"
#include <type_traits>

namespace p4s {

constexpr int ENC_THREADS = 256;
constexpr int ENC_BLOCK = 256;                   // the reference's BLOCK_VALUES (:871)
constexpr int ENC_MINIBLOCKS = 8;                // :872
constexpr int ENC_VPL = 4;                       // deltas per lane: 64 lanes x 4 = one block
constexpr int ENC_STAGE_WORDS = (10 + 8 + 256 * 8) / 4 + 4;   // a block of 64-bit fields, in dwords (+ slack for a field's third dword)

struct EncShared {
    uint32_t stage[4][ENC_STAGE_WORDS];          // one staging area per wavefront
    uint32_t block_bytes[2][4];                  // sizes of a round's four blocks, by the round's parity
};

// ULEB128 of v into bytes (at most 10); returns the length.
__device__ __forceinline__ uint32_t enc_uleb(uint64_t v, uint8_t *b) {
    uint32_t n = 0;
    while (v & ~0x7Full) { b[n++] = (uint8_t)((v & 0x7F) | 0x80); v >>= 7; }
    b[n++] = (uint8_t)v;
    return n;
}

template <typename T>      // uint32_t or uint64_t: the reference's unsigned_type
__global__ __launch_bounds__(ENC_THREADS, 8) void k_delta_encode(
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
    uint64_t pos;                                    // bytes of the page so far (uniform)
    bool full;
    {
        uint8_t hdr[2 + 1 + 10 + 10];
        uint32_t hl = 0;
        hdr[hl++] = 0x80; hdr[hl++] = 0x02;          // 256
        hdr[hl++] = ENC_MINIBLOCKS;
        hl += enc_uleb(n, hdr + hl);
        const int64_t first = n ? (int64_t)(S)v[0] : 0;
        hl += enc_uleb(((uint64_t)first << 1) ^ (uint64_t)(first >> 63), hdr + hl);
        pos = hl;
        full = pos > cap;
        if (!full && tid == 0)
            for (uint32_t i = 0; i < hl; ++i) out[i] = hdr[i];
    }

    uint32_t *stage = sh.stage[wave];
    uint32_t par = 0;
    const uint32_t l0 = lane * ENC_VPL;                                 // this lane's first delta of its block
    // a lane's values of the block that starts at value b: the one before its first delta, then its ENC_VPL values
    // (loaded a round ahead: the next round's loads are in flight while this round is put together)
    auto fetch = [&](uint64_t b, T (&x)[ENC_VPL + 1]) {
        const uint32_t c = b < n ? (n - b < ENC_BLOCK ? (uint32_t)(n - b) : (uint32_t)ENC_BLOCK) : 0u;
        x[0] = l0 < c ? v[b + l0 - 1] : (T)0;
#pragma unroll
        for (int j = 0; j < ENC_VPL; ++j) x[j + 1] = l0 + j < c ? v[b + l0 + j] : (T)0;
    };
    T nxt[ENC_VPL + 1];
    fetch(1 + (uint64_t)wave * ENC_BLOCK, nxt);
    for (uint64_t r0 = 1; r0 < n; r0 += 4 * ENC_BLOCK, par ^= 1u) {
        const uint64_t b0 = r0 + (uint64_t)wave * ENC_BLOCK;          // this wavefront's block starts at value b0
        const uint32_t cnt = b0 < n ? (n - b0 < ENC_BLOCK ? (uint32_t)(n - b0) : (uint32_t)ENC_BLOCK) : 0u;
        // ---- deltas (:893-898)
        T d[ENC_VPL];
#pragma unroll
        for (int j = 0; j < ENC_VPL; ++j) d[j] = l0 + j < cnt ? (T)(nxt[j + 1] - nxt[j]) : (T)0;
        fetch(b0 + 4 * ENC_BLOCK, nxt);
        // ---- the block's smallest delta as signed_type (:900-904)
        long long m = 0x7FFFFFFFFFFFFFFFll;
#pragma unroll
        for (int j = 0; j < ENC_VPL; ++j)
            if (l0 + j < cnt) { const long long x = (long long)(S)d[j]; m = x < m ? x : m; }
        // (butterflies inside each row of 16 lanes by DPP: lane ^ 1, lane ^ 2, mirror of 8, mirror of 16; then the four rows)
#define P4S_ENC_MIN_STEP(CTRL) { const long long o = (long long)dpp_mov64<CTRL, 0xF>((uint64_t)m); m = o < m ? o : m; }
        P4S_ENC_MIN_STEP(0xB1) P4S_ENC_MIN_STEP(0x4E) P4S_ENC_MIN_STEP(0x141) P4S_ENC_MIN_STEP(0x140)
#undef P4S_ENC_MIN_STEP
        {
            long long r = (long long)bcast_u64((uint64_t)m, 0);
#pragma unroll
            for (int row = 1; row < 4; ++row) { const long long o = (long long)bcast_u64((uint64_t)m, 16 * row); r = o < r ? o : r; }
            m = r;
        }
        const long long mn = cnt ? m : 0;
        // ---- delta - minimum; a miniblock's width is the bit width of the OR of its values (:905-914)
        unsigned long long orv = 0;
#pragma unroll
        for (int j = 0; j < ENC_VPL; ++j) {
            d[j] = l0 + j < cnt ? (T)(d[j] - (T)mn) : (T)0;
            orv |= (unsigned long long)d[j];
        }
        orv |= dpp_mov64<0xB1, 0xF>(orv);            // lane ^ 1
        orv |= dpp_mov64<0x4E, 0xF>(orv);            // lane ^ 2
        orv |= dpp_mov64<0x141, 0xF>(orv);           // mirror of eight lanes: the other quad of the miniblock
        const uint32_t w = orv ? 64u - (uint32_t)__builtin_clzll(orv) : 0u;      // of this lane's miniblock (lane >> 3)
        // ---- sizes: miniblock k takes 4 * w_k bytes while it holds a value (:926-935)
        const uint32_t used = (cnt + 31u) >> 5;
        const uint32_t mine = ((lane & 7u) == 0 && (lane >> 3) < used) ? 4u * w : 0u;
        const uint32_t inc = wave_iscan(mine);
        const uint32_t before = __shfl(inc - mine, (int)(lane & ~7u));             // bytes of the miniblocks in front of this lane's
        const uint32_t total = __shfl(inc, 63);
        uint8_t mb[10];
        const uint32_t ml = enc_uleb(((uint64_t)mn << 1) ^ (uint64_t)(mn >> 63), mb);
        const uint32_t body0 = ml + ENC_MINIBLOCKS;
        const uint32_t block_bytes = cnt ? body0 + total : 0u;
        if (lane == 0) sh.block_bytes[par][wave] = block_bytes;
        // ---- the block's bytes in the wavefront's staging area
        if (cnt) {
            for (uint32_t i = lane; i < (block_bytes + 3u) / 4u + 2u; i += 64) stage[i] = 0;
            if (lane == 0) {
                for (uint32_t i = 0; i < ml; ++i) atomicOr(&stage[i >> 2], (uint32_t)mb[i] << (8u * (i & 3u)));
            }
            if ((lane & 7u) == 0 && w) {
                const uint32_t at = ml + (lane >> 3);
                atomicOr(&stage[at >> 2], w << (8u * (at & 3u)));
            }
            uint64_t bit = (uint64_t)(body0 + before) * 8u + (uint64_t)((lane & 7u) * ENC_VPL) * w;
            if (w <= 16u) {
                // narrow fields: the lane's four make one value of at most 64 bits (three dwords at most)
                const uint64_t x = (uint64_t)d[0] | ((uint64_t)d[1] << w) | ((uint64_t)d[2] << (2u * w)) | ((uint64_t)d[3] << (3u * w));
                if (x) {
                    const uint32_t dw = (uint32_t)(bit >> 5), s = (uint32_t)(bit & 31u);
                    if ((uint32_t)(x << s)) atomicOr(&stage[dw], (uint32_t)(x << s));
                    const uint64_t hi = s ? (x >> (32u - s)) : (x >> 32);
                    if ((uint32_t)hi) atomicOr(&stage[dw + 1], (uint32_t)hi);
                    if ((uint32_t)(hi >> 32)) atomicOr(&stage[dw + 2], (uint32_t)(hi >> 32));
                }
            } else
#pragma unroll
            for (int j = 0; j < ENC_VPL; ++j) {
                const uint64_t x = (uint64_t)d[j];
                if (x) {
                    const uint32_t dw = (uint32_t)(bit >> 5), s = (uint32_t)(bit & 31u);
                    atomicOr(&stage[dw], (uint32_t)(x << s));
                    const uint64_t hi = s ? (x >> (32u - s)) : (x >> 32);     // the bits above the first dword
                    if ((uint32_t)hi) atomicOr(&stage[dw + 1], (uint32_t)hi);
                    if ((uint32_t)(hi >> 32)) atomicOr(&stage[dw + 2], (uint32_t)(hi >> 32));
                }
                bit += w;
            }
        }
        __syncthreads();
        // ---- where the block goes: behind the blocks of the waves before this one
        uint32_t mypos = 0, round_bytes = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = sh.block_bytes[par][i];
            if ((uint32_t)i < wave) mypos += x;
            round_bytes += x;
        }
        if (!full && pos + round_bytes > cap) full = true;     // (a round that does not fit is not written at all)
        if (!full && cnt) {
            const uint8_t *src = reinterpret_cast<const uint8_t *>(stage);
            uint8_t *dst = out + pos + mypos;
            const uint32_t mis = (uint32_t)((uintptr_t)dst & 3u);
            const uint32_t headb = mis ? (4u - mis < block_bytes ? 4u - mis : block_bytes) : 0u;   // bytes up to a dword boundary of the output
            if (lane < headb) dst[lane] = src[lane];
            const uint32_t nd = (block_bytes - headb) >> 2;
            for (uint32_t i = lane; i < nd; i += 64) {
                uint32_t x;
                __builtin_memcpy(&x, src + headb + 4u * i, 4);                // (LDS reads need no alignment)
                reinterpret_cast<uint32_t *>(dst + headb)[i] = x;
            }
            const uint32_t tail0 = headb + 4u * nd;
            if (lane < block_bytes - tail0) dst[tail0 + lane] = src[tail0 + lane];
        }
        pos += round_bytes;
    }
    if (tid == 0) {
        if (full) write_status(&status[page], E_OUT_CAPACITY, 0, pos, 0, 0, 0);
        else write_status(&status[page], P4S_E_NONE, n, 0, 0, n * sizeof(T), pos);
    }
}

}  // namespace p4s
