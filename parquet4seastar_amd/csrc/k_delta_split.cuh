// DELTA_BINARY_PACKED pages split over several workgroups (small batches: with one workgroup per page a batch of
// few pages leaves most of the chip idle and a page takes the latency of its 512-block chain plus its decode).
//
//   k_delta_split_guess   one wave per page: blocks of one size follow each other at equal distances (constant
//                         miniblock widths are common), so every lane reads the header it expects 1, 2, .. 63 block
//                         sizes further on, straight from global memory; the leading lanes that found a block of the
//                         expected size are where the chain goes.  Every B-th block start becomes the start of a part.
//                         A page whose blocks keep changing size (or of more than 8 miniblocks a block) is left to
//   k_delta_split_walk    one workgroup per page: the block-header walk of the decode kernel without the decode
//                         (windows through LDS, wave 0 follows the chain of block headers).  Pages the fast chain
//                         cannot walk to their last whole block (odd shapes, errors, reads that stop before the
//                         stream's end) are not split.
//   k_delta_binary_packed (page, part) pairs decoded like pages of their own: the first part from the page's first
//                         value, the others with a running value of 0.
//   k_delta_split_finish  adds to every later part the value its predecessor ended on (a prefix sum over the parts'
//                         last values), writes the page's status; a page one of whose parts met an error is marked
//                         and decoded again, whole, by a last launch of the decode kernel, so that every error
//                         outcome is the one-workgroup decoder's.
#pragma once
#include "k_delta.cuh"

namespace p4s {

constexpr uint32_t DELTA_SPLIT_MIN_BLOCKS = 16;      // full blocks a page needs to be worth splitting

struct DeltaNullSink {
    static constexpr bool PREFETCH_WINDOW = true;
    static constexpr bool SHIFTED_STORES = false;
    template <typename T> __device__ __forceinline__ void first(T) const {}
};

constexpr uint32_t DELTA_SPLIT_UNDECIDED = 0xFFFFFFFFu;   // nparts: k_delta_split_guess leaves the page to k_delta_split_walk

// What both passes work out from the page header: whether the page is cut at all, and how.
struct DeltaSplitShape {
    bool ok;
    uint32_t nblocks, per_part, nparts, blk_vals;
};
__device__ __forceinline__ DeltaSplitShape delta_split_shape(bool go, const DeltaResult &res, const DeltaWalk &st, uint64_t count, uint32_t fan) {
    DeltaSplitShape g{false, 0, 0, 0, 0};
    // a full read of a stream of ordinary shape with enough whole blocks
    const uint64_t blk_vals64 = (uint64_t)st.nmb * st.vpm;
    g.ok = go && !res.err && count == res.total && st.nmb <= 32 && st.vpm <= 65536 && blk_vals64 > 0 && blk_vals64 <= 0xFFFFFFFFull;
    if (g.ok) {
        g.blk_vals = (uint32_t)blk_vals64;
        g.nblocks = (uint32_t)((count - 1) / g.blk_vals);             // whole blocks (the rest is the last part's tail)
        g.ok = g.nblocks >= DELTA_SPLIT_MIN_BLOCKS;
    }
    if (g.ok) {
        g.per_part = (g.nblocks + fan - 1) / fan;
        g.per_part = (g.per_part + 3u) & ~3u;                         // whole tiles of the decoder where blocks are 256 values
        g.nparts = (g.nblocks + g.per_part - 1) / g.per_part;
        g.ok = g.nparts >= 2;
    }
    return g;
}

template <typename T>
__global__ __launch_bounds__(64) void k_delta_split_guess(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        DeltaPartRec *__restrict__ parts, DeltaSplitPage *__restrict__ split, uint32_t fan) {
    const uint32_t page = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    DeltaNullSink sink;
    DeltaResult res;
    DeltaWalk st;
    T first;
    uint64_t count = 0, p = 0;
    DeltaPartRec *my = parts + (uint64_t)page * fan;
    const bool go = delta_begin<T>(pv, pd.num_values, false, sink, res, st, first, count, p);
    const DeltaSplitShape g = delta_split_shape(go, res, st, count, fan);
    bool undecided = false;
    if (g.ok) {
        const uint32_t nmb = st.nmb, vpm32 = (uint32_t)st.vpm;
        const uint32_t wmask_lo = nmb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nmb)) - 1u);
        const uint32_t wmask_hi = nmb <= 4 ? 0u : (nmb == 8 ? 0xFFFFFFFFu : ((1u << (8 * (nmb - 4))) - 1u));
        if (lane == 0) my[0].at = DeltaResume{p, 1, g.per_part * g.blk_vals, 0u};
        uint64_t P = p;              // page offset of block `bdone`: known, not guessed
        uint32_t bdone = 0, D = 0, poor = 0;
        undecided = nmb > 8;
        while (!undecided && bdone < g.nblocks) {
            const uint64_t pos = P + (uint64_t)lane * D;
            // 24 bytes around the header, by aligned loads (the input allocation starts 16-byte aligned)
            const bool readable = pos + 24 <= pv.room;
            const uintptr_t addr = reinterpret_cast<uintptr_t>(pv.base) + (readable ? pos : 0);
            const uint32_t *q = reinterpret_cast<const uint32_t *>(addr & ~(uintptr_t)3);
            uint32_t r[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) r[i] = readable ? P4S_LD(q + i, 120) : 0u;
            const uint32_t sh8 = (uint32_t)(addr & 3u);
            const uint32_t h0 = __builtin_amdgcn_alignbyte(r[1], r[0], sh8), h1 = __builtin_amdgcn_alignbyte(r[2], r[1], sh8),
                           h2 = __builtin_amdgcn_alignbyte(r[3], r[2], sh8), h3 = __builtin_amdgcn_alignbyte(r[4], r[3], sh8),
                           h4 = __builtin_amdgcn_alignbyte(r[5], r[4], sh8);           // bytes pos .. pos+19
            const uint64_t stops = ~(((uint64_t)h1 << 32) | h0) & 0x8080808080808080ull;
            const uint32_t stops2 = ~h2 & 0x8080u;               // bytes 8 and 9: a 64-bit minimum takes up to 10 bytes
            const uint32_t t = stops ? ((uint32_t)__builtin_ctzll(stops) >> 3)
                                     : (stops2 ? 8u + ((uint32_t)__builtin_ctz(stops2) >> 3) : 10u);   // last ULEB128 byte
            const uint32_t s1 = t + 1u, sd = s1 >> 2, sb = s1 & 3u;                    // width bytes start at byte t+1
            const uint32_t x0 = sd == 0 ? h0 : (sd == 1 ? h1 : h2);
            const uint32_t x1 = sd == 0 ? h1 : (sd == 1 ? h2 : h3);
            const uint32_t x2 = sd == 0 ? h2 : (sd == 1 ? h3 : h4);
            const uint32_t w0 = __builtin_amdgcn_alignbyte(x1, x0, sb) & wmask_lo;
            const uint32_t w1 = __builtin_amdgcn_alignbyte(x2, x1, sb) & wmask_hi;
            const uint32_t sum = __builtin_amdgcn_sad_u8(w0, 0u, __builtin_amdgcn_sad_u8(w1, 0u, 0u));
            // some width byte above 64
            const bool bad = ((((w0 & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) | w0 | ((w1 & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) | w1) & 0x80808080u) != 0;
            const uint32_t size = s1 + nmb + ((sum * vpm32 + 7) >> 3);
            const bool ok = readable && t < 10 && !bad && pos + size <= pv.size;
            const uint64_t m_ok = __ballot(ok);
            const uint64_t m_same = __ballot(ok && size == D);
            const uint32_t cap = g.nblocks - bdone;
            const uint32_t m = D == 0 ? 0u : (~m_same ? (uint32_t)__builtin_ctzll(~m_same) : 64u);      // leading repeats
            uint32_t c;
            if (m >= cap) c = cap;
            else if (m == 64u) c = 64u;
            else c = m + (uint32_t)((m_ok >> m) & 1ull);       // block m starts at the right place: take it if it is sound
            // a header this pass cannot read, or sizes that keep changing: the walk through LDS does better
            if (c == 0 || (c < 8u && c < cap && ++poor > 1u)) { undecided = true; break; }
            // starts of parts among these blocks
            if (lane < c) {
                const uint32_t b = bdone + lane;
                if (b > 0 && b % g.per_part == 0) {
                    const uint32_t s = b / g.per_part;
                    const uint64_t before = (uint64_t)b * g.blk_vals;                   // deltas before the part
                    const bool last = s == g.nparts - 1;
                    my[s].at = DeltaResume{pos, 1 + before, last ? (uint32_t)(count - 1 - before) : g.per_part * g.blk_vals, last ? 1u : 0u};
                }
            }
            const uint32_t size_l = bcast_u32(size, (int)(c - 1u));
            P += (uint64_t)(c - 1u) * D + size_l;
            if (c < 2u) D = size_l;
            bdone += c;
        }
    }
    if (lane == 0) {
        split[page].nparts = undecided ? DELTA_SPLIT_UNDECIDED : (g.ok ? g.nparts : 0u);
        split[page].redo = 0;
        split[page].count = g.ok ? res.count : 0;
    }
}

// The walk takes its pages through windows four times the decoder's (a round costs the latency of its window's
// loads whatever the window's size; nothing else competes for the LDS here).
using DeltaWalkRound = DeltaRound<49152, 3072>;

template <typename T>
__global__ __launch_bounds__(DELTA_THREADS) void k_delta_split_walk(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        DeltaPartRec *__restrict__ parts, DeltaSplitPage *__restrict__ split, uint32_t fan) {
    __shared__ DeltaWalkRound sh;
    __shared__ uint32_t s_bad;
    const uint32_t page = blockIdx.x;
    if (split[page].nparts != DELTA_SPLIT_UNDECIDED) return;        // k_delta_split_guess has dealt with the page
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = first_u32(tid >> 6);
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    DeltaNullSink sink;
    DeltaResult res;
    DeltaWalk st;
    T first;
    uint64_t count = 0, p = 0;
    DeltaPartRec *my = parts + (uint64_t)page * fan;
    const bool go = delta_begin<T>(pv, pd.num_values, false, sink, res, st, first, count, p);
    // a full read of a stream of ordinary shape with enough whole blocks
    const uint64_t blk_vals64 = (uint64_t)st.nmb * st.vpm;
    bool ok = go && !res.err && count == res.total && st.nmb <= 32 && st.vpm <= 65536 && blk_vals64 > 0 && blk_vals64 <= 0xFFFFFFFFull;
    uint32_t nblocks = 0, per_part = 0, nparts = 0, blk_vals = 0;
    if (ok) {
        blk_vals = (uint32_t)blk_vals64;
        nblocks = (uint32_t)((count - 1) / blk_vals);               // whole blocks (the rest is the last part's tail)
        ok = nblocks >= DELTA_SPLIT_MIN_BLOCKS;
    }
    if (ok) {
        per_part = (nblocks + fan - 1) / fan;
        per_part = (per_part + 3u) & ~3u;                           // whole tiles of the decoder where blocks are 256 values
        nparts = (nblocks + per_part - 1) / per_part;
        ok = nparts >= 2;
    }
    if (tid == 0) s_bad = 0;
    __syncthreads();
    if (ok) {
        if (tid == 0) my[0].at = DeltaResume{p, 1, nparts == 1 ? (uint32_t)(count - 1) : per_part * blk_vals, nparts == 1 ? 1u : 0u};
        uint32_t bdone = 0;
        int64_t wlo = window_load<DeltaWalkRound::WIN, DELTA_THREADS>(sh.win, pv, p);
        if (tid < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DeltaWalkRound::WIN)[tid] = 0;
        __syncthreads();
        for (;;) {
            if (wave == 0) {
                delta_walk_round(sh, pv, wlo, st, DELTA_THREADS * DeltaVpt<T>::value);
                st.make_uniform();
            }
            __syncthreads();
            const uint32_t nb = sh.nblk;
            if (sh.err || nb == 0) {
                // the general walker ran: fine for what follows the last whole block, a reason not to split before it
                if (bdone < nblocks && tid == 0) s_bad = 1;
                break;
            }
            // starts of parts among this round's blocks
            if (tid < nb) {
                const uint32_t g = bdone + tid;
                if (g > 0 && g % per_part == 0 && g < nblocks) {
                    const uint32_t s = g / per_part;
                    const uint64_t pos = (uint64_t)(wlo + (int64_t)(sh.blk[tid] & 0xFFFFu));
                    const uint64_t before = (uint64_t)g * blk_vals;                 // deltas before the part
                    const bool last = s == nparts - 1;
                    my[s].at = DeltaResume{pos, 1 + before, last ? (uint32_t)(count - 1 - before) : per_part * blk_vals, last ? 1u : 0u};
                }
            }
            bdone += nb;
            if (bdone >= nblocks || !sh.more) break;
            const uint64_t next = sh.next_pos;
            __syncthreads();
            wlo = window_load<DeltaWalkRound::WIN, DELTA_THREADS>(sh.win, pv, next);
            if (tid < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DeltaWalkRound::WIN)[tid] = 0;
            __syncthreads();
        }
        __syncthreads();
        if (s_bad || bdone < nblocks) ok = false;
    }
    if (tid == 0) {
        split[page].nparts = ok ? nparts : 0u;
        split[page].redo = 0;
        split[page].count = ok ? res.count : 0;
    }
}

// blockIdx.x = page * fan + s.  Part s > 0 of a split page: its values += the value part s-1 ended on.
template <typename T>
__global__ __launch_bounds__(256) void k_delta_split_finish(
        const p4s_page_desc *__restrict__ pages, T *__restrict__ d_out, p4s_page_status *__restrict__ status,
        const DeltaPartRec *__restrict__ parts, DeltaSplitPage *__restrict__ split, uint32_t fan) {
    const uint32_t page = blockIdx.x / fan, s = blockIdx.x % fan;
    const uint32_t np = split[page].nparts;
    if (np == 0 || s >= np) return;
    const DeltaPartRec *my = parts + (uint64_t)page * fan;
    bool bad = false;
    T base = 0;
    for (uint32_t q = 0; q < np; ++q) {
        bad |= my[q].err != 0;
        if (q < s) base += (T)my[q].last;
    }
    if (bad) {
        if (s == 0 && threadIdx.x == 0 && blockIdx.y == 0) split[page].redo = 1;
        return;
    }
    const p4s_page_desc pd = pages[page];
    if (s == 0 && threadIdx.x == 0 && blockIdx.y == 0) write_status(&status[page], P4S_E_NONE, split[page].count, 0, 0, my[np - 1].end_pos, 0);
    T *out = d_out + pd.out_offset;
    const uint64_t lo = my[s].at.done, n = my[s].at.deltas;
    const uint64_t limit = pd.num_values;
    // The decoder's 16-byte aligned stores write the value before a tile together with the tile (for a whole page that
    // is the same value again): the first tile of part s+1 has put its own running value, 0, over this part's last
    // value.  So that one is written from the part's record, not read back.
    const bool clobbered = s + 1 < np;
    const uint64_t n_rmw = clobbered ? n - 1 : n;
    if (clobbered && threadIdx.x == 0 && blockIdx.y == 0 && lo + n - 1 < limit) P4S_ST(out + lo + n - 1, (T)((T)my[s].last + base), 97);
    if (s == 0) return;
    // blockIdx.y = slice of the part: four independent read-modify-writes per thread and step
    const uint64_t step = (uint64_t)gridDim.y * blockDim.x * 4;
    for (uint64_t i0 = ((uint64_t)blockIdx.y * blockDim.x + threadIdx.x) * 4; i0 < n_rmw; i0 += step) {
        T v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint64_t idx = lo + i0 + j; v[j] = (i0 + j < n_rmw && idx < limit) ? P4S_LD(out + idx, 95) : (T)0; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const uint64_t idx = lo + i0 + j; if (i0 + j < n_rmw && idx < limit) P4S_ST(out + idx, (T)(v[j] + base), 96); }
    }
}

}  // namespace p4s
