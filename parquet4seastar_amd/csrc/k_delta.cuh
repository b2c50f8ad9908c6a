// DELTA_BINARY_PACKED decode kernel (ref: delta_binary_packed_decoder<T>,
// src/encoding.cc:141-250).  One 256-thread workgroup per page.
//
// A page is a strictly serial structure twice over: block b+1 starts where the
// miniblock widths of block b say it does, and value i is value i-1 plus a delta.
// The kernel therefore streams the page through an LDS window in rounds:
//   1. all four waves copy the next window of encoded bytes global -> LDS
//      (16 B per lane, coalesced, 16-byte aligned);
//   2. wave 0 walks the block headers inside the window with its 64 lanes spread
//      over the miniblocks of a block (width bytes -> DPP exclusive scan -> bit
//      position of every miniblock) and writes a segment table to LDS;
//   3. all waves unpack the round's deltas straight out of LDS, add min_delta,
//      and run the prefix sum: in-lane over the lane's consecutive values, DPP
//      scan across the wave, LDS exchange across the four waves, running carry
//      across tiles.  Each lane stores 16 contiguous bytes.
// Arithmetic is unsigned wrap-around in the width of the output type: the low
// 32 bits of the reference's uint64 running sum only depend on the low 32 bits
// of every addend, so INT32 pages run entirely in 32-bit.
#pragma once
#include "p4s_common.cuh"
#include <type_traits>

namespace p4s {

constexpr int DELTA_THREADS = 256;
constexpr int DELTA_WIN = 12288;       // encoded bytes staged per round (8 workgroups per CU fit in LDS)
constexpr int DELTA_SLACK = 32;        // field64 may touch up to 11 bytes past a field
constexpr int DELTA_MAXSEG = 384;      // miniblock segments per round
#ifndef P4S_DELTA_SLOTS
#define P4S_DELTA_SLOTS 4      // full tiles per chunk of the regular decode loop (1, 2, 4 or 8)
#endif
#ifndef P4S_DELTA_VPT64
#define P4S_DELTA_VPT64 2      // measured: 16 contiguous bytes per lane per store beat 32 (0.715 vs 0.777 ms)
#endif
// consecutive values per lane per tile: 4 x 32-bit or P4S_DELTA_VPT64 x 64-bit (16 B per lane)
template <typename T> struct DeltaVpt { static constexpr int value = sizeof(T) == 8 ? P4S_DELTA_VPT64 : 4; };
constexpr uint64_t DELTA_VPM_CAP = 1ull << 40;

// The staged bytes and the segment table of one round.
template <int WIN_, int MAXSEG_>
struct DeltaRound {
    static constexpr int WIN = WIN_;
    static constexpr int MAXSEG = MAXSEG_;
    alignas(16) uint8_t win[WIN + DELTA_SLACK];
    uint64_t seg_lo[MAXSEG];           // (bit offset in window) | width << 24 | (low half of the block's min_delta) << 32
    uint32_t seg_hi[MAXSEG];           // high half of min_delta
    uint32_t blk[MAXSEG];              // fast chain: (window offset of a block header) | (ULEB128 length - 1) << 16
    uint32_t nblk;                     // blocks found by the fast chain this round (0: general path ran)
    uint32_t wide;                     // some segment of the round is wider than 32 bits
    uint32_t big;                      // some segment of the round has width > 20 or |min_delta| >= 2^20: 64-bit sums needed
    uint32_t round_vals;               // deltas decoded this round
    uint32_t c0;                       // values in segment 0 (later segments hold vpm each)
    int32_t err;
    uint32_t err_aux;
    uint32_t more;                     // another round follows
    uint64_t next_pos;                 // page byte offset the next window must cover
    uint64_t end_pos;                  // page byte offset after the last bit consumed so far
#ifdef P4S_EXP_TIMING
    uint32_t dbg_iters;                // iterations of the fast chain, summed over the page
    uint32_t dbg_lds, dbg_rest;        // cycles of the fast chain spent on the header reads / on everything else
#endif
};

// Shared state of the barrier-per-tile decoder (delta_decode_stream).
struct DeltaShared : DeltaRound<DELTA_WIN, DELTA_MAXSEG> {
    uint64_t wtot[2][16];              // per-wave tile totals of an iteration, double buffered
};

// Walker state, uniform across wave 0 (kept in its registers between rounds).
struct DeltaWalk {
    uint64_t pos_bit;       // page bit position of the next unread delta
    uint64_t widths_pos;    // page byte offset of the current block's width bytes
    uint64_t min_delta;
    uint64_t vpm;           // values per miniblock (capped)
    uint32_t nmb;           // miniblocks per block
    uint32_t mb_idx;        // next fresh miniblock of the current block (== nmb: need a block header)
    uint64_t mb_left;       // values left in the partially consumed miniblock (0: none)
    uint32_t cur_w;         // its width
    uint32_t deltas_left;   // deltas still to decode
    bool full;              // caller asked for every value: the last miniblock must be complete
    // Quotients the fast chain needs every round, computed once (integer division is ~35 instructions):
    uint32_t blk_vals;      // values per block (0: not computed yet)
    uint32_t maxblk;        // blocks whose segments fit the table
    uint32_t q_tile;        // blocks per decoder tile when that is a power of two, else 0
    uint32_t cache_left;    // deltas_left for which cache_blocks holds deltas_left / blk_vals
    uint32_t cache_blocks;
    uint32_t last_d;        // block size the fast chain last spaced its guesses by (0: none yet); only ever a guess
    // every field is the same in all lanes: say so to the compiler (scalar registers)
    __device__ __forceinline__ void make_uniform() {
        blk_vals = first_u32(blk_vals); maxblk = first_u32(maxblk); q_tile = first_u32(q_tile);
        cache_left = first_u32(cache_left); cache_blocks = first_u32(cache_blocks); last_d = first_u32(last_d);
        pos_bit = first_u64(pos_bit); widths_pos = first_u64(widths_pos); min_delta = first_u64(min_delta);
        vpm = first_u64(vpm); nmb = first_u32(nmb); mb_idx = first_u32(mb_idx); mb_left = first_u64(mb_left);
        cur_w = first_u32(cur_w); deltas_left = first_u32(deltas_left);
    }
};

// A segment whose per-value addend (delta + min_delta) may not fit 22 signed bits: sums over a 512-value tile
// then need more than 32 bits.
__device__ __forceinline__ bool delta_is_big(uint32_t w, uint64_t min_delta) {
    const int64_t m = (int64_t)min_delta;
    return w > 20 || m >= (1 << 20) || m < -(1 << 20);
}

// Segment table accessors.
template <typename R>
__device__ __forceinline__ void delta_seg_set(R &sh, uint32_t s, uint32_t bw, uint64_t min_delta) {
    sh.seg_lo[s] = (uint64_t)bw | (min_delta << 32);
    sh.seg_hi[s] = (uint32_t)(min_delta >> 32);
}
template <typename R>
__device__ __forceinline__ uint32_t delta_seg_bw(const R &sh, uint32_t s) { return (uint32_t)sh.seg_lo[s]; }
template <typename S, typename R>
__device__ __forceinline__ S delta_seg_min(const R &sh, uint32_t s) {
    if constexpr (sizeof(S) == 4) return (S)(sh.seg_lo[s] >> 32);
    else return (S)(((uint64_t)sh.seg_hi[s] << 32) | (sh.seg_lo[s] >> 32));
}

// Byte `p` of the page: from the LDS window when it is staged there, else from global.
template <typename R>
__device__ __forceinline__ uint32_t delta_rd8(const R &sh, const PageView &pv, int64_t wlo, uint64_t p) {
    int64_t rel = (int64_t)p - wlo;
    if (rel >= 0 && rel < R::WIN) return sh.win[rel];
    return p < pv.size ? (uint32_t)P4S_LD(pv.base + p, 6) : 0u;
}

// One round of the walk (wave 0 only, all 64 lanes, uniform control flow).
// Appends segments until the window, the table or the page is exhausted.
// The general walk: partially needed blocks, the tail of a page, miniblocks wider than the window, more than
// 32 miniblocks per block, every error.  Kept out of line: it needs many registers (64-bit scans over the
// miniblocks) and inlining it made the register allocator spill values of the hot loops around it.
template <typename R>
__device__ __forceinline__ void delta_walk_general(R &sh, const PageView &pv, int64_t wlo, DeltaWalk &st) {
    constexpr int DELTA_WIN = R::WIN;          // shadow the defaults: the round's own sizes
    constexpr int DELTA_MAXSEG = R::MAXSEG;
    const uint32_t lane = lane_id();
    const uint64_t in_bits = pv.size * 8;
    const uint64_t whi_bits = (uint64_t)(wlo + DELTA_WIN) * 8;   // wlo + WIN > 0 always
    uint32_t nseg = 0;
    uint32_t round_vals = 0;
    uint32_t c0 = 0;
    bool stop = false;
    int32_t err = 0;
    uint32_t err_aux = 0;

    while (!stop && st.deltas_left > 0) {
        if (st.mb_left == 0 && st.mb_idx == st.nmb) {
            // ---- block header: zigzag ULEB128 min_delta, then nmb width bytes (ref :157-170)
            const uint64_t P = (st.pos_bit + 7) >> 3;
            if ((int64_t)P + 16 > wlo + DELTA_WIN && nseg > 0) { stop = true; break; }  // let the next window cover it
            uint32_t b = 0;
            bool valid = false;
            if (lane < 10) {
                valid = P + lane < pv.size;
                b = valid ? delta_rd8(sh, pv, wlo, P + lane) : 0x80u;
            }
            const uint64_t term = __ballot(lane < 10 && valid && !(b & 0x80u));
            const uint64_t inval = __ballot(lane < 10 && !valid);
            const uint32_t t = term ? (uint32_t)__builtin_ctzll(term) : 64u;
            const uint32_t iv = inval ? (uint32_t)__builtin_ctzll(inval) : 64u;
            if (t == 64u || iv < t) { err = P4S_E_DELTA_BLOCK_HEADER_EOF; break; }
            uint64_t u = 0;
            for (uint32_t i = 0; i <= t; ++i) u |= (uint64_t)(bcast_u32(b, (int)i) & 0x7Fu) << (7 * i);
            st.min_delta = (uint64_t)zigzag64(u);
            st.widths_pos = P + t + 1;
            if (st.widths_pos + st.nmb > pv.size) { err = P4S_E_DELTA_BLOCK_HEADER_EOF; break; }
            st.pos_bit = (st.widths_pos + st.nmb) * 8;
            st.mb_idx = 0;
        }
        // ---- up to 64 miniblocks at once: lane 0 may be the partially consumed one
        const bool has_partial = st.mb_left > 0;
        const uint32_t fresh = has_partial ? (lane == 0 ? 0xFFFFFFFFu : lane - 1) : lane;   // fresh index offset
        const bool is_partial = has_partial && lane == 0;
        const bool active = is_partial || (st.mb_idx + (uint64_t)fresh < st.nmb);
        uint32_t w = 0;
        if (is_partial) w = st.cur_w;
        else if (active) w = delta_rd8(sh, pv, wlo, st.widths_pos + st.mb_idx + fresh);
        const uint64_t avail = active ? (is_partial ? st.mb_left : st.vpm) : 0;
        const uint64_t inc_avail = wave_iscan(avail);
        const uint64_t ex = inc_avail - avail;
        const bool need = active && ex < (uint64_t)st.deltas_left;
        const uint64_t left_here = need ? (uint64_t)st.deltas_left - ex : 0;
        const uint32_t cnt = need ? (uint32_t)(avail < left_here ? avail : left_here) : 0;
        const bool last_needed = need && left_here <= avail;
        const uint64_t bits = need ? avail * (uint64_t)(w > 64 ? 0 : w) : 0;
        const uint64_t inc_bits = wave_iscan(bits);
        const uint64_t start = st.pos_bit + (inc_bits - bits);
        // ---- errors, in stream order: a bad width is seen when the miniblock is opened (before any
        // read from it), a short body when its values are read (ref :224-233, :241-249)
        const uint64_t req = (last_needed && st.full) ? avail : cnt;
        const bool bad_w = need && w > 64;
        const bool eof = need && !bad_w && w > 0 && (start > in_bits || (in_bits - start) / w < req);
        const uint64_t errmask = __ballot(bad_w || eof);
        // ---- what fits in this round
        // (needed lanes form a prefix of the wave: lane index == rank among them)
        const bool fits = need && (w == 0 || start + (uint64_t)cnt * w <= whi_bits) && (nseg + lane < DELTA_MAXSEG);
        const uint64_t needmask = __ballot(need);
        const uint64_t fitmask = __ballot(fits);
        const uint32_t nneed = (uint32_t)__popcll(needmask);
        uint32_t F = (~fitmask) ? (uint32_t)__builtin_ctzll(~fitmask) : 64u;   // leading lanes that fit
        if (F > nneed) F = nneed;
        if (errmask) {
            const uint32_t el = (uint32_t)__builtin_ctzll(errmask);
            if (el <= F) {   // the faulty miniblock is reached before this round would stop
                const uint32_t ew = bcast_u32(w, (int)el);
                if (ew > 64) { err = P4S_E_DELTA_BAD_WIDTH; err_aux = ew; } else { err = P4S_E_DELTA_DATA_EOF; }
                break;
            }
        }
        // ---- emit whole segments for lanes < F
        if (lane < F) {
            const uint32_t s = nseg + lane;
            // zero-width segments read nothing and may lie outside the window: give them offset 0
            if (w > 32) sh.wide = 1;
            if (delta_is_big(w, st.min_delta)) sh.big = 1;
            delta_seg_set(sh, s, (w ? (uint32_t)(start - (uint64_t)wlo * 8) : 0u) | (w << 24), st.min_delta);
        }
        if (nseg == 0 && F > 0) c0 = bcast_u32(cnt, 0);
        uint32_t taken = 0;       // deltas covered by the whole segments
        if (F > 0) {
            const uint64_t e = bcast_u64(ex, (int)(F - 1));
            taken = (uint32_t)e + bcast_u32(cnt, (int)(F - 1));
        }
        nseg += F;
        round_vals += taken;
        if (F < nneed) {
            // lane F is needed but does not fit entirely: take the values that do, then end the round
            const uint32_t wF = bcast_u32(w, (int)F);
            const uint32_t cntF = bcast_u32(cnt, (int)F);
            const uint64_t startF = bcast_u64(start, (int)F);
            const uint64_t availF = bcast_u64(avail, (int)F);
            const uint32_t freshF = bcast_u32(fresh, (int)F);
            uint32_t pf = 0;
            if (nseg < DELTA_MAXSEG && startF < whi_bits) {
                const uint64_t room = wF ? (whi_bits - startF) / wF : cntF;
                pf = (uint32_t)(room < cntF ? room : cntF);
            }
            if (pf > 0) {
                if (lane == 0) {
                    if (wF > 32) sh.wide = 1;
                    if (delta_is_big(wF, st.min_delta)) sh.big = 1;
                    delta_seg_set(sh, nseg, (wF ? (uint32_t)(startF - (uint64_t)wlo * 8) : 0u) | (wF << 24), st.min_delta);
                }
                if (nseg == 0) c0 = pf;
                ++nseg;
                round_vals += pf;
            }
            st.deltas_left -= taken + pf;
            st.pos_bit = startF + (uint64_t)pf * wF;
            st.cur_w = wF;
            st.mb_left = availF - pf;
            if (!(has_partial && F == 0)) st.mb_idx += freshF + 1;
            stop = true;
        } else {
            // every needed miniblock of this chunk was taken whole
            const uint32_t L = nneed - 1;
            const uint32_t wL = bcast_u32(w, (int)L);
            const uint32_t cntL = bcast_u32(cnt, (int)L);
            const uint64_t startL = bcast_u64(start, (int)L);
            const uint64_t availL = bcast_u64(avail, (int)L);
            const uint32_t freshL = bcast_u32(fresh, (int)L);
            st.deltas_left -= taken;
            if (!(has_partial && L == 0)) st.mb_idx += freshL + 1;
            if (st.deltas_left == 0) {
                // the reference consumes the rest of the last miniblock when every value was read (:241-249)
                st.pos_bit = startL + (st.full ? availL : (uint64_t)cntL) * wL;
                st.mb_left = 0;
            } else {
                st.pos_bit = startL + availL * wL;
                st.mb_left = 0;
            }
        }
    }
    if (lane == 0) {
        sh.round_vals = round_vals;
        sh.c0 = c0;
        sh.err = err;
        sh.err_aux = err_aux;
        sh.more = (err == 0 && st.deltas_left > 0) ? 1u : 0u;
        sh.next_pos = st.pos_bit >> 3;
        sh.end_pos = (st.pos_bit + 7) >> 3;
    }
}

// Turns the blocks located by the fast chain into miniblock segments, one (block, miniblock)
// pair per thread: min_delta from the block's ULEB128, the miniblock's width byte, and its bit
// position = first body bit + vpm * (sum of the widths before it).
// One round of the walk (wave 0 only, all 64 lanes, uniform control flow).
// Appends segments until the window, the table or the page is exhausted.  tile_vals: values the decoder takes
// per full tile.
template <typename R>
__device__ __forceinline__ void delta_walk_round(R &sh, const PageView &pv, int64_t wlo, DeltaWalk &st, uint32_t tile_vals) {
    constexpr int DELTA_WIN = R::WIN;          // shadow the defaults: the round's own sizes
    constexpr int DELTA_MAXSEG = R::MAXSEG;
    const uint32_t lane = opaque(lane_id());
#ifdef P4S_EXP_TIMING_WALK2
    const uint64_t tw_entry = __builtin_readcyclecounter();
    uint64_t tw_loop0 = tw_entry, tw_loop1 = tw_entry;
#endif
    // ---- fast chain: whole, fully needed blocks whose header and bodies lie inside the window.
    // The serial part of a page is only "where does the next block start"; it is kept to a handful
    // of instructions per block with all state in scalar registers: one LDS byte per lane covers the
    // header (<= 10 bytes of ULEB128) and up to 32 width bytes, a ballot finds the end of the
    // ULEB128, a DPP sum of the width bytes gives the body size.  Only the block's header position
    // is recorded; delta_expand_blocks() turns the blocks into miniblock segments afterwards, in
    // parallel over the whole workgroup.  Anything unusual is left to the general path below.
    sh.nblk = 0;
    sh.wide = 0;
    sh.big = 0;
    if (st.mb_left == 0 && st.mb_idx == st.nmb && st.nmb <= 32 && st.vpm <= 65536) {
        const uint32_t nmb = first_u32(st.nmb);
        const uint32_t vpm32 = first_u32((uint32_t)st.vpm);
        if (st.blk_vals == 0) {
            st.blk_vals = nmb * vpm32;
            st.maxblk = (uint32_t)DELTA_MAXSEG / nmb;
            const uint32_t q = tile_vals / st.blk_vals;
            st.q_tile = (q >= 1 && q * st.blk_vals == tile_vals && (q & (q - 1)) == 0) ? q : 0u;
            st.cache_left = ~0u;
        }
        const uint32_t blk_vals = st.blk_vals;
        const uint32_t maxblk = st.maxblk;
        const int64_t rel0 = (int64_t)((st.pos_bit + 7) >> 3) - wlo;
        int64_t lim64 = (int64_t)pv.size - wlo;                 // window bytes that belong to the page
        if (lim64 > DELTA_WIN) lim64 = DELTA_WIN;
        if (rel0 >= 0 && rel0 < lim64) {
            const uint32_t lim = first_u32((uint32_t)lim64);
            uint32_t P = first_u32((uint32_t)rel0);             // window-relative offset of the block header
            uint32_t left = first_u32(st.deltas_left);
            uint32_t nblk = 0;
            if (st.cache_left != left) { st.cache_left = left; st.cache_blocks = left / blk_vals; }
            uint32_t blocks_left = st.cache_blocks;            // whole blocks still wanted
            if (blocks_left > maxblk) blocks_left = maxblk;
            // Many blocks per iteration when consecutive blocks have the same size (constant widths are common).
            // D is the size of the last block taken; the blocks after it are guessed to start D bytes apart.  A
            // guessed position is right when every block before it measured the size D again, so the leading
            // blocks that did, plus the first that did not, are taken: the chain always advances by at least
            // one block.  D survives a single block of another size (a minimum delta whose ULEB128 is a byte
            // shorter or longer than its neighbours') and the end of the round: it changes only when the very
            // first block of an iteration contradicts it.
            if (nmb <= 8) {
                // One lane per block: the lane reads 24 bytes around its header with aligned loads, finds the
                // end of the ULEB128 with bit tricks and sums the width bytes with byte-SAD.
                const uint32_t wmask_lo = nmb >= 4 ? 0xFFFFFFFFu : ((1u << (8 * nmb)) - 1u);
                const uint32_t wmask_hi = nmb <= 4 ? 0u : (nmb == 8 ? 0xFFFFFFFFu : ((1u << (8 * (nmb - 4))) - 1u));
                uint32_t D = first_u32(st.last_d);
#ifdef P4S_EXP_TIMING_WALK2
                tw_loop0 = __builtin_readcyclecounter();
#endif
                while (nblk < blocks_left && P + 24 <= lim) {
                    const uint32_t pos = P + lane * D;
                    const bool inwin = pos + 24 <= lim;
                    const uint32_t *q = reinterpret_cast<const uint32_t *>(sh.win + (inwin ? (pos & ~3u) : 0u));
#if defined(P4S_EXP_TIMING) && !defined(P4S_EXP_TIMING_WALK2)
                    const uint64_t tk0 = __builtin_readcyclecounter();
#endif
                    const uint32_t r0 = q[0], r1 = q[1], r2 = q[2], r3 = q[3], r4 = q[4], r5 = q[5];
#if defined(P4S_EXP_TIMING) && !defined(P4S_EXP_TIMING_WALK2)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const uint64_t tk1 = __builtin_readcyclecounter();
#endif
                    const uint32_t sh8 = pos & 3u;
                    const uint32_t h0 = __builtin_amdgcn_alignbyte(r1, r0, sh8), h1 = __builtin_amdgcn_alignbyte(r2, r1, sh8),
                                   h2 = __builtin_amdgcn_alignbyte(r3, r2, sh8), h3 = __builtin_amdgcn_alignbyte(r4, r3, sh8),
                                   h4 = __builtin_amdgcn_alignbyte(r5, r4, sh8);           // bytes pos .. pos+19
                    const uint64_t stops = ~(((uint64_t)h1 << 32) | h0) & 0x8080808080808080ull;
                    const uint32_t stops2 = ~h2 & 0x8080u;           // bytes 8 and 9: a 64-bit minimum takes up to 10 bytes
                    // last ULEB128 byte (10: too long)
                    const uint32_t t = stops ? ((uint32_t)__builtin_ctzll(stops) >> 3)
                                             : (stops2 ? 8u + ((uint32_t)__builtin_ctz(stops2) >> 3) : 10u);
                    const uint32_t s1 = t + 1u, sd = s1 >> 2, sb = s1 & 3u;                 // width bytes start at byte t+1
                    const uint32_t x0 = sd == 0 ? h0 : (sd == 1 ? h1 : h2);
                    const uint32_t x1 = sd == 0 ? h1 : (sd == 1 ? h2 : h3);
                    const uint32_t x2 = sd == 0 ? h2 : (sd == 1 ? h3 : h4);
                    const uint32_t w0 = __builtin_amdgcn_alignbyte(x1, x0, sb) & wmask_lo;
                    const uint32_t w1 = __builtin_amdgcn_alignbyte(x2, x1, sb) & wmask_hi;
                    const uint32_t sum = __builtin_amdgcn_sad_u8(w0, 0u, __builtin_amdgcn_sad_u8(w1, 0u, 0u));
                    // some width byte above 64
                    const bool bad = ((((w0 & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) | w0 | ((w1 & 0x7F7F7F7Fu) + 0x3F3F3F3Fu) | w1) & 0x80808080u) != 0;
                    const uint32_t size = s1 + nmb + ((sum * vpm32 + 7) >> 3);
                    const bool ok = inwin && t < 10 && !bad && pos + size <= lim;
                    const uint64_t m_ok = __ballot(ok);
                    const uint64_t m_same = __ballot(ok && size == D);
                    const uint32_t cap = blocks_left - nblk;
                    const uint32_t m = D == 0 ? 0u : (~m_same ? (uint32_t)__builtin_ctzll(~m_same) : 64u);   // leading repeats
                    uint32_t c;
                    if (m >= cap) c = cap;
                    else if (m == 64u) c = 64u;
                    else c = m + (uint32_t)((m_ok >> m) & 1ull);       // block m starts at the right place: take it if it is sound
                    if (c == 0) break;
                    if (lane < c) sh.blk[nblk + lane] = pos | (t << 16);
                    const uint32_t size_l = bcast_u32(size, (int)(c - 1u));
                    P += (c - 1u) * D + size_l;
                    if (c < 2u) D = size_l;
                    nblk += c;
#if defined(P4S_EXP_TIMING) && !defined(P4S_EXP_TIMING_WALK2)
                    if (lane == 0) { sh.dbg_iters += 1; sh.dbg_lds += (uint32_t)(tk1 - tk0); sh.dbg_rest += (uint32_t)(__builtin_readcyclecounter() - tk1); }
#endif
                }
#ifdef P4S_EXP_TIMING_WALK2
                tw_loop1 = __builtin_readcyclecounter();
#endif
                left -= nblk * blk_vals;
                st.last_d = D;
            } else {
            // 8 < nmb <= 32: the wave splits into G groups of L lanes, group g reads the header at P + g * D.
            const uint32_t lshift = nmb <= 8 ? 4u : (nmb <= 22 ? 5u : 6u);
            const uint32_t L = 1u << lshift, G = 64u >> lshift;
            const uint32_t g = lane >> lshift, j = lane & (L - 1u);
            const uint64_t lmask = L == 64u ? ~0ull : ((1ull << L) - 1ull);
            const bool is_last = j == L - 1u;
            uint32_t D = first_u32(st.last_d);
            while (nblk < maxblk && left >= blk_vals && P + L <= lim) {
                const uint32_t pos = P + g * D;
                const bool inwin = pos + L <= lim;
                const uint32_t b = inwin ? sh.win[pos + j] : 0u;
                const uint64_t cont = __ballot((b & 0x80u) != 0);
                const uint32_t t = (uint32_t)__builtin_ctz(~(uint32_t)(cont >> (g << lshift)) | 0x80000000u);   // last ULEB128 byte
                const bool is_w = (j - t - 1) < nmb;                                  // lanes holding a width byte
                const uint64_t badw = __ballot(is_w && b > 64);
                const bool grp_bad = ((badw >> (g << lshift)) & lmask) != 0;
                uint32_t sum = is_w ? b : 0u;
                sum += dpp_mov<0x111, 0xF>(sum);
                sum += dpp_mov<0x112, 0xF>(sum);
                sum += dpp_mov<0x114, 0xF>(sum);
                sum += dpp_mov<0x118, 0xF>(sum);
                if (L >= 32u) sum += dpp_mov<0x142, 0xA>(sum);
                if (L == 64u) sum += dpp_mov<0x143, 0xC>(sum);
                const uint32_t size = t + 1 + nmb + ((sum * vpm32 + 7) >> 3);       // on the group's last lane
                const bool ok_any = inwin && t < 10 && t + 1 + nmb <= L && !grp_bad && pos + size <= lim;
                const uint64_t m_any = __ballot(is_last && ok_any);
                const uint64_t m_same = __ballot(is_last && ok_any && size == D);
                uint32_t cap = maxblk - nblk;
                const uint32_t lcap = left / blk_vals;
                if (lcap < cap) cap = lcap;
                const uint32_t usable = D == 0 ? 1u : G;
                uint32_t c = 0;
                bool fin = false;
#pragma unroll
                for (uint32_t gi = 0; gi < 4; ++gi) {
                    if (!fin && gi < usable && c < cap) {
                        const uint32_t bit = ((gi + 1u) << lshift) - 1u;
                        if ((m_same >> bit) & 1ull) ++c;
                        else { if ((m_any >> bit) & 1ull) ++c; fin = true; }
                    }
                }
                if (c == 0) break;
                if (j == 0 && g < c) sh.blk[nblk + g] = pos | (t << 16);
                const uint32_t size_l = bcast_u32(size, (int)((c << lshift) - 1u));
                P += (c - 1u) * D + size_l;
                if (c < 2u) D = size_l;
                nblk += c;
                left -= c * blk_vals;
#ifdef P4S_EXP_TIMING
                if (lane == 0) sh.dbg_iters += 1;
#endif
            }
            st.last_d = D;
            }
            // When more blocks follow, end the round on a whole number of tiles: the ragged rest of a round
            // costs a pass of the slow tile loop, the blocks given back open the next round instead.
            if (nblk > 0 && left >= blk_vals && st.q_tile > 1) {
                const uint32_t keep = nblk & ~(st.q_tile - 1u);
                if (keep > 0 && keep < nblk) {
                    left += (nblk - keep) * blk_vals;
                    P = sh.blk[keep] & 0xFFFFu;
                    nblk = keep;
                }
            }
            if (nblk > 0) {
                st.cache_blocks -= nblk;
                st.cache_left = left;
                st.deltas_left = left;
                st.pos_bit = ((uint64_t)wlo + P) * 8;
                if (lane == 0) {
                    sh.nblk = nblk;
                    sh.round_vals = nblk * blk_vals;
                    sh.c0 = vpm32;
                    sh.err = 0;
                    sh.err_aux = 0;
                    sh.more = left > 0 ? 1u : 0u;
                    sh.next_pos = st.pos_bit >> 3;
                    sh.end_pos = st.pos_bit >> 3;
                }
#ifdef P4S_EXP_TIMING_WALK2
                if (lane == 0) { sh.dbg_iters += 1; sh.dbg_lds += (uint32_t)(tw_loop0 - tw_entry); sh.dbg_rest += (uint32_t)(__builtin_readcyclecounter() - tw_loop1); }
#endif
                return;
            }
        }
    }

#ifdef P4S_EXP_NOGENERAL       // timing experiment: what the kernel costs without the general walk's registers
    if (lane == 0) { sh.round_vals = 0; sh.c0 = 0; sh.err = P4S_E_DELTA_DATA_EOF; sh.err_aux = 0; sh.more = 0; sh.next_pos = 0; sh.end_pos = 0; }
#else
    delta_walk_general(sh, pv, wlo, st);
#endif
}

// Run by `nthreads` threads numbered `t0` (the whole workgroup, or one wave).
template <typename R>
__device__ __forceinline__ void delta_expand_blocks(R &sh, uint32_t nmb, uint32_t vpm32, uint32_t t0, uint32_t nthreads) {
    const uint32_t nseg = sh.nblk * nmb;
    const bool nmb_pow2 = (nmb & (nmb - 1u)) == 0;
    const uint32_t nmb_shift = 31u - (uint32_t)__builtin_clz(nmb | 1u);
    for (uint32_t s = t0; s < nseg; s += nthreads) {
        const uint32_t b = nmb_pow2 ? (s >> nmb_shift) : (s / nmb), m = s - b * nmb;
        const uint32_t info = sh.blk[b];
        const uint32_t rel = info & 0xFFFFu, t = info >> 16;
        uint64_t u = 0;
        for (uint32_t i = 0; i <= t; ++i) u |= (uint64_t)(sh.win[rel + i] & 0x7Fu) << (7 * i);
        const uint8_t *wb = sh.win + rel + t + 1;
        uint32_t before = 0;
        for (uint32_t i = 0; i < m; ++i) before += wb[i];
        const uint32_t w = wb[m];
        const uint32_t bit = (rel + t + 1 + nmb) * 8 + before * vpm32;
        const uint64_t mn = (uint64_t)zigzag64(u);
        if (w > 32) sh.wide = 1;
        if (delta_is_big(w, mn)) sh.big = 1;
        delta_seg_set(sh, s, (w ? bit : 0u) | (w << 24), mn);
    }
}

// How a round's values map to its segments: segment 0 holds c0 values, every later one vpm.
struct DeltaGeom {
    uint32_t vpm32;
    bool vpm_pow2;
    uint32_t vpm_shift;
    __device__ __forceinline__ void locate(uint32_t v0, uint32_t c0, uint32_t &s, uint32_t &k) const {
        s = 0; k = v0;
        if (v0 >= c0) {
            const uint32_t r = v0 - c0;
            const uint32_t q = vpm_pow2 ? (r >> vpm_shift) : (r / vpm32);
            s = 1 + q;
            k = r - q * vpm32;
        }
    }
};

// Values v0 .. v0+VPT-1 of the round (those below nv): unpack, add min_delta, in-lane running sum in d[].
// S is the type the sums are kept in (uint32 when every addend of the round fits 22 signed bits, else T).
template <typename S, int VPT, typename R>
__device__ __forceinline__ void delta_load_tile(const R &sh, const DeltaGeom &g, uint32_t v0, uint32_t nv, uint32_t c0,
                                                bool wide, S (&d)[VPT]) {
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win);
    uint32_t s, k;
    g.locate(v0, c0, s, k);
    uint32_t seg_cnt = v0 >= c0 ? g.vpm32 : c0;
    uint32_t bw = 0, w = 0, bit = 0;
    S mn = 0;
    if (v0 < nv) {
        bw = delta_seg_bw(sh, s);
        w = bw >> 24;
        bit = (bw & 0xFFFFFFu) + k * w;
        mn = delta_seg_min<S>(sh, s);
    }
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        S x = 0;
        if (v0 + j < nv) {
            if (k == seg_cnt) {          // crossed into the next segment
                ++s;
                k = 0;
                seg_cnt = g.vpm32;
                bw = delta_seg_bw(sh, s);
                w = bw >> 24;
                bit = bw & 0xFFFFFFu;
                mn = delta_seg_min<S>(sh, s);
            }
            if constexpr (sizeof(S) == 8) x = (wide ? field64(win32, bit, w) : (uint64_t)field32(win32, bit, w)) + mn;
            else x = field32(win32, bit, w) + mn;
            bit += w;
            ++k;
        }
        d[j] = j ? d[j - 1] + x : x;
    }
}

// Fast form for a lane whose VPT values lie in one segment and inside the round: one segment lookup,
// straight-line extraction.  NARROW: every width of the round is <= 20 (one bit-field extract masks).
template <typename S, int VPT, bool NARROW, typename R>
__device__ __forceinline__ void delta_load_tile_fast(const R &sh, const DeltaGeom &g, uint32_t v0, uint32_t c0,
                                                     bool wide, S (&d)[VPT]) {
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win);
    uint32_t s, k;
    g.locate(v0, c0, s, k);
    const uint32_t bw = delta_seg_bw(sh, s);
    const S mn = delta_seg_min<S>(sh, s);
    const uint32_t w = bw >> 24;
    uint32_t bit = (bw & 0xFFFFFFu) + k * w;
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        S x;
        if constexpr (sizeof(S) == 8) {
            x = (wide ? field64(win32, bit, w) : (uint64_t)field32(win32, bit, w)) + mn;
        } else if constexpr (NARROW) {
            const uint32_t i = bit >> 5;
            x = __builtin_amdgcn_ubfe(__funnelshift_r(win32[i], win32[i + 1], bit & 31u), 0u, w) + mn;
        } else {
            x = field32(win32, bit, w) + mn;
        }
        bit += w;
        d[j] = j ? d[j - 1] + x : x;
    }
}

// The VPT values that start at value k of segment s (all inside the segment), running sums of type S.
// NARROW: every width of the round is <= 20 (one bit-field extract masks); wide: some width is above 32.
template <int VPT, bool NARROW, typename S, typename R>
__device__ __forceinline__ void delta_unpack_seg(const R &sh, uint32_t s, uint32_t k, bool wide, S (&d)[VPT]) {
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win);
    const uint64_t e = sh.seg_lo[s];
    const uint32_t bw = (uint32_t)e;
    S mn;
    if constexpr (sizeof(S) == 8) mn = (S)(((uint64_t)sh.seg_hi[s] << 32) | (e >> 32)); else mn = (S)(e >> 32);
    const uint32_t w = bw >> 24;
    uint32_t bit = (bw & 0xFFFFFFu) + __umul24(k, w);
#pragma unroll
    for (int j = 0; j < VPT; ++j) {
        S x;
        if constexpr (sizeof(S) == 8) {
            x = (wide ? field64(win32, bit, w) : (uint64_t)field32(win32, bit, w)) + mn;
        } else if constexpr (NARROW) {
            const uint32_t i = bit >> 5;
            x = __builtin_amdgcn_ubfe(__funnelshift_r(win32[i], win32[i + 1], bit & 31u), 0u, w) + mn;
        } else {
            x = field32(win32, bit, w) + mn;
        }
        bit += w;
        d[j] = j ? d[j - 1] + x : x;
    }
}

// Outcome of decoding one DELTA_BINARY_PACKED stream (uniform across the workgroup).
struct DeltaResult {
    int32_t err;         // P4S_E_NONE or the page error
    uint64_t aux0, aux1;
    uint64_t count;      // values handed to the sink
    uint64_t total;      // value count announced by the page header
    uint64_t end_pos;    // stream offset after the last byte consumed (size - bytes_left())
    uint64_t last;       // the running value after the last delta added (the last value decoded, as uint64)
};

// A decode that covers only a part of a page's stream (a page split over several workgroups, k_delta_split.cuh): it
// starts at a block header instead of the first one and ends after `deltas` deltas.  Parts other than the first do
// not know the value before their first delta: they decode with a running value of 0 and the caller adds the base
// to their values afterwards.
struct DeltaResume {
    uint64_t pos;        // page byte offset of the block header the part starts at
    uint64_t done;       // values of the page before the part's first delta (1 for the first part)
    uint32_t deltas;     // deltas the part decodes
    uint32_t last_part;  // the page's last part keeps the page's end-of-stream rules
};

// Parses the page header (ref :178-205; every thread does, the loads are uniform), hands value 0 to the sink and
// sets up the walker.  Returns false when nothing is left to decode (res holds the outcome).
// count = values to decode, p = page offset of the first block header.
template <typename T, typename Sink>
__device__ __forceinline__ bool delta_begin(const PageView &pv, uint64_t n_req, bool all, Sink &sink, DeltaResult &res,
                                            DeltaWalk &st, T &first, uint64_t &count, uint64_t &p) {
    constexpr int VPT = DeltaVpt<T>::value;
    (void)VPT;
    res.err = 0; res.aux0 = 0; res.aux1 = 0; res.count = 0; res.total = 0; res.end_pos = 0; res.last = 0;
    uint64_t block_size = 0, nmb64 = 0, total = 0, first_u = 0;
    p = 0;
    int32_t herr = 0;
    {
        uint32_t k;
        if ((k = uleb_global(pv, p, 10, &block_size)) == 0) herr = P4S_E_DELTA_HEADER_EOF;
        p += k;
        if (!herr) { if ((k = uleb_global(pv, p, 10, &nmb64)) == 0) herr = P4S_E_DELTA_HEADER_EOF; p += k; }
        if (!herr && nmb64 == 0) herr = P4S_E_DELTA_ZERO_MINIBLOCKS;
        if (!herr) { if ((k = uleb_global(pv, p, 10, &total)) == 0) herr = P4S_E_DELTA_HEADER_EOF; p += k; }
        if (!herr) { if ((k = uleb_global(pv, p, 10, &first_u)) == 0) herr = P4S_E_DELTA_HEADER_EOF; p += k; }
    }
    if (herr) { res.err = herr; return false; }
    res.total = total;
    const uint64_t emit = total < n_req ? total : n_req;      // values read_batch returns
    count = all ? total : emit;                // values actually decoded
    res.count = emit;
    res.end_pos = p;
    const uint64_t vpm = block_size / nmb64;
    // (deviation, raised where the reference parses the header: whatever n is)
    if (total > 1 && vpm == 0) { res.err = P4S_E_DELTA_BAD_SHAPE; res.aux0 = block_size; res.aux1 = nmb64; return false; }
    // A read that stops before the stream's last value still consumes the delta that follows the last value it
    // returns (ref :212-237: one delta is read after every value while values remain), so what is wrong with that
    // delta's block header or miniblock is reported, and the position afterwards counts it.  The sum it yields
    // lies past the caller's n and is dropped by the sink.
    const bool extra_delta = !all && count >= 1 && count < total;
    // the block header cannot hold more width bytes than the page has bytes
    if ((count > 1 || extra_delta) && nmb64 > pv.size) { res.err = P4S_E_DELTA_BLOCK_HEADER_EOF; return false; }
    if (count == 0) return false;
    // A stream announcing more than 2^32-1 values cannot belong to a page (deviation: the reference would try to
    // materialise them).  The reference meets whatever else is wrong with such a page first, while it decodes:
    // so decode 2^32-1 values and only then reject the page (delta_decode_stream, after its loop).
    if (count > 0xFFFFFFFFull) count = 0xFFFFFFFFull;
    first = (T)(uint64_t)zigzag64(first_u);
    sink.first(first);
    if (count == 1 && !extra_delta) return false;
#ifdef P4S_EXP_FILLONLY      // timing experiment: the kernel's store pattern without any decode work
    for (uint64_t idx = (P4S_EXP_FILLONLY == 2 ? 0 : 1) + (uint64_t)threadIdx.x * VPT; idx < count; idx += DELTA_THREADS * VPT) {
        T r[VPT];
        for (int j = 0; j < VPT; ++j) r[j] = (T)(idx + j);
        sink.tile(idx, r, idx + VPT <= count ? VPT : (uint32_t)(count - idx));
#if P4S_EXP_FILLONLY == 3          // a barrier every four stores, as the chunked decoder has
        if (((idx / (DELTA_THREADS * VPT)) & 3u) == 3u) __syncthreads();
#endif
    }
    return false;
#endif

    st.pos_bit = p * 8;
    st.widths_pos = 0;
    st.min_delta = 0;
    st.vpm = vpm < DELTA_VPM_CAP ? vpm : DELTA_VPM_CAP;
    st.nmb = (uint32_t)nmb64;
    st.mb_idx = st.nmb;
    st.mb_left = 0;
    st.cur_w = 0;
    st.deltas_left = (uint32_t)(extra_delta ? count : count - 1);
    st.full = count >= total;
    st.blk_vals = 0; st.maxblk = 0; st.q_tile = 0; st.cache_left = 0; st.cache_blocks = 0; st.last_d = 0;

    return true;
}

// Timing experiment (-DP4S_EXP_TIMING): cycles spent between the marks of a round, summed over the page's rounds
// and reported in the status record's aux fields instead of zeros (load | walk << 32, expand | decode << 32).
#ifdef P4S_EXP_TIMING
#define P4S_TICK(i) do { const uint64_t now_ = __builtin_readcyclecounter(); if ((i) > 0) tick_acc[(i) - 1] += now_ - tick_last; tick_last = now_; } while (0)
#else
#define P4S_TICK(i) do { } while (0)
#endif

// Decodes min(n_req, total) values of the stream `pv` and hands them to `sink`:
//   sink.first(v)                      - value 0 (called by every thread, uniform)
//   sink.tile(idx0, r, valid)          - values idx0 .. idx0+valid-1 of this thread (valid <= VPT, consecutive);
//                                        called by every thread of the workgroup for every tile, so a sink
//                                        may use barriers and wave scans.
// With `all` set the stream is decoded to its end even when n_req is smaller, so that end_pos is the start of
// whatever follows the stream (DELTA_LENGTH_BYTE_ARRAY / DELTA_BYTE_ARRAY, ref src/encoding.cc:277-290).
// WALK_ONLY: the block headers are followed to the stream's end (errors and end_pos as in a decode) but nothing is
// unpacked and the sink only sees the first value: what a reader of the NEXT stream needs to know about this one.
template <typename T, typename Sink, bool WALK_ONLY = false>
__device__ __forceinline__ void delta_decode_stream(DeltaShared &sh, const PageView &pv, uint64_t n_req, bool all,
                                                    Sink &sink, DeltaResult &res, const DeltaResume *part = nullptr) {
    constexpr int VPT = DeltaVpt<T>::value;
    const uint32_t tid0 = threadIdx.x;
    const uint32_t wave = first_u32(tid0 >> 6);    // wave-uniform: lets the walker branch on scalars
    DeltaWalk st;
    T first;
    uint64_t count, p;
    if (!delta_begin<T>(pv, n_req, all, sink, res, st, first, count, p)) return;

    const uint32_t vpm32 = (uint32_t)(st.vpm > 0xFFFFFFFFull ? 0xFFFFFFFFull : st.vpm);
    const DeltaGeom geom{vpm32, (vpm32 & (vpm32 - 1)) == 0, 31u - (uint32_t)__builtin_clz(vpm32 | 1u)};

    T carry = first;              // value preceding the next delta
    uint64_t done = 1;            // values written so far
    res.last = (uint64_t)first;
    if (part) {
        if (part->done > 1) { p = part->pos; st.pos_bit = p * 8; carry = 0; done = part->done; }
        st.deltas_left = part->deltas;
        if (!part->last_part) st.full = false;
    }
#ifdef P4S_EXP_TIMING
    uint64_t tick_acc[7] = {0, 0, 0, 0, 0, 0, 0}, tick_last = 0;
    if (tid0 == 0) { sh.dbg_iters = 0; sh.dbg_lds = 0; sh.dbg_rest = 0; }
#endif
    uint32_t buf = 0;
    // The window of round k+1 is fetched into registers as soon as the walk of round k has said where it starts,
    // stays in flight while round k is unpacked and stored, and is written to LDS when round k is done.
    WindowRegs<DELTA_WIN, DELTA_THREADS> wregs;
    int64_t wlo = window_fetch<DELTA_WIN, DELTA_THREADS>(pv, p, tid0, wregs);
    window_stage<DELTA_WIN, DELTA_THREADS>(sh.win, tid0, wregs);
    if (tid0 < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DELTA_WIN)[tid0] = 0;
    __syncthreads();
    for (;;) {
        // per-thread values derived from the thread index are recomputed every round: hoisted out of the loop
        // they would be held in registers (or spilled) across the whole page
        const uint32_t tid = opaque(tid0);
        const uint32_t lane = tid & 63u;
        P4S_TICK(0);
        P4S_TICK(1);
        P4S_TICK(1);
        if (wave == 0) {
            // the walk is a serial chain of dependent instructions: let it issue ahead of the other workgroups'
            // unpacking waves on this SIMD
#ifndef P4S_EXP_NOPRIO
            __builtin_amdgcn_s_setprio(3);
#endif
            delta_walk_round(sh, pv, wlo, st, DELTA_THREADS * VPT);
#ifndef P4S_EXP_NOPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
            st.make_uniform();
        }
        P4S_TICK(2);
        __syncthreads();
        if (!WALK_ONLY && sh.nblk) {
            delta_expand_blocks(sh, st.nmb, vpm32, tid, DELTA_THREADS);
            __syncthreads();
        }
        P4S_TICK(3);
        // sinks whose tile() needs many registers would only make the prefetched window spill: they fetch late
        int64_t wlo_next = 0;
        if constexpr (Sink::PREFETCH_WINDOW) {
            if (sh.more) wlo_next = window_fetch<DELTA_WIN, DELTA_THREADS>(pv, sh.next_pos, tid, wregs);
        }
#ifdef P4S_EXP_NODECODE      // timing experiment: window + walk only
        const uint32_t nv = 0;
        done += sh.round_vals;
#else
        const uint32_t nv = WALK_ONLY ? 0u : sh.round_vals;
        if (WALK_ONLY) done += sh.round_vals;
#endif
        const uint32_t c0 = sh.c0;
        const bool wide = sh.wide != 0;    // uniform: rounds of narrow miniblocks take the 32-bit extract
        // uniform: when every addend of the round fits 22 signed bits, the sums inside a tile are done in 32 bits
        // (exact: a tile holds at most 512 addends) and widened only when added to the running value
        const bool narrow = sizeof(T) == 8 && sh.big == 0;
        // One pass over the round's values, two tiles per iteration (one barrier, two independent dependency
        // chains).  S = type of the sums inside a tile: uint32 on the narrow path, else T.
        auto run_tiles = [&](auto s_tag) {
            using S = decltype(s_tag);
            constexpr bool WIDEN = sizeof(S) < sizeof(T);
            // a tile-local sum as a value of T (32-bit sums are exact two's complement values: widen with sign)
            auto widen = [](S v) -> T {
                if constexpr (WIDEN) return (T)(int64_t)(int32_t)v; else return (T)v;
            };
            // lanes never straddle segments when the first segment and all later ones hold multiples of VPT values
            const bool aligned_segs = (c0 % VPT == 0) && (vpm32 % VPT == 0);
            S *wt = reinterpret_cast<S *>(&sh.wtot[0][0]);     // [buf][tile][wave]
            constexpr uint32_t TILE = DELTA_THREADS * VPT;
            // 32-bit sums: two tiles per iteration (one barrier, two independent dependency chains); 64-bit sums
            // keep to one tile, two would not fit the register budget of 8 waves per SIMD
            constexpr bool PAIR = sizeof(S) == 4;
            uint32_t tb = 0;
            {
                // Regular rounds (every segment holds vpm values, vpm a power of two that divides the tile) go in
                // chunks of up to SLOTS tiles with two barriers per chunk: each wave unpacks its quarter of
                // every tile of the chunk (independent chains, the LDS latencies overlap), scans it and keeps the
                // tile-local running sums in registers; the per-wave totals of all tiles are exchanged at once,
                // every wave scans them (64-bit) for itself and stores.  A lane's position inside its segment
                // never changes and its segment index advances by a constant; tiles are full; the running
                // value stays in scalar registers.
                // (64-bit tile sums: two tiles per chunk, the register budget of 8 waves per SIMD holds no more)
                constexpr int SLOTS = sizeof(S) == 8 ? 2 : P4S_DELTA_SLOTS;
                const bool regular = geom.vpm_pow2 && c0 == vpm32 && vpm32 % VPT == 0 && vpm32 <= TILE &&
                                     (sizeof(S) == 8 || !wide);
                if (regular) {
                    const uint32_t lane_off = tid * VPT;
                    const uint32_t k0 = lane_off & (vpm32 - 1u);
                    const uint32_t s_lane = lane_off >> geom.vpm_shift;
                    S *wts = reinterpret_cast<S *>(&sh.wtot[0][0]);                      // [2][SLOTS * 4]
                    T c = uniform_value(carry);
                    // one chunk of NS full tiles, straight-line: the NS unpack chains and the NS wave scans are
                    // written interleaved so that their latencies overlap
                    auto chunk = [&](auto ns_tag) {
                        constexpr int NS = decltype(ns_tag)::value;
#ifdef P4S_EXP_DUMMYDECODE      // timing experiment: the chunk's stores without unpacking, scans or barriers
                        {
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                T r[VPT];
#pragma unroll
                                for (int j = 0; j < VPT; ++j) r[j] = c + (T)(tb + i * TILE + lane_off + j);
                                sink.tile_full(done + tb + i * TILE, lane_off, r);
                            }
                            tb += NS * TILE;
                            return;
                        }
#endif
                        P4S_TICK(4);
                        S *w = wts + buf * (SLOTS * 4);
                        S d[NS][VPT], inc[NS];
#pragma unroll
                        for (int i = 0; i < NS; ++i)
                            delta_unpack_seg<VPT, WIDEN>(sh, s_lane + ((tb + i * TILE) >> geom.vpm_shift), k0, wide, d[i]);
#pragma unroll
                        for (int i = 0; i < NS; ++i) inc[i] = d[i][VPT - 1];
                        wave_iscan_n(inc);
                        if (lane == 63) {
#pragma unroll
                            for (int i = 0; i < NS; ++i) w[i * 4 + wave] = inc[i];
                        }
                        // tile-local inclusive sums (the lane's own earlier values included)
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const S before = inc[i] - d[i][VPT - 1];
#pragma unroll
                            for (int j = 0; j < VPT; ++j) d[i][j] += before;
                        }
                        P4S_TICK(5);
                        __syncthreads();
                        P4S_TICK(6);
                        // the totals in value order (tile-major), as values of T, scanned by every wave for itself
                        T x = 0;
                        if (lane < (uint32_t)(NS * 4)) x = widen((S)w[lane]);
                        const T p = wave_iscan(x);
                        const T ex = p - x;
#pragma unroll
                        for (int i = 0; i < NS; ++i) {
                            const T base = c + bcast_value(ex, (int)(i * 4 + wave));
                            T r[VPT];
#pragma unroll
                            for (int j = 0; j < VPT; ++j) r[j] = base + widen((S)d[i][j]);
                            if constexpr (Sink::SHIFTED_STORES) {
                                // Value 0 of a page is its header's first value, so a lane's VPT values start one
                                // element past a 16-byte boundary.  Every lane stores the value before its own (the
                                // lane before it holds it; for lane 0 it is `base`) with its first VPT-1 values: 16-byte
                                // aligned stores.  The last value of the chunk has no lane after it: stored on its own.
                                T prevv = wave_shr1(r[VPT - 1]);
                                if (lane == 0) prevv = base;
                                T a[VPT];
                                a[0] = prevv;
#pragma unroll
                                for (int j = 1; j < VPT; ++j) a[j] = r[j - 1];
                                sink.tile_full(done + tb + i * TILE - 1, lane_off, a);
                                if (i == NS - 1 && tid == DELTA_THREADS - 1) sink.one(done + tb + i * TILE + lane_off + VPT - 1, r[VPT - 1]);
                            } else {
                                sink.tile_full(done + tb + i * TILE, lane_off, r);
                            }
                        }
                        c += bcast_value(p, NS * 4 - 1);
                        tb += NS * TILE;
                        buf ^= 1u;
                        P4S_TICK(7);
                    };
                    // largest chunks first; tiles the sink cannot take whole are left to the loop below
                    uint32_t ntile = (nv - tb) / TILE;
                    while (ntile > 0 && !sink.full_ok(done + tb, ntile * TILE)) --ntile;
                    for (; ntile >= (uint32_t)SLOTS; ntile -= SLOTS) chunk(std::integral_constant<int, SLOTS>{});
                    if constexpr (SLOTS >= 8) { if (ntile >= 4) { chunk(std::integral_constant<int, 4>{}); ntile -= 4; } }
                    if (ntile >= 2) { chunk(std::integral_constant<int, 2>{}); ntile -= 2; }
                    if (ntile >= 1) chunk(std::integral_constant<int, 1>{});
                    carry = c;
                }
            }
            for (; tb < nv; tb += (PAIR ? 2 : 1) * TILE) {
                const uint32_t v0a = tb + tid * VPT, v0b = v0a + TILE;
                S da[VPT], db[VPT];
                if (aligned_segs && tb + (PAIR ? 2 : 1) * TILE <= nv) {
                    delta_load_tile_fast<S, VPT, WIDEN>(sh, geom, v0a, c0, wide, da);
                    if constexpr (PAIR) delta_load_tile_fast<S, VPT, WIDEN>(sh, geom, v0b, c0, wide, db);
                } else {
                    delta_load_tile<S, VPT>(sh, geom, v0a, nv, c0, wide, da);
                    if constexpr (PAIR) delta_load_tile<S, VPT>(sh, geom, v0b, nv, c0, wide, db);
                }
                const S ta = da[VPT - 1];
                const S ia = wave_iscan(ta);
                S tb2 = 0, ib = 0;
                if constexpr (PAIR) { tb2 = db[VPT - 1]; ib = wave_iscan(tb2); }
                if (lane == 63) {
                    wt[buf * 8 + wave] = ia;
                    if constexpr (PAIR) wt[buf * 8 + 4 + wave] = ib;
                }
                __syncthreads();
                S ba = ia - ta, bb = ib - tb2, sa = 0, sb = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const S xa = wt[buf * 8 + i];
                    if ((uint32_t)i < wave) ba += xa;
                    sa += xa;
                    if constexpr (PAIR) {
                        const S xb = wt[buf * 8 + 4 + i];
                        if ((uint32_t)i < wave) bb += xb;
                        sb += xb;
                    }
                }
                const T all_a = widen(sa), all_b = widen(sb);
                T ra[VPT];
#pragma unroll
                for (int j = 0; j < VPT; ++j) ra[j] = carry + widen((S)(ba + da[j]));
                const uint32_t valid_a = v0a >= nv ? 0u : (nv - v0a < (uint32_t)VPT ? nv - v0a : (uint32_t)VPT);
                sink.tile(done + v0a, ra, valid_a);
                if constexpr (PAIR) {
                    T rb[VPT];
#pragma unroll
                    for (int j = 0; j < VPT; ++j) rb[j] = carry + all_a + widen((S)(bb + db[j]));
                    const uint32_t valid_b = v0b >= nv ? 0u : (nv - v0b < (uint32_t)VPT ? nv - v0b : (uint32_t)VPT);
                    sink.tile(done + v0b, rb, valid_b);
                }
                carry += all_a + all_b;
                buf ^= 1u;
            }
        };
        if (narrow) run_tiles(uint32_t{}); else run_tiles(T{});
        P4S_TICK(4);
        done += nv;
        res.last = (uint64_t)carry;
        const int32_t err = sh.err;
        if (err) { res.err = err; res.aux0 = sh.err_aux; return; }
        res.end_pos = sh.end_pos;     // the walker's position: size - bytes_left()
#ifdef P4S_EXP_TIMING
#if defined(P4S_EXP_TIMING_DECODE)
        res.aux0 = (tick_acc[4] & 0xFFFFFFFFull) | (tick_acc[5] << 32);
        res.aux1 = (tick_acc[6] & 0xFFFFFFFFull) | (tick_acc[3] << 32);
#elif defined(P4S_EXP_TIMING_WALK)
        res.aux0 = (uint64_t)sh.dbg_lds | (tick_acc[1] << 32);
        res.aux1 = (uint64_t)sh.dbg_rest | ((uint64_t)sh.dbg_iters << 32);
#else
        res.aux0 = (tick_acc[0] & 0xFFFFFFFFull) | (tick_acc[1] << 32);
        res.aux1 = (tick_acc[2] & 0xFFFFFFFFull) | (tick_acc[3] << 32);
#endif
#endif
        if (!sh.more) break;
        if constexpr (!Sink::PREFETCH_WINDOW) wlo_next = window_fetch<DELTA_WIN, DELTA_THREADS>(pv, sh.next_pos, tid, wregs);
        wlo = wlo_next;
        __syncthreads();          // everyone has read the round's window and shared state before they are rebuilt
        window_stage<DELTA_WIN, DELTA_THREADS>(sh.win, tid, wregs);
        if (tid < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DELTA_WIN)[tid] = 0;
        __syncthreads();
    }
    // the stream announced more values than delta_begin let through: nothing else was wrong with the first 2^32-1
    if ((all ? res.total : (res.total < n_req ? res.total : n_req)) > 0xFFFFFFFFull) {
        uint64_t block_size = 0, nmb64 = 0;
        const uint32_t k = uleb_global(pv, 0, 10, &block_size);
        uleb_global(pv, k, 10, &nmb64);
        res.err = P4S_E_DELTA_BAD_SHAPE; res.aux0 = block_size; res.aux1 = nmb64;
    }
}

// Sink of the plain decoder: value i of the page goes to out[i] (i < limit).
template <typename T>
struct DeltaStoreSink {
    T *out;
    uint64_t limit;
    __device__ __forceinline__ void first(T v) const {
        if (threadIdx.x == 0 && limit > 0) P4S_ST(out, v, 90);
    }
    static constexpr bool PREFETCH_WINDOW = true;
#ifdef P4S_EXP_NOSHIFT      // timing experiment: stores at the values' own (8 bytes off) addresses
    static constexpr bool SHIFTED_STORES = false;
#else
    static constexpr bool SHIFTED_STORES = true;      // tile_full may be called one element early (see the chunk loop)
#endif
    __device__ __forceinline__ void one(uint64_t idx, T v) const { P4S_ST(out + idx, v, 94); }
    // whether values base .. base+n-1 can be stored without per-lane bounds checks
    __device__ __forceinline__ bool full_ok(uint64_t base, uint32_t n) const { return base + n <= limit; }
    // a lane's VPT values at base + lane_off (base uniform)
    __device__ __forceinline__ void tile_full(uint64_t base, uint32_t lane_off, const T (&r)[DeltaVpt<T>::value]) const {
#ifdef P4S_EXP_NOSTORE
        if (r[0] != (T)0x5A5A5A5A5A5A5A5Aull) return;
#endif
#ifdef P4S_EXP_ALIGNSTORE     // timing experiment: what 16-byte aligned stores would buy (stores land one value early)
        __builtin_memcpy(out + base + lane_off - 1, r, sizeof(r));
#else
        // written once and not read again by this kernel: non-temporal 16-byte stores (about 2 % on the headline)
        if (P4S_INB(out + base + lane_off, sizeof(r), 91)) {
            typedef uint32_t v4u32_ __attribute__((ext_vector_type(4)));
            v4u32_ tmp[sizeof(r) / 16];
            __builtin_memcpy(tmp, r, sizeof(r));
#pragma unroll
            for (unsigned i = 0; i < sizeof(r) / 16; ++i)
                __builtin_nontemporal_store(tmp[i], reinterpret_cast<v4u32_ *>(out + base + lane_off) + i);
        }
#endif
    }
    __device__ __forceinline__ void tile(uint64_t idx0, const T (&r)[DeltaVpt<T>::value], uint32_t valid) const {
        constexpr int VPT = DeltaVpt<T>::value;
#ifdef P4S_EXP_NOSTORE       // timing experiment: keep the arithmetic alive, drop (almost) every store
        if (r[VPT - 1] != (T)0x5A5A5A5A5A5A5A5Aull) return;
#endif
        if (valid == (uint32_t)VPT && idx0 + VPT <= limit) {
            if (P4S_INB(out + idx0, sizeof(T) * VPT, 92)) __builtin_memcpy(out + idx0, r, sizeof(T) * VPT);
        } else {
            // (the lane's last value is never stored here: valid < VPT, or it lies at or past the limit.  With it in
            // this loop the compiler shares its store with the branch above, which then stores 12 + 4 bytes)
#pragma unroll
            for (int j = 0; j < VPT - 1; ++j)
                if ((uint32_t)j < valid && idx0 + j < limit) P4S_ST(out + idx0 + j, r[j], 93);
        }
    }
};

// Per-page record and per-part record of a batch whose pages are split over several workgroups (k_delta_split.cuh).
struct DeltaSplitPage {
    uint32_t nparts;     // 0: the page is decoded whole
    uint32_t redo;       // a part met an error: the page is decoded again, whole, for the error's exact outcome
    uint64_t count;      // values the page's status reports
};
struct DeltaPartRec {
    DeltaResume at;
    int32_t err;         // outcome of the part's decode
    uint32_t pad;
    uint64_t last;       // the part's last running value (relative to 0 for parts after the first)
    uint64_t end_pos;
};

template <typename T>
// 8 waves per SIMD and <= 80 SGPRs: eight 256-thread workgroups (pages) resident per CU.
// With `parts` null: one workgroup per page.  With `parts` set: blockIdx.x = page * fan + s; part s of a split page
// is decoded like a page of its own (its outcome goes to its record; k_delta_split_finish adds the bases and writes
// the page's status), a page that is not split is decoded whole by s == 0.  With `only_redo` set: only pages whose
// record asks for it are decoded (whole), the others are left as they are.
__global__ __launch_bounds__(DELTA_THREADS, 8) void k_delta_binary_packed(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        T *__restrict__ d_out, p4s_page_status *__restrict__ status,
        DeltaPartRec *__restrict__ parts, DeltaSplitPage *__restrict__ split, uint32_t fan, uint32_t only_redo) {
    __shared__ DeltaShared sh;
    uint32_t page = blockIdx.x;
    DeltaPartRec *rec = nullptr;
    if (split) {
        if (only_redo) {
            if (!split[page].redo) return;
        } else {
            page = blockIdx.x / fan;
            const uint32_t s = blockIdx.x % fan;
            const uint32_t np = split[page].nparts;
            if (np == 0) { if (s != 0) return; }
            else { if (s >= np) return; rec = &parts[(uint64_t)page * fan + s]; }
        }
    }
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    DeltaStoreSink<T> sink{d_out + pd.out_offset, pd.num_values};
    DeltaResult res;
    if (rec) {
        const DeltaResume at = rec->at;
        delta_decode_stream<T>(sh, pv, pd.num_values, false, sink, res, &at);
        if (threadIdx.x == 0) { rec->err = res.err; rec->last = res.last; rec->end_pos = res.end_pos; }
        return;
    }
    delta_decode_stream<T>(sh, pv, pd.num_values, false, sink, res);
#ifdef P4S_EXP_TIMING
    if (threadIdx.x == P4S_EXP_TIMING) {       // the thread whose clock readings are reported
#else
    if (threadIdx.x == 0) {
#endif
        if (res.err) write_status(&status[page], res.err, 0, res.aux0, res.aux1, 0, 0);
#ifdef P4S_EXP_TIMING
        else write_status(&status[page], P4S_E_NONE, res.count, res.aux0, res.aux1, res.end_pos, 0);
#else
        else write_status(&status[page], P4S_E_NONE, res.count, 0, 0, res.end_pos, 0);
#endif
    }
}

}  // namespace p4s
