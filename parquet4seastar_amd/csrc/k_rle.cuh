// RLE / bit-packed hybrid decode (ref: RleDecoder::GetBatch / NextCounts,
// include/parquet4seastar/rle_encoding.hh:282-348) and the three decoders built
// on it: level_decoder (encoding.hh:42-82, encoding.cc:26-79), dict_decoder
// (encoding.cc:468-508) and rle_decoder_boolean (encoding.cc:510-516).
//
// One 256-thread workgroup per page (eight resident per CU), in rounds that are software pipelined: while wave 0 walks
// the run headers of round k+1, the other waves (and wave 0 once it is done) expand round k; windows and run tables
// are double buffered in LDS.  A round is:
//   1. all waves stage an 8 KiB window of the stream in LDS (16 B per lane and piece, coalesced);
//   2. wave 0 walks the run headers.  The only serial fact of the format is where the next header starts: every lane
//      parses the ULEB128 header that WOULD start at byte cur+lane (all 64 offsets at once) and the true chain through
//      the 64-byte span is followed with scalar readlane steps, which also add up the run lengths, so that the run
//      table (first output index, first expansion item, payload offset of every run) is complete when the walk is
//      (rle_walk_fast).  That pass only takes runs about which nothing can go wrong; a round that starts with anything
//      else (a header of three bytes or more, the last 80 bytes of the stream, a truncated run, the run that holds the
//      n-th value) is walked by rle_find_headers, which only lists header offsets, and
//   3. placed by all waves (rle_place_runs): one run per lane re-parses its header with the full rules of the
//      reference, a workgroup prefix sum of the run lengths gives every run its first output index, and the first run
//      that ends the decode (count reached, corrupt or truncated run) is found with one LDS atomic minimum;
//   4. expansion in items of 8 values of ONE run, counted from the run's first value (so a literal item starts on a
//      byte boundary of the stream and its fields sit at compile-time bit positions), in tiles of 64 items that the
//      waves claim from an LDS counter: each lane finds its item's run by binary search in the LDS table, then
//      broadcasts the repeated value or unpacks the literals from the LDS window (LDS reads need no alignment on this
//      hardware: the item's w bytes are read where they lie).  Dictionary pages gather through the index on the spot.
// The walk is one dependent chain per page (about 800 cycles per 64-byte span); it bounds streams of short runs.
// Level streams of narrow values (width <= 4) use a second LDS layout (3 KiB windows, 512-run tables), a walk that
// finds all headers of a span at once by pointer jumping across the lanes (rle_walk_dense) and an expansion that finds
// an item's run through a bit map of run starts and stages the values as bytes (rle_expand_dense).
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int RLE_THREADS = 256;
constexpr int RLE_WIN = 4096;
constexpr int RLE_SLACK = 16;
constexpr int RLE_MAXRUNS = 512;
constexpr int RLE_VPT = 8;
// What part of a hybrid stream one decode covers.  A whole page is {0, 0, 0, ~0}.  A page split over several
// workgroups gives each of them the stream offset of a run header, the output index of that run's first value, the
// output range the workgroup writes (values before out_begin are decoded past, not stored; the decode ends at the
// n handed to rle_decode_stream) and, for the counting pass, the stream offset at which the header chain is cut.
struct RleSpan {
    uint64_t cur0;        // stream offset of the first run header
    uint64_t out0;        // output index of that run's first value
    uint64_t out_begin;   // first output index that is stored (>= out0)
    uint64_t in_limit;    // run headers at or past this stream offset are not taken (~0: no limit)
};

// LDS of one decode.  The kernel holds a union of two layouts and picks one per page by the bit width; the scalar
// members come first so that they sit at the same offsets in both.
struct RleSharedScalars {
    uint64_t wtot[4];                    // per-wave totals of the placing pass
    uint64_t next_cur;                   // stream offset of the next unread header
    unsigned long long errkey;           // dict: min over (position << 32 | bad index)
    uint32_t wtot_items[4];
    uint32_t n_items[2];                 // expansion items of the round
    uint32_t nhdr;                       // headers found this round
    uint32_t stream_end;                 // the header chain ended (clean end of stream or unreadable header)
    uint32_t kstop;                      // first run that ends the decode (DRUNS + 1: none)
    uint32_t nruns[2];                   // runs to expand
    uint32_t tile_next;                  // next unclaimed expansion tile of the round
    uint32_t out_end;                    // values produced after this round
    uint32_t finished;
    uint32_t irregular;                  // a run ended the decode before n values (unreadable header, truncated run)
    uint32_t need_place;                 // the round's headers are listed and still have to be placed
    uint32_t level_hits;                 // levels: values equal to max level
    uint32_t level_above;                // levels: values above max level (column_chunk_reader.hh:124-133 rejects them)
#ifdef P4S_EXP_TIMING
    uint64_t dbg[3];                     // cycle counts of the timing experiment
    uint32_t cnt_iter, cnt_fallback, cnt_rounds, cnt_pad;
#endif
};

// Which run an expansion item belongs to, without a search: bit x of item_bits is set when item x is the first of its
// run (runs without values are left out of such a round's table); with rank[j] = runs that start before item 64 j (a
// prefix sum over the words' population counts, which every expanding wave takes for itself at the start of a round),
// item x lies in run rank[x / 64] + popcount(bits of word x / 64 up to and including bit x % 64) - 1.
constexpr int RLE_ITEM_CAP = 4096;    // items (of 8 values) a round may have for that; rounds with more are searched
template <bool BITS_> struct RleBitsExtra {};
template <> struct RleBitsExtra<true> {
    uint64_t item_bits[2][RLE_ITEM_CAP / 64];
    uint32_t bits_ok[2];              // item_bits describes the round (the walk made the table, without runs of no values, and it has at most RLE_ITEM_CAP items)
};
template <bool DENSE_> struct RleDenseExtra {};
template <> struct RleDenseExtra<true> {
    // per wave: the values of one tile of 64 items on their way out, a byte each (a value of a stream of up to 8 bits
    // is a byte, also the unmasked one of a repeated run)
    alignas(16) uint8_t stage8[RLE_THREADS / 64][RLE_VPT * 64 + 16];
};
// Level streams of narrow values (width <= 4; many run headers per 64-byte span): two windows of the stream,
// restaged wherever the header walk has got to, and two run tables, so that round k + 1 is walked while round k is
// expanded.  A round ends when its window or its run table is used up.
template <int DWIN_, int DRUNS_, bool DENSE_>
struct RleSharedT : RleSharedScalars, RleBitsExtra<true>, RleDenseExtra<DENSE_> {
    static constexpr int DWIN = DWIN_;        // window bytes
    static constexpr int DRUNS = DRUNS_;      // runs per round
    static constexpr bool DENSE = DENSE_;     // many headers per 64-byte span expected
    static constexpr bool BITS = true;        // the walks of this layout mark the items that start a run
    static constexpr int RINGSIZE = 0;        // windows are linear
    static constexpr int RPT = (DRUNS + RLE_THREADS - 1) / RLE_THREADS;      // runs per thread in the placing pass
    alignas(16) uint8_t win[2][DWIN + RLE_SLACK];
    uint32_t run_out[2][DRUNS + 2];   // first output index of each run (+ end sentinel)
    uint32_t run_item[2][DRUNS + 2];  // first expansion item of each run (+ end sentinel); an item is 8 values of one run
    uint16_t run_info[2][DRUNS + 2];  // (window offset of the payload) << 1 | is_literal
    // (rle_find_headers lists the window offsets of a round's run headers, in stream order, as uint16 in the memory of
    // run_item[b]: the placing pass has read them all before it writes the first item count)
};
using RleSharedDense = RleSharedT<3072, 512, true>;
static_assert(sizeof(RleSharedDense) <= 20480, "eight workgroups per CU");
// Dictionary pages of wider indices keep linear windows too, 8 KiB of them with a small run table: what they cost are
// the lookups (the L1 fill path of the CU), and long rounds of mostly literal runs keep all four waves busy with them.
using RleSharedDict = RleSharedT<8192, 128, false>;      // (measured: 7680 / 160 is 1 % slower on uniform indices)
static_assert(sizeof(RleSharedDict) <= 20480, "eight workgroups per CU");

// Wider values: the stream passes through a ring of three 2 KiB tiles at fixed stream offsets (tile t = bytes
// [base + t * 2048, base + (t + 1) * 2048) of the stream, in ring slot t % 3), so that what is fetched next never
// depends on how far the header walk has got: the loads of a tile are issued a whole round before it is needed.
// A round is the runs whose headers the walk found in one tile (at most DRUNS of them); their payload may reach
// into the next tile, which is why three are kept.  The ring's first bytes are mirrored behind its end, so that a
// read of up to RLE_MIRROR bytes may start anywhere in it.  Rounds that are anything but plain (the end of the stream,
// headers of four bytes or more, the run that holds the n-th value, every error) use the same memory as ONE linear
// window of RLE_RING bytes, read afresh from where the walk stands, and the passes that know all the rules
// (rle_find_headers, rle_place_runs).
constexpr int RLE_TILE = 2048;
constexpr int RLE_TILE_SHIFT = 11;
constexpr int RLE_RING_TILES = 3;
constexpr int RLE_RING = RLE_TILE * RLE_RING_TILES;
constexpr int RLE_MIRROR = 64;
struct RleRoundRec {                  // what the walk of a round hands to its expansion (and to the next walk)
    uint64_t next_cur;                // stream offset of the first header the round did not take
    uint32_t out_end;                 // values produced after the round
    int32_t tile;                     // the tile its headers lie in
    uint32_t stop;                    // the walk cannot go on from next_cur: a linear round takes over there
    uint32_t bits;                    // item_bits describes the round (it has at most RLE_ITEM_CAP items)
};
struct RleShared : RleSharedScalars {
    static constexpr int DWIN = RLE_RING;
    static constexpr int DRUNS = 176;
    static constexpr bool DENSE = false;
    static constexpr int RINGSIZE = RLE_RING;
    static constexpr int RPT = 1;
    alignas(16) uint8_t win[1][RLE_RING + RLE_MIRROR];
    uint32_t run_out[2][DRUNS + 2];
    uint32_t run_item[2][DRUNS + 2];
    uint16_t run_info[2][DRUNS + 2];  // (payload offset from the start of the round's tile) << 1 | is_literal
    uint16_t hdr_list[DRUNS + 256];   // tile offsets of the headers a ring walk has found (it may overshoot DRUNS by a register's worth)
    RleRoundRec rnd[2];
    uint64_t item_bits[2][RLE_ITEM_CAP / 64];
    uint32_t tile_ctr[2];             // per run table: the next unclaimed tile of 64 expansion items
    // per wave: the values of one tile of 64 items on their way out.  The items of a tile cover one contiguous range of
    // the output; staged here they leave in stores of 16 contiguous bytes per lane, 16-byte aligned, whatever the
    // runs' lengths (stores of 32 bytes per lane, or of single values at run ends, leave cache lines half written and
    // cost more than twice as much)
    alignas(16) uint32_t stage[RLE_THREADS / 64][RLE_VPT * 64 + 4];
};
static_assert(sizeof(RleShared) <= 20480, "eight workgroups per CU");
static_assert(RLE_MIRROR >= RLE_SLACK && RLE_TILE == (1 << RLE_TILE_SHIFT), "ring geometry");
#ifndef P4S_RLE_DENSE_MAX_WIDTH
#define P4S_RLE_DENSE_MAX_WIDTH 4
#endif
constexpr uint32_t RLE_DENSE_MAX_WIDTH = P4S_RLE_DENSE_MAX_WIDTH;     // bit widths up to this take the dense layout

// A run header as the reference's NextCounts sees it (rle_encoding.hh:321-348).
struct RleHeader {
    uint32_t hl;        // bytes of the ULEB128 header
    uint32_t count;     // header >> 1
    bool lit;
    bool fail;          // NextCounts returns false here: the stream ends
    uint32_t skip;      // bytes from this header to the next one (saturated)
};

// `x` = the 5 bytes at the header position (little endian), `avail` = stream bytes from that position on
// (<= 0 past the end), `vbytes` = ceil(w / 8), `tbytes` = sizeof(output element).
__device__ __forceinline__ RleHeader rle_parse(uint64_t x, int32_t avail, uint32_t w, uint32_t vbytes, uint32_t tbytes) {
    RleHeader h;
    const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
    const uint32_t stop = ~lo & 0x80808080u;
    // index of the first byte without a continuation bit (5: none within the 5 bytes a uint32 may take)
    const uint32_t t = stop ? ((uint32_t)__builtin_ctz(stop) >> 3) : ((hi & 0x80u) ? 5u : 4u);
    h.hl = t + 1;
    h.fail = (t >= 5) || ((int32_t)h.hl > avail);                          // bit_stream_utils.hh:405-422
    uint32_t v = (lo & 0x7Fu) | ((lo >> 1) & 0x3F80u) | ((lo >> 2) & 0x1FC000u) | ((lo >> 3) & 0xFE00000u) | ((hi & 0x7Fu) << 28);
    // keep the 7-bit groups up to and including byte t
    const uint32_t keep = t >= 4 ? 0xFFFFFFFFu : ((1u << (7 * (t + 1))) - 1u);
    v &= keep;
    h.count = v >> 1;
    h.lit = v & 1u;
    if (h.lit) {
        if (h.count > 0x0FFFFFFFu) h.fail = true;                           // rle_encoding.hh:332-334
        const uint64_t s = (uint64_t)h.hl + (uint64_t)h.count * w;
        h.skip = s > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)s;
    } else {
        if (vbytes > tbytes || (int32_t)(h.hl + vbytes) > avail) h.fail = true;   // bit_stream_utils.hh:370-377
        h.skip = h.hl + vbytes;
    }
    return h;
}

// 5 bytes at window offset `rel` (the window keeps RLE_SLACK readable bytes past its end)
// (LDS reads need no alignment on this hardware: one 8-byte read)
__device__ __forceinline__ uint64_t rle_win40(const uint32_t *win32, uint32_t rel) {
    uint64_t x;
    __builtin_memcpy(&x, reinterpret_cast<const uint8_t *>(win32) + rel, 8);
    return x;
}

__device__ __forceinline__ int32_t rle_avail(const PageView &pv, int64_t pos) {
    const int64_t a = (int64_t)pv.size - pos;
    return a > 0x7FFFFFFF ? 0x7FFFFFFF : (a < -0x7FFFFFFF ? -0x7FFFFFFF : (int32_t)a);
}

// Pass 2 of a round (wave 0, uniform control flow): appends the window offsets of the real run headers to
// the header list.  `cur` = stream offset of the next header.  A run whose payload crosses the window end is left to
// the next window unless it is the first of the round (then the part past the window is read from global memory).
template <typename SH>
__device__ __forceinline__ void rle_find_headers(SH &sh, uint32_t b, const PageView &pv, int64_t wlo, uint64_t &cur,
                                                 uint32_t w, uint32_t tbytes, uint64_t in_limit) {
    const uint32_t lane = lane_id();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win[b]);
    const uint32_t vbytes = (w + 7) >> 3;
    uint32_t nh = 0;
    bool stream_end = false, force_slow = false;
    for (;;) {
        if (cur >= pv.size || cur >= in_limit) { stream_end = true; break; }   // clean end of stream (or of this decode's part of it)
        const int64_t rel64 = (int64_t)cur - wlo;
        if (rel64 + 72 > SH::DWIN) break;                                       // span not staged: next window
        if (nh + 64 > SH::DRUNS) break;
        const uint32_t rel0 = first_u32((uint32_t)rel64);
        const uint32_t rel = rel0 + lane;
        const int32_t avail0 = rle_avail(pv, (int64_t)cur);                     // uniform; lane l has avail0 - l bytes left
        const uint64_t x = rle_win40(win32, rel);
        uint32_t skip;
        uint64_t failmask, longmask = 0;
        if (!force_slow && avail0 >= 64 + 8) {
            // Fast parse: headers of one or two bytes, span well inside the stream (no end-of-stream checks).
            // A header of three or more bytes on the real chain ends the iteration; the next one parses in full.
            const uint32_t b0 = (uint32_t)x & 0xFFu, b1 = ((uint32_t)x >> 8) & 0xFFu;
            const uint32_t two = b0 >> 7;
            const uint32_t v = (b0 & 0x7Fu) | (two ? (b1 & 0x7Fu) << 7 : 0u);
            const uint32_t hl = 1u + two;
            const bool lit = v & 1u;
            skip = lit ? hl + (v >> 1) * w : hl + vbytes;
            failmask = __ballot(!lit && vbytes > tbytes);                       // bit_stream_utils.hh:370-377
            longmask = __ballot((two & (b1 >> 7)) != 0);
        } else {
            const RleHeader h = rle_parse(x, avail0 - (int32_t)lane, w, vbytes, tbytes);
            skip = h.skip;
            failmask = __ballot(h.fail);
        }
        force_slow = false;
        // offsets at or past the end of the stream hold no header (the chain ends cleanly when it lands on one), and
        // headers at or past in_limit belong to another decode (cur lies below both here)
        const uint64_t end_at = in_limit < pv.size ? in_limit : pv.size;
        const uint64_t limmask = end_at - cur < 64 ? ~0ull << (end_at - cur) : 0ull;
        // a run whose payload ends past the window (skip saturates well below 2^32 - DWIN)
        const uint64_t stopmask = failmask | longmask | limmask | __ballot(rel + skip > (uint32_t)SH::DWIN);
        // ---- follow the real chain through the span
        uint64_t M = 0;
        uint32_t r = 0;
        bool hit_fail = false, crossing = false, at_limit = false;
        while (r < 64) {
            if ((stopmask >> r) & 1ull) break;
            M |= 1ull << r;
            r += bcast_u32(skip, (int)r);
        }
        if (r < 64) {
            if ((limmask >> r) & 1ull) {
                at_limit = true;                // the stream ends here, or the chain goes on in another decode's part of it
            } else if ((longmask >> r) & 1ull) {
                force_slow = true;              // re-parse from this header with the full parser
            } else if ((failmask >> r) & 1ull) {
                M |= 1ull << r;                 // recorded: the placing pass ends the decode there
                hit_fail = true;
            } else if (nh == 0 && M == 0) {
                M = 1;                          // first run of the round: taken, its tail is read from global memory
                r = bcast_u32(skip, 0);
            } else {
                crossing = true;
            }
        }
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
        if ((M >> lane) & 1ull) reinterpret_cast<uint16_t *>(sh.run_item[b])[nh + rank] = (uint16_t)rel;
        nh += (uint32_t)__popcll(M);
        if (hit_fail) { stream_end = true; break; }
        cur += r;
        if (at_limit) { stream_end = true; break; }
        if (crossing) break;
    }
    if (lane == 0) {
        sh.nhdr = nh;
        sh.stream_end = stream_end ? 1u : 0u;
        sh.next_cur = cur;
    }
}

// v_writelane_b32: `old` with lane `lane` (uniform) replaced by the uniform `val`
extern "C" __device__ uint32_t rle_writelane(uint32_t val, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");

struct RleWalkTicks { uint64_t t0, t1, t2; };      // cycle counts of the walk's phases (timing experiment)

__device__ __forceinline__ uint32_t rle_ring_wrap(uint32_t at) { return at >= (uint32_t)RLE_RING ? at - (uint32_t)RLE_RING : at; }

// The header walk of a ring round (one wave, uniform control flow): the runs whose headers lie in the tile that
// holds `cur`, up to DRUNS of them, go into run table `b`; sh.rnd[b] says where the walk stands afterwards.
// Where the next header starts is the only serial fact of the format, and a single wave issues an instruction every
// five to eight cycles at best, so the chain is written for the fewest instructions per header and for the scalar
// unit (the vector unit is what the expanding waves are short of).  The wave holds the tile in registers (lane l of
// register i = tile bytes 256 i + 4 l .. + 3, and the same shifted by one lane), so the four bytes at any tile offset
// are one v_alignbyte and one v_readlane away, with no memory access in the loop; a header of one or two bytes is
// parsed and stepped over in 16 to 25 instructions (inline assembly: the compiler's version of the same loop takes
// twice as many and five branches).  The headers' tile offsets are listed in LDS; afterwards the lanes parse one
// header each and two wave prefix sums give every run its first output index and first expansion item, so the run
// table is complete when the walk is.  Headers of three bytes are looked at outside the loop.  The walk stops
// (rnd.stop: a linear round takes over at rnd.next_cur) in front of a header of four bytes or more, in front of a
// run that the stream cannot hold, in front of the run that holds the n-th value, when the tile lies within 2 KiB of
// the end of the stream (a one-byte header's run is at most 1 + 63 * 32 bytes, so up to there the loop need not
// check), or when the chain has left the tiles that are staged.
// `staged` = index of the last tile the ring holds; `cur` / `out` = the walk's state (next header, values so far).
__device__ __forceinline__ void rle_walk_ring(RleShared &sh, uint32_t b, int64_t base, int32_t staged, uint64_t end_at,
                                              uint64_t &cur, uint32_t &out, uint64_t n, uint32_t w, RleWalkTicks &ticks) {
    const uint32_t lane = lane_id();
    const uint8_t *ring = sh.win[0];
#ifdef P4S_EXP_TIMING_WALK
    uint64_t wt_last = __builtin_readcyclecounter();
#define RLE_WTICK(i) do { const uint64_t n_ = __builtin_readcyclecounter(); ticks.t##i += n_ - wt_last; wt_last = n_; } while (0)
#else
#define RLE_WTICK(i) do { } while (0)
#endif
    const uint32_t vbytes = (w + 7) >> 3;
    const int64_t d64 = (int64_t)cur - base;
    const int64_t tq = d64 >> RLE_TILE_SHIFT;
    bool stop = tq > (int64_t)staged;
    const int32_t t = (int32_t)first_u32((uint32_t)(stop ? (int64_t)staged : tq));
    const int64_t tile_off = base + ((int64_t)t << RLE_TILE_SHIFT);
    // the run of every one-byte header of the tile then ends inside the stream
    if ((uint64_t)(tile_off + RLE_TILE + 2048) > end_at) stop = true;
    uint32_t nr = 0, items = 0;
    bool use_bits = false;
    if (!stop) {
        const uint32_t origin = first_u32((uint32_t)(t % RLE_RING_TILES)) * (uint32_t)RLE_TILE;
        const bool next_ok = t + 1 <= staged;
        const uint32_t next_origin = origin + RLE_TILE >= (uint32_t)RLE_RING ? 0u : origin + (uint32_t)RLE_TILE;
        // ---- the tile's bytes: X[i] = one dword per lane, Y[i] = the dword behind it (the next lane's; behind the
        // tile's last dword comes the next tile, and where that is not staged yet a header that reaches into it ends the round)
        uint32_t X[RLE_TILE / 256];
        {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(ring + origin) + lane;
#pragma unroll
            for (int i = 0; i < RLE_TILE / 256; ++i) X[i] = src[i * 64];
        }
        // (continuation bits where the next tile is missing: a header that reaches into it is then not taken by the loop)
        const uint32_t edge15 = next_ok ? first_u32(*reinterpret_cast<const uint32_t *>(ring + next_origin)) : 0x80808080u;
        RLE_WTICK(0);
        // ---- the chain
        uint32_t pos = first_u32((uint32_t)(d64 - ((int64_t)t << RLE_TILE_SHIFT)));      // tile offset of the next header
        // tile offset of the end of the stream (the tile ends at least 2 KiB before it)
        const uint32_t end_rel = first_u32(end_at - (uint64_t)tile_off > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(end_at - (uint64_t)tile_off));
        // the last tile offset at which a header's four bytes are all staged
        const uint32_t pos_lim = next_ok ? (uint32_t)RLE_TILE : (uint32_t)RLE_TILE - 3u;
        // the headers' tile offsets go to sh.hdr_list (`la` = LDS address of the next entry); the chain may run up to 256
        // headers past DRUNS before it is stopped, the list has room for them and they are walked again next round
        const uint32_t la0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t *)sh.hdr_list;
        uint32_t la = la0;
        for (;;) {
            bool esc = false;
#pragma unroll
            for (int i = 0; i < RLE_TILE / 256; ++i) {
                if (pos < (uint32_t)(i + 1) * 256u) {
                    if (first_u32(la) - la0 >= 2u * (uint32_t)RleShared::DRUNS) break;      // the table is full
                    uint32_t Yi = dpp_mov<0x130, 0xF>(X[i]);                          // wave_shl:1: the next lane's dword
                    if (lane == 63) Yi = i < RLE_TILE / 256 - 1 ? bcast_u32(X[i < RLE_TILE / 256 - 1 ? i + 1 : i], 0) : edge15;
                    uint32_t h, c, t0, st, vh, vd;
                    asm volatile(
                        "s_mov_b32 %[st], 0\n\t"
                        "1:\n\t"
                        "s_bfe_u32 %[t0], %[pos], 0x60002\n\t"               // lane that holds the dword of byte pos
                        "v_alignbyte_b32 %[vh], %[Y], %[X], %[pos]\n\t"     // every lane: its 8 bytes shifted down by pos % 4 bytes
                        "v_readlane_b32 %[h], %[vh], %[t0]\n\t"             // the four bytes at pos
                        "s_bitcmp1_b32 %[h], 7\n\t"
                        "s_cbranch_scc1 3f\n\t"
                        "s_bfe_u32 %[c], %[h], 0x60001\n\t"                 // one-byte header: count = bits 6..1
                        "s_bitcmp1_b32 %[h], 0\n\t"
                        "s_mul_i32 %[c], %[c], %[w]\n\t"
                        "s_cselect_b32 %[c], %[c], %[vb]\n\t"               // literal: count * w bytes; repeated: ceil(w / 8)
                        "2:\n\t"
                        "v_mov_b32 %[vd], %[pos]\n\t"
                        "ds_write_b16 %[la], %[vd]\n\t"
                        "v_add_u32 %[la], 2, %[la]\n\t"
                        "s_add_u32 %[pos], %[pos], %[c]\n\t"
                        "s_add_u32 %[pos], %[pos], 1\n\t"
                        "s_cmp_lt_u32 %[pos], %[lim]\n\t"
                        "s_cbranch_scc1 1b\n\t"
                        "s_branch 9f\n\t"
                        "3:\n\t"                                            // two bytes or more
                        "s_bitcmp1_b32 %[h], 15\n\t"
                        "s_cbranch_scc1 8f\n\t"
                        "s_bfe_u32 %[c], %[h], 0x60001\n\t"
                        "s_bfe_u32 %[t0], %[h], 0x70008\n\t"
                        "s_lshl_b32 %[t0], %[t0], 6\n\t"
                        "s_or_b32 %[c], %[c], %[t0]\n\t"                    // count = bits 6..1 | bits 14..8 << 6
                        "s_bitcmp1_b32 %[h], 0\n\t"
                        "s_mul_i32 %[c], %[c], %[w]\n\t"
                        "s_cselect_b32 %[c], %[c], %[vb]\n\t"
                        "s_add_u32 %[c], %[c], 1\n\t"                       // the second header byte
                        "s_add_u32 %[t0], %[pos], %[c]\n\t"
                        "s_cmp_lt_u32 %[t0], %[endrel]\n\t"                 // the run ends inside the stream
                        "s_cbranch_scc1 2b\n\t"
                        "8:\n\t"
                        "s_mov_b32 %[st], 1\n\t"
                        "9:\n\t"
                        "s_waitcnt lgkmcnt(0)\n\t"
                        : [pos] "+s"(pos), [la] "+v"(la), [h] "=&s"(h), [c] "=&s"(c), [t0] "=&s"(t0), [st] "=&s"(st),
                          [vh] "=&v"(vh), [vd] "=&v"(vd)
                        : [X] "v"(X[i]), [Y] "v"(Yi), [w] "s"(w), [vb] "s"(vbytes), [endrel] "s"(end_rel), [lim] "n"((i + 1) * 256)
                        : "memory", "scc");
                    if (st) { esc = true; break; }
                }
            }
            if (!esc) break;                                  // the tile is done, or the table is full
            uint32_t cnt = (first_u32(la) - la0) >> 1;
            uint16_t *list = sh.hdr_list;
            // a header the loop does not take (pos < RLE_TILE here): three bytes are taken here
            if (cnt >= (uint32_t)RleShared::DRUNS) break;
            if (pos >= pos_lim) break;                              // its bytes are not all staged: the round ends in front of it    // its bytes are not all staged: the round ends in front of it
            const uint32_t at = rle_ring_wrap(origin + pos);
            uint32_t x;
            __builtin_memcpy(&x, ring + at, 4);                     // (the mirror behind the ring covers the wrap)
            x = first_u32(x);
            const uint32_t c0 = (x >> 7) & 1u, c1 = c0 & (x >> 15), c2 = c1 & (x >> 23);
            const uint32_t v = (x & 0x7Fu) | (c0 ? (x >> 1) & 0x3F80u : 0u) | (c1 ? (x >> 2) & 0x1FC000u : 0u);
            const uint32_t hl = 1u + c0 + c1;
            const uint32_t sk = ((v & 1u) ? (v >> 1) * w : vbytes) + hl;                // < 2^25 + 3
            if (c2 || (uint64_t)(tile_off + (int64_t)pos + (int64_t)sk) > end_at) { stop = true; break; }
            list[cnt] = (uint16_t)pos;
            la += 2;
            pos += sk;
        }
        uint32_t cnt = (first_u32(la) - la0) >> 1;
        if (cnt > (uint32_t)RleShared::DRUNS) {              // the headers past the table's end: next round
            cnt = (uint32_t)RleShared::DRUNS;
            pos = first_u32((uint32_t)sh.hdr_list[RleShared::DRUNS]);
        }
        uint16_t *list = sh.hdr_list;
        RLE_WTICK(1);
        // ---- the runs' places, 64 headers at a time
        const uint32_t left = first_u32(n - out > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(n - out));   // (out < n)
        uint32_t sum = 0;                 // values of the runs placed so far
        bool cut = false;
        static_assert(RLE_ITEM_CAP / 64 == 64, "one word of item_bits per lane");
        sh.item_bits[b][lane] = 0;
        for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
            const bool real = k0 + lane < cnt;
            const uint32_t hp = real ? (uint32_t)list[k0 + lane] : 0u;
            uint32_t x;
            __builtin_memcpy(&x, ring + rle_ring_wrap(origin + hp), 4);
            const uint32_t c0 = (x >> 7) & 1u, c1 = c0 & (x >> 15);
            const uint32_t v = (x & 0x7Fu) | (c0 ? (x >> 1) & 0x3F80u : 0u) | (c1 ? (x >> 2) & 0x1FC000u : 0u);
            const uint32_t hl = 1u + c0 + c1;
            const bool lit = (v & 1u) != 0;
            const uint32_t vals = real ? (lit ? (v >> 1) * 8u : (v >> 1)) : 0u;          // < 2^23
            uint32_t sc[2] = {vals, (vals + RLE_VPT - 1) / RLE_VPT};
            const uint32_t own_vals = sc[0], own_items = sc[1];
            wave_iscan_n(sc);
            // runs that end before the n-th value (a round has at most DRUNS runs of less than 2^23 values: no wrap)
            const uint64_t keep = __ballot(real && sum + sc[0] < left);
            const uint32_t m = (uint32_t)__popcll(keep);        // a prefix of the lanes: the sums do not decrease
            // runs without values are left out of the table
            const uint64_t has = __ballot(lane < m && vals != 0);
            if (lane < m && vals != 0) {
                const uint32_t k = nr + __builtin_amdgcn_mbcnt_hi((uint32_t)(has >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)has, 0u));
                const uint32_t it0 = items + sc[1] - own_items;
                sh.run_out[b][k] = out + sum + sc[0] - own_vals;
                sh.run_item[b][k] = it0;
                sh.run_info[b][k] = (uint16_t)(((hp + hl) << 1) | (lit ? 1u : 0u));
                if (it0 < (uint32_t)RLE_ITEM_CAP) atomicOr(reinterpret_cast<uint32_t *>(sh.item_bits[b]) + (it0 >> 5), 1u << (it0 & 31u));
            }
            if (m) {
                sum += bcast_u32(sc[0], (int)(m - 1));
                items += bcast_u32(sc[1], (int)(m - 1));
                nr += (uint32_t)__popcll(has);
            }
            if (k0 + m < cnt && m < 64) {                        // the run that holds the n-th value: not taken
                cut = true;
                pos = bcast_u32(hp, (int)m);
                break;
            }
        }
        use_bits = items <= (uint32_t)RLE_ITEM_CAP;
        RLE_WTICK(2);
        if (cut) stop = true;
        out += sum;
        cur = (uint64_t)(tile_off + (int64_t)pos);
    }
    if (lane == 0) {
        sh.run_out[b][nr] = out;
        sh.run_item[b][nr] = items;
        sh.nruns[b] = nr;
        sh.n_items[b] = items;
        RleRoundRec rr;
        rr.next_cur = cur; rr.out_end = out; rr.tile = t; rr.stop = stop ? 1u : 0u; rr.bits = use_bits ? 1u : 0u;
        sh.rnd[b] = rr;
    }
#undef RLE_WTICK
}

// The common case of passes 2 and 3 in one (one wave, uniform control flow): like rle_find_headers, every lane parses
// the header that would start at byte cur + lane and the true chain through the 64-byte span is followed with scalar
// steps, but the steps also add up the run lengths, so the run table is written here and no placing pass is needed.
// Only runs about which nothing can go wrong are taken: headers of one or two bytes, at least 80 stream bytes left at
// the start of the span, payload inside the stream, a value width that fits the output type.  At the first header
// that is anything else the round ends; when that is the round's first header nothing is written and the caller
// falls back to rle_find_headers + rle_place_runs, which decide every special case.
// Nor does it take the round in which the n-th value is reached.  `outp` (< n) = values produced before this round.
template <typename SH>
__device__ __forceinline__ bool rle_walk_fast(SH &sh, uint32_t b, const PageView &pv, int64_t wlo, uint64_t &cur,
                                              uint64_t outp, uint64_t n, uint32_t w, uint32_t tbytes, uint64_t in_limit) {
    const uint32_t lane = lane_id();
    const uint8_t *win = sh.win[b];
    const uint32_t vbytes = (w + 7) >> 3;
    if (vbytes > tbytes) return false;
    const uint64_t end_at = in_limit < pv.size ? in_limit : pv.size;
    if (cur >= end_at) return false;
    // (wave-uniform values, said so to the compiler: the loop below then runs on the scalar unit)
    const uint32_t room = first_u32(end_at - cur > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(end_at - cur));
    if (room < 80) return false;
    uint32_t rel = first_u32((uint32_t)((int64_t)cur - wlo));
    const uint32_t rel_first = rel;
    const uint32_t rel_end = rel + room;                                     // window offset of the end of the stream
    const uint32_t rel_stop = rel_end - 79 < (uint32_t)SH::DWIN - 64 ? rel_end - 79 : (uint32_t)SH::DWIN - 64;
    const uint32_t pay_lim = rel_end < (uint32_t)SH::DWIN ? rel_end : (uint32_t)SH::DWIN;
    const uint32_t left = first_u32(n - outp > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(n - outp));
    const uint32_t out_first = first_u32((uint32_t)outp);
    uint32_t out = out_first, items = 0, nr = 0;
    sh.item_bits[b][lane] = 0;
    while (rel < rel_stop && nr <= (uint32_t)SH::DRUNS - 64) {
        uint16_t h16;
        __builtin_memcpy(&h16, win + rel + lane, 2);
        const uint32_t h = h16;
        const bool two = (h & 0x80u) != 0;
        const uint32_t v = two ? (h & 0x7Fu) | ((h >> 1) & 0x3F80u) : (h & 0x7Fu);
        const uint32_t hl = two ? 2u : 1u, cnt = v >> 1;
        const bool lit = (h & 1u) != 0;
        const uint32_t skip = (lit ? cnt * w : vbytes) + hl;
        const uint32_t vals = lit ? cnt * 8 : cnt;
        const uint32_t it8 = (vals + RLE_VPT - 1) / RLE_VPT;
        const uint32_t pay_end = rel + lane + skip;
        const bool is3 = (h & 0x8080u) == 0x8080u;                             // three bytes or more
        // headers this pass does not decide, runs that the stream cannot hold, payloads that end past the window
        uint64_t stopmask = __ballot(is3 || pay_end > pay_lim);
        bool last = false;
        if (nr == 0) {
            // the round's first run is taken when only the window is too short for it (the expansion reads the rest
            // from global memory); it is then the round's only run
            if ((stopmask & 1ull) && !(__ballot(is3 || pay_end > rel_end) & 1ull)) { stopmask &= ~1ull; last = true; }
        }
        uint32_t r = 0, e_out = 0xFFFFFFFFu, e_item = 0;       // lane l: where the run whose header is at byte l starts
        for (;;) {
            if ((stopmask >> r) & 1ull) { last = true; break; }
            e_out = rle_writelane(out, r, e_out);
            e_item = rle_writelane(items, r, e_item);
            out += bcast_u32(vals, (int)r);
            items += bcast_u32(it8, (int)r);
            r += bcast_u32(skip, (int)r);
            if (r >= 64) break;
        }
        // (a run starts below n <= 2^32 - 1; runs without values are on the chain but not in the table: GetBatch skips
        // them, rle_encoding.hh:282-319, and the bit words count the runs that have items)
        const bool real = e_out != 0xFFFFFFFFu && vals > 0;
        const uint64_t M = __ballot(real);
        if (real) {
            const uint32_t k = nr + __builtin_amdgcn_mbcnt_hi((uint32_t)(M >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)M, 0u));
            sh.run_out[b][k] = e_out;
            sh.run_item[b][k] = e_item;
            sh.run_info[b][k] = (uint16_t)(((rel + lane + hl) << 1) | (lit ? 1u : 0u));
        }
        nr += (uint32_t)__popcll(M);
        rel += r;
        if (last) break;
    }
    // The round that reaches the n-th value is left to the passes that know how a decode ends.  (Run lengths are below
    // 2^16 here and a round has at most DRUNS runs: the sum cannot wrap.)
    if (nr == 0 || out - out_first >= left) return false;
    const bool fin = false;
    cur += rel - rel_first;
    if (lane == 0) {
        sh.run_out[b][nr] = out;
        sh.run_item[b][nr] = items;
        sh.nruns[b] = nr;
        sh.n_items[b] = items;
        sh.bits_ok[b] = items <= (uint32_t)RLE_ITEM_CAP ? 1u : 0u;
        sh.out_end = out;
        sh.finished = fin ? 1u : 0u;
        sh.stream_end = 0u;
        sh.next_cur = cur;
        sh.need_place = 0u;
    }
    return true;
}

// The walk of a round for streams with many headers in a 64-byte span (narrow values, short runs; linear windows),
// where following the chain header by header is the cost: every lane parses the header that would start at byte
// cur + lane and the lanes on the chain are found by pointer jumping.  Lane l holds the set of
// lanes reachable from l in fewer than 2^k steps and its 2^k-th successor; after RLE_DENSE_DOUBLINGS doublings the
// chain from the span's entry is the union of the sets of every 2^k-th lane on it, collected by readlane hops.  A lane ends the chain when its run's payload leaves the span (taken) or when this pass does not
// decide its header (not taken, the round ends there).  Output indices and item counts of the chain's runs are wave
// prefix sums.  Only runs about which nothing can go wrong are taken (headers of one or two bytes, payload inside the
// stream and the window, at least 80 stream bytes left); when the round's first header is anything else nothing is written and
// the caller falls back to rle_find_headers + rle_place_runs.  Nor does it take the round in which the n-th value is
// reached.  `outp` (< n) = values produced before this round.
// Doublings of the pointer jumping (each is a dependent trip through the LDS crossbar, about 300 cycles for the one
// walking wave): after d of them a lane knows the 2^d - 1 lanes behind it on its chain, and the chain from the span's
// entry is put together from every 2^d-th of its lanes by readlane hops (about 60 cycles each).  A level stream with
// 10 % nulls has some 24 headers in a span: three doublings and three hops instead of six doublings.
#ifndef P4S_RLE_DENSE_DOUBLINGS
#define P4S_RLE_DENSE_DOUBLINGS 3
#endif
constexpr int RLE_DENSE_DOUBLINGS = P4S_RLE_DENSE_DOUBLINGS;
#ifndef P4S_RLE_DENSE_GROUP
#define P4S_RLE_DENSE_GROUP 1
#endif
template <typename SH, int G = P4S_RLE_DENSE_GROUP>
__device__ __forceinline__ bool rle_walk_dense(SH &sh, uint32_t b, const PageView &pv, int64_t wlo, uint64_t &cur,
                                               uint64_t outp, uint64_t n, uint32_t w, uint32_t tbytes, uint64_t in_limit) {
    const uint32_t lane = lane_id();
    const uint8_t *win = sh.win[b];
    const uint32_t vbytes = (w + 7) >> 3;
    if (vbytes > tbytes) return false;
    const uint64_t end_at = in_limit < pv.size ? in_limit : pv.size;
    if (cur >= end_at) return false;
    const uint32_t room = first_u32(end_at - cur > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(end_at - cur));
    if (room < 80) return false;
    uint32_t rel = first_u32((uint32_t)((int64_t)cur - wlo));
    const uint32_t rel_first = rel;
    const uint32_t rel_end = rel + room;
    const uint32_t rel_stop = rel_end - 79 < (uint32_t)SH::DWIN - 64 ? rel_end - 79 : (uint32_t)SH::DWIN - 64;
    const uint32_t pay_lim = rel_end < (uint32_t)SH::DWIN ? rel_end : (uint32_t)SH::DWIN;
    const uint32_t left = first_u32(n - outp > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)(n - outp));
    const uint32_t out_first = first_u32((uint32_t)outp);
    uint32_t out = out_first, items = 0, nr = 0;
    sh.item_bits[b][lane] = 0;
    // (a round is also ended by its item count, so that the bit words can describe it; a single span that brings more
    // than the other half of RLE_ITEM_CAP - a repeated run of 16 Ki values or more - leaves the round to the search)
    while (rel < rel_stop && nr <= (uint32_t)SH::DRUNS - 64 && items < (uint32_t)RLE_ITEM_CAP / 2) {
        // G spans of 64 bytes at a time, at rel + 64 s: their pointer jumping does not depend on where the chain enters
        // them, so the G chains of cross-lane reads run side by side and share their waits; where the chain enters
        // span s + 1 is then one readlane away from span s
#ifdef P4S_EXP_TIMING
        if (lane == 0) atomicAdd(&sh.cnt_iter, 1u);
#endif
        uint32_t skip[G], vals[G], info[G];
        uint64_t stopmask[G], M[G];
        bool first_too_long = false, first_undecided = false;
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const uint32_t at = rel + 64u * s + lane;
            uint16_t h16;
            __builtin_memcpy(&h16, win + at, 2);             // (spans past the window read other LDS; they are not entered)
            const uint32_t h = h16;
            const bool two = (h & 0x80u) != 0;
            const uint32_t v = two ? (h & 0x7Fu) | ((h >> 1) & 0x3F80u) : (h & 0x7Fu);
            const uint32_t hl = two ? 2u : 1u, cnt = v >> 1;
            const bool lit = (h & 1u) != 0;
            skip[s] = (lit ? cnt * w : vbytes) + hl;
            vals[s] = lit ? cnt * 8 : cnt;
            info[s] = ((at + hl) << 1) | (lit ? 1u : 0u);
            const uint32_t pay_end = at + skip[s];
            const bool is3 = (h & 0x8080u) == 0x8080u;
            stopmask[s] = __ballot(is3 || pay_end > pay_lim);
            M[s] = 0;
            if (s == 0) {
                first_too_long = nr == 0 && (stopmask[0] & 1ull);
                first_undecided = (__ballot(is3 || pay_end > rel_end) & 1ull) != 0;
            }
        }
        uint32_t r;
        bool last = false;
        if (first_too_long) {
            // the round's first run, too long for the window alone: taken as the round's only run; anything else
            // about it: the other passes decide
            if (first_undecided) break;
            M[0] = 1ull; r = bcast_u32(skip[0], 0); last = true;
        } else {
            // successor on the chain; a lane that ends the chain points at itself
            uint32_t j[G], rlo[G], rhi[G];
#pragma unroll
            for (int s = 0; s < G; ++s) {
                const bool term = ((stopmask[s] >> lane) & 1ull) || lane + skip[s] >= 64u;
                j[s] = term ? lane : lane + skip[s];
                rlo[s] = lane < 32 ? 1u << lane : 0u;
                rhi[s] = lane < 32 ? 0u : 1u << (lane - 32);
            }
#pragma unroll
            for (int k = 0; k < RLE_DENSE_DOUBLINGS; ++k) {
                uint32_t nlo[G], nhi[G], nj[G];
#pragma unroll
                for (int s = 0; s < G; ++s) {
                    const int idx = (int)(j[s] << 2);
                    nlo[s] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)rlo[s]);
                    nhi[s] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)rhi[s]);
                    nj[s] = (uint32_t)__builtin_amdgcn_ds_bpermute(idx, (int)j[s]);
                }
#pragma unroll
                for (int s = 0; s < G; ++s) { rlo[s] |= nlo[s]; rhi[s] |= nhi[s]; j[s] = nj[s]; }
            }
            // through the spans: x = offset from rel of the next header
            uint32_t x = 0, nr_most = nr;
            bool open = true;
#pragma unroll
            for (int s = 0; s < G; ++s) {
                if (open && x < 64u * (s + 1)) {
                    if (rel + x >= rel_stop || nr_most > (uint32_t)SH::DRUNS - 64) {
                        open = false;                        // the next iteration, or the next round, goes on from x
                    } else {
                        // the set of the lane the chain enters at, and of every 2^DOUBLINGS-th lane after it on the
                        // chain (a step of this loop is three readlanes; a doubling is a trip through the LDS crossbar)
                        uint32_t e = x - 64u * s;
                        uint64_t chain = 0;
                        for (;;) {
                            chain |= bcast_u64(((uint64_t)rhi[s] << 32) | rlo[s], (int)e);
                            const uint32_t nx = bcast_u32(j[s], (int)e);
                            if ((chain >> nx) & 1ull) break;           // (a lane that ends the chain points at itself)
                            e = nx;
                        }
                        const uint64_t hit = chain & stopmask[s];       // at most the chain's last lane
                        M[s] = chain & ~stopmask[s];
                        nr_most += (uint32_t)__popcll(M[s]);
                        if (hit) {
                            x = 64u * s + (uint32_t)__builtin_ctzll(hit);
                            last = true;
                            open = false;
                        } else {
                            const uint32_t t = 63u - (uint32_t)__builtin_clzll(chain);
                            x = 64u * s + t + bcast_u32(skip[s], (int)t);
                        }
                    }
                }
            }
            r = x;
        }
        // runs without values are on the chain but not in the table (GetBatch skips them, rle_encoding.hh:282-319)
        uint32_t sc[2 * G], own[2 * G];
        uint64_t Mr[G];
#pragma unroll
        for (int s = 0; s < G; ++s) {
            const bool real = ((M[s] >> lane) & 1ull) && vals[s] > 0;
            Mr[s] = __ballot(real);
            own[2 * s] = sc[2 * s] = real ? vals[s] : 0u;
            own[2 * s + 1] = sc[2 * s + 1] = real ? (vals[s] + RLE_VPT - 1) / RLE_VPT : 0u;
        }
        wave_iscan_n(sc);
#pragma unroll
        for (int s = 0; s < G; ++s) {
            if ((Mr[s] >> lane) & 1ull) {
                const uint32_t k = nr + __builtin_amdgcn_mbcnt_hi((uint32_t)(Mr[s] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)Mr[s], 0u));
                sh.run_out[b][k] = out + sc[2 * s] - own[2 * s];
                sh.run_item[b][k] = items + sc[2 * s + 1] - own[2 * s + 1];
                sh.run_info[b][k] = (uint16_t)info[s];
            }
            out += bcast_u32(sc[2 * s], 63);
            items += bcast_u32(sc[2 * s + 1], 63);
            nr += (uint32_t)__popcll(Mr[s]);
        }
        rel += r;
        if (last) break;
    }
    // the round that reaches the n-th value is left to the passes that know how a decode ends
    if (nr == 0 || out - out_first >= left) return false;
    cur += rel - rel_first;
    if (lane == 0) {
        sh.run_out[b][nr] = out;
        sh.run_item[b][nr] = items;
        sh.nruns[b] = nr;
        sh.n_items[b] = items;
        sh.bits_ok[b] = items <= (uint32_t)RLE_ITEM_CAP ? 1u : 0u;
        sh.out_end = out;
        sh.finished = 0u;
        sh.stream_end = 0u;
        sh.next_cur = cur;
        sh.need_place = 0u;
    }
    return true;
}

// Values the reference's caller still gets out of a literal run that the stream cannot fully supply, when it
// drains the decoder in `chunk`-sized batches (dict_decoder: 1000, src/encoding.cc:483-506): whole leading
// batches succeed, the first short one returns nothing (rle_encoding.hh:299-304).
__device__ __forceinline__ uint64_t rle_partial(uint64_t start, uint64_t want, uint64_t avail_vals, uint32_t chunk) {
    if (!chunk) return 0;
    uint64_t pos = start, got = 0;
    while (got < want) {
        uint64_t piece = chunk - (pos % chunk);
        if (piece > want - got) piece = want - got;
        if (got + piece > avail_vals) break;
        got += piece;
        pos += piece;
    }
    return got;
}

// Pass 3 of a round (all threads): output index of every run, the run that ends the decode, the round's totals.
// `outp` = values produced before this round; returns the values produced after it.
template <typename SH>
__device__ __forceinline__ uint64_t rle_place_runs(SH &sh, uint32_t b, const PageView &pv, int64_t wlo, uint64_t outp, uint64_t n,
                                                   uint32_t w, uint32_t tbytes, uint32_t chunk) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win[b]);
    const uint32_t vbytes = (w + 7) >> 3;
    const uint32_t nh = sh.nhdr;
    const uint32_t k0 = tid * SH::RPT;
    uint64_t vals[SH::RPT];
    uint32_t info[SH::RPT];
    uint32_t room[SH::RPT];          // payload bytes the stream still holds after the header
    bool fail[SH::RPT];
    uint64_t local = 0;
#pragma unroll
    for (int j = 0; j < SH::RPT; ++j) {
        vals[j] = 0; info[j] = 0; room[j] = 0; fail[j] = false;
        if (k0 + j < nh) {
            const uint32_t rel = reinterpret_cast<const uint16_t *>(sh.run_item[b])[k0 + j];
            const int32_t avail = rle_avail(pv, wlo + (int64_t)rel);
            const RleHeader h = rle_parse(rle_win40(win32, rel), avail, w, vbytes, tbytes);
            fail[j] = h.fail;
            room[j] = h.fail ? 0u : (uint32_t)(avail - (int32_t)h.hl);
            info[j] = ((rel + h.hl) << 1) | (h.lit ? 1u : 0u);
            vals[j] = h.fail ? 0 : (h.lit ? (uint64_t)h.count * 8 : (uint64_t)h.count);
        }
        local += vals[j];
    }
    const uint64_t inc = wave_iscan(local);
    if (lane == 63) sh.wtot[wave] = inc;
    if (tid == 0) sh.kstop = SH::DRUNS + 1;
    __syncthreads();
    uint64_t start = outp + (inc - local), total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t t = sh.wtot[i];
        if ((uint32_t)i < wave) start += t;
        total += t;
    }
    // the one run of this thread that may turn out to end the decode
    uint32_t stop_k = 0xFFFFFFFFu, stop_room = 0;
    uint64_t stop_start = 0, stop_vals = 0;
    bool stop_trunc = false;
    // Expansion items (groups of 8 values that lie in one run) of the runs that are wanted.  Sums are taken modulo
    // 2^32: they are exact for every run up to the one that ends the decode, and no run past it is expanded.
    uint32_t items[SH::RPT], items_local = 0;
#pragma unroll
    for (int j = 0; j < SH::RPT; ++j) {
        items[j] = 0;
        if (k0 + j < nh) {
            const bool needed = !fail[j] && start < n;
            if (needed) {
                const uint64_t take = vals[j] < n - start ? vals[j] : n - start;
                items[j] = (uint32_t)((take + RLE_VPT - 1) / RLE_VPT);
            }
            items_local += items[j];
            bool trunc = false;
            if (needed && (info[j] & 1u) && w > 0) {
                // a literal run that cannot supply what is wanted of it ends the decode (rle_encoding.hh:299-304)
                const uint64_t want = vals[j] < n - start ? vals[j] : n - start;
                trunc = want > ((uint64_t)room[j] * 8) / w;
            }
            if ((!needed || trunc) && stop_k == 0xFFFFFFFFu) {
                stop_k = k0 + j; stop_start = start; stop_vals = vals[j]; stop_room = room[j]; stop_trunc = trunc;
            }
            sh.run_out[b][k0 + j] = (uint32_t)(start < n ? start : n);
            sh.run_info[b][k0 + j] = (uint16_t)info[j];
        }
        start += vals[j];
    }
    if (stop_k != 0xFFFFFFFFu) atomicMin(&sh.kstop, stop_k);
    if (tid == 0) {
        const uint64_t end = outp + total;
        sh.run_out[b][nh] = (uint32_t)(end < n ? end : n);
    }
    const uint32_t items_inc = wave_iscan(items_local);
    if (lane == 63) sh.wtot_items[wave] = items_inc;
    __syncthreads();
    {
        uint32_t it = items_inc - items_local;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if ((uint32_t)i < wave) it += sh.wtot_items[i];
#pragma unroll
        for (int j = 0; j < SH::RPT; ++j) {
            if (k0 + j < nh) sh.run_item[b][k0 + j] = it;
            it += items[j];
        }
    }
    const uint32_t ks = sh.kstop;
    const uint32_t K = ks < nh ? ks : nh;
    if (K == nh) {
        if (tid == 0) {
            const uint32_t oe = sh.run_out[b][nh];
            sh.nruns[b] = nh;
            sh.out_end = oe;
            sh.finished = (sh.stream_end || (uint64_t)oe >= n) ? 1u : 0u;
        }
    } else if (K == stop_k) {
        // this thread owns the run that ends the decode
        if (stop_start < n) sh.irregular = 1u;
        uint32_t nruns = K;
        uint64_t oe = stop_start < n ? stop_start : n;
        if (stop_trunc) {
            const uint64_t want = stop_vals < n - stop_start ? stop_vals : n - stop_start;
            const uint64_t got = rle_partial(stop_start, want, ((uint64_t)stop_room * 8) / w, chunk);
            if (got) {
                nruns = K + 1;
                oe = stop_start + got;
                sh.run_out[b][K + 1] = (uint32_t)oe;
            }
        }
        sh.nruns[b] = nruns;
        sh.out_end = (uint32_t)oe;
        sh.finished = 1u;
    }
    __syncthreads();
    if (tid == 0) {
        // the last run may have been cut short (count reached, truncated literal run): its items follow its placed length
        const uint32_t nr = sh.nruns[b];
        uint32_t total_items = 0;
        if (nr) total_items = sh.run_item[b][nr - 1] + (sh.run_out[b][nr] - sh.run_out[b][nr - 1] + RLE_VPT - 1) / RLE_VPT;
        sh.run_item[b][nr] = total_items;
        sh.n_items[b] = total_items;
    }
    return sh.out_end;
}

// w bits at bit position `bit` of the stream, read from global memory with aligned dword loads
// (the dword holding the first bit and the next one always cover a field of up to 32 bits).
__device__ __forceinline__ uint32_t rle_field_global(const PageView &pv, uint64_t bit, uint32_t w) {
    const uint64_t byte = bit >> 3;
    const uintptr_t a = (uintptr_t)(pv.base + byte);
    const uint32_t mis = (uint32_t)(a & 3u);
    const uint32_t sh = mis * 8 + (uint32_t)(bit & 7u);
    const int64_t room = (int64_t)pv.room - (int64_t)byte + mis;      // readable bytes from the aligned address
    uint32_t d0 = 0, d1 = 0;
    if (room >= 8) {
        const uint32_t *q = reinterpret_cast<const uint32_t *>(a - mis);   // d_in is 16-byte aligned: still inside the allocation
        d0 = P4S_LD(q, 50);
        d1 = P4S_LD(q + 1, 51);
    } else {
        const uint8_t *q = reinterpret_cast<const uint8_t *>(a - mis);
        for (int i = 0; i < 4; ++i) {
            if (i < room) d0 |= (uint32_t)P4S_LD(q + i, 52) << (8 * i);
            if (i + 4 < room) d1 |= (uint32_t)P4S_LD(q + i + 4, 53) << (8 * i);
        }
    }
    return mask32(__funnelshift_r(d0, d1, sh), w);
}

// Output-side policy: what happens to a decoded hybrid value.  A lane's group is RLE_VPT consecutive values of one run
// starting at output index v0 (any alignment); `first` .. `cnt` - 1 are the ones to store.
template <typename OutT>
struct RleSinkPlain {
    static constexpr bool COUNT_ONLY = false;
    OutT *out;
    // values x[0 .. cnt) (cnt <= 4) for output indices pos .. pos + cnt - 1: consecutive lanes hold consecutive groups
    template <typename SH>
    __device__ __forceinline__ void store4(uint64_t pos, uint32_t cnt, const uint32_t x[4], SH &) const {
        OutT tmp[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) tmp[j] = (OutT)x[j];
        if (cnt == 4) {
            if (P4S_INB(out + pos, sizeof(tmp), 54)) __builtin_memcpy(out + pos, tmp, sizeof(tmp));
        } else {
            // (three at most here: with a fourth in this loop the compiler shares its store with the branch above,
            // whose 16 bytes then leave as 12 + 4 in two store instructions)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if ((uint32_t)j < cnt) P4S_ST(out + pos + j, tmp[j], 55);
        }
    }
    template <typename SH>
    __device__ __forceinline__ void store_vals(uint64_t v0, uint32_t first, uint32_t cnt, const uint32_t x[RLE_VPT], SH &) const {
        OutT tmp[RLE_VPT];
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) tmp[j] = (OutT)x[j];
        if (first == 0 && cnt == RLE_VPT) {
            if (P4S_INB(out + v0, sizeof(tmp), 54)) __builtin_memcpy(out + v0, tmp, sizeof(tmp));
        } else {
#pragma unroll
            for (int j = 0; j < RLE_VPT; ++j)
                if ((uint32_t)j >= first && (uint32_t)j < cnt) P4S_ST(out + v0 + j, tmp[j], 55);
        }
    }
};

template <int ELEM>
struct RleSinkDict {
    static constexpr bool COUNT_ONLY = false;
    uint8_t *out;
    const uint8_t *dict;
    uint32_t dict_size;       // indices above it are errors (sizes of 2^32 and more: no 32-bit index is)
    // values x[0 .. cnt) (cnt <= 4) for output indices pos .. pos + cnt - 1: consecutive lanes hold consecutive groups
    template <typename SH>
    __device__ __forceinline__ void store4(uint64_t pos, uint32_t cnt, const uint32_t x[4], SH &sh) const {
        if constexpr (ELEM == 4) {
            if (cnt == 4) {
                const uint32_t *d = reinterpret_cast<const uint32_t *>(dict);
                bool bad = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) bad |= x[j] > dict_size;
                if (!bad) {
                    uint32_t g[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[j] = P4S_LD(d + x[j], 56);
                    uint32_t *o = reinterpret_cast<uint32_t *>(out) + pos;
                    if (P4S_INB(o, 16, 57)) __builtin_memcpy(o, g, 16);
                    return;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((uint32_t)j >= cnt) continue;
            const uint32_t idx = x[j];
            if (idx > dict_size) {            // encoding.cc:489 (idx == dict_size passes)
                atomicMin(&sh.errkey, ((unsigned long long)(pos + j) << 32) | idx);
                continue;
            }
            if constexpr (ELEM == 1) {
                P4S_ST(out + pos + j, P4S_LD(dict + idx, 58), 59);
            } else if constexpr (ELEM == 4) {
                P4S_ST(reinterpret_cast<uint32_t *>(out) + pos + j, P4S_LD(reinterpret_cast<const uint32_t *>(dict) + idx, 60), 61);
            } else if constexpr (ELEM == 8) {
                P4S_ST(reinterpret_cast<uint64_t *>(out) + pos + j, P4S_LD(reinterpret_cast<const uint64_t *>(dict) + idx, 62), 63);
            } else {
                const uint32_t *sp = reinterpret_cast<const uint32_t *>(dict) + (uint64_t)idx * 3;
                uint32_t *dp = reinterpret_cast<uint32_t *>(out) + (pos + j) * 3;
                if (P4S_INB(sp, 12, 64) && P4S_INB(dp, 12, 65)) { dp[0] = sp[0]; dp[1] = sp[1]; dp[2] = sp[2]; }
            }
        }
    }
    template <typename SH>
    __device__ __forceinline__ void store_vals(uint64_t v0, uint32_t first, uint32_t cnt, const uint32_t x[RLE_VPT], SH &sh) const {
        if constexpr (ELEM == 4) {
            if (first == 0 && cnt == RLE_VPT) {
                const uint32_t *d = reinterpret_cast<const uint32_t *>(dict);
                bool bad = false;
#pragma unroll
                for (int j = 0; j < RLE_VPT; ++j) bad |= x[j] > dict_size;
                if (!bad) {
                    uint32_t g[RLE_VPT];
#pragma unroll
                    for (int j = 0; j < RLE_VPT; ++j) g[j] = P4S_LD(d + x[j], 56);
                    uint32_t *o = reinterpret_cast<uint32_t *>(out) + v0;
                    if (P4S_INB(o, 4 * RLE_VPT, 57)) __builtin_memcpy(o, g, 4 * RLE_VPT);
                    return;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) {
            if ((uint32_t)j < first || (uint32_t)j >= cnt) continue;
            const uint32_t idx = x[j];
            if (idx > dict_size) {            // encoding.cc:489 (idx == dict_size passes)
                atomicMin(&sh.errkey, ((unsigned long long)(v0 + j) << 32) | idx);
                continue;
            }
            if constexpr (ELEM == 1) {
                P4S_ST(out + v0 + j, P4S_LD(dict + idx, 58), 59);
            } else if constexpr (ELEM == 4) {
                P4S_ST(reinterpret_cast<uint32_t *>(out) + v0 + j, P4S_LD(reinterpret_cast<const uint32_t *>(dict) + idx, 60), 61);
            } else if constexpr (ELEM == 8) {
                P4S_ST(reinterpret_cast<uint64_t *>(out) + v0 + j, P4S_LD(reinterpret_cast<const uint64_t *>(dict) + idx, 62), 63);
            } else {
                const uint32_t *s = reinterpret_cast<const uint32_t *>(dict) + (uint64_t)idx * 3;
                uint32_t *d = reinterpret_cast<uint32_t *>(out) + (v0 + j) * 3;
                if (P4S_INB(s, 12, 64) && P4S_INB(d, 12, 65)) { d[0] = s[0]; d[1] = s[1]; d[2] = s[2]; }
            }
        }
    }
};

// Eight W-bit values packed LSB first from byte `byte` of the LDS window (a literal run's groups of eight start on
// byte boundaries): ceil(W / 4) dwords read at that byte, then fields at compile-time positions.
template <int W>
__device__ __forceinline__ void rle_unpack8(const uint32_t *win32, uint32_t byte, uint32_t x[RLE_VPT]) {
    if constexpr (W == 0) {
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) x[j] = 0;
    } else {
        constexpr int NA = (W + 3) / 4;
        uint32_t d[NA];
        __builtin_memcpy(d, reinterpret_cast<const uint8_t *>(win32) + byte, 4 * NA);      // LDS reads need no alignment
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) {
            constexpr uint32_t M = W == 32 ? 0xFFFFFFFFu : ((1u << (W & 31)) - 1u);
            const int i = (j * W) >> 5, sh = (j * W) & 31;
            if (sh + W <= 32) x[j] = (d[i] >> sh) & M;
            else x[j] = __funnelshift_r(d[i], d[i + 1 < NA ? i + 1 : i], sh) & M;
        }
    }
}

__device__ __forceinline__ void rle_unpack8_w(uint32_t w, const uint32_t *win32, uint32_t byte, uint32_t x[RLE_VPT]) {
    switch (w) {
#define P4S_RLE_CASE(W) case W: rle_unpack8<W>(win32, byte, x); break;
        P4S_RLE_CASE(0) P4S_RLE_CASE(1) P4S_RLE_CASE(2) P4S_RLE_CASE(3) P4S_RLE_CASE(4) P4S_RLE_CASE(5) P4S_RLE_CASE(6)
        P4S_RLE_CASE(7) P4S_RLE_CASE(8) P4S_RLE_CASE(9) P4S_RLE_CASE(10) P4S_RLE_CASE(11) P4S_RLE_CASE(12) P4S_RLE_CASE(13)
        P4S_RLE_CASE(14) P4S_RLE_CASE(15) P4S_RLE_CASE(16) P4S_RLE_CASE(17) P4S_RLE_CASE(18) P4S_RLE_CASE(19) P4S_RLE_CASE(20)
        P4S_RLE_CASE(21) P4S_RLE_CASE(22) P4S_RLE_CASE(23) P4S_RLE_CASE(24) P4S_RLE_CASE(25) P4S_RLE_CASE(26) P4S_RLE_CASE(27)
        P4S_RLE_CASE(28) P4S_RLE_CASE(29) P4S_RLE_CASE(30) P4S_RLE_CASE(31)
        default: rle_unpack8<32>(win32, byte, x); break;
#undef P4S_RLE_CASE
    }
}

// Where the bytes of a round's window are: `base` = the LDS memory, `origin` = offset in it of the window's byte 0
// (a ring slot, or 0), `lim` = bytes of the window that LDS holds, counted from byte 0 (what lies further is read
// from global memory), `wlo` = stream offset of byte 0.
struct RleWinRef {
    const uint8_t *base;
    uint32_t origin;
    uint32_t lim;
    int64_t wlo;
    const uint64_t *item_bits;      // which items start a run (RleShared), or null: an item's run is then searched in the table
    uint32_t *tile_ctr;             // the round's next unclaimed tile of 64 items (zero before the round's expansion begins)
};

// Pass 4 of a round: values [ob, oe) from run table `b` and window `wr`.  The work is cut into items: 8 consecutive values of
// ONE run, counted from the run's first value, so that a literal item starts on a byte boundary of the stream and never
// straddles runs.  Called by whole waves; a wave claims tiles of 64 items from sh.tile_next (reset before the barrier
// that precedes the round's expansion) until none is left, and each lane finds its item's run by binary search.
template <typename Sink, typename SH>
__device__ __forceinline__ void rle_expand_round(SH &sh, uint32_t b, const PageView &pv, const RleWinRef wr, uint64_t ob, uint64_t oe,
                                                 uint32_t w, const Sink &sink, bool count_levels, uint32_t max_level,
                                                 uint32_t &hits, uint32_t &above) {
    const uint32_t lane = lane_id();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(wr.base);
    const int64_t wlo = wr.wlo;
    const uint32_t *run_out = sh.run_out[b];
    const uint32_t *run_item = sh.run_item[b];
    const uint16_t *run_info = sh.run_info[b];
    const uint32_t nruns = sh.nruns[b];
    const uint32_t n_items = sh.n_items[b];
    const uint32_t vbytes = (w + 7) >> 3;
    const uint32_t vmask = vbytes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * vbytes)) - 1u);
    uint32_t lo_hint = 0;
    // with the bit words of the round (wr.item_bits: which items start a run; the table then holds no run without
    // items, and the round no more than RLE_ITEM_CAP items): lane j keeps the word of tile j and the runs before it
    const bool by_bits = wr.item_bits != nullptr;
    uint64_t my_bits = 0;
    uint32_t my_rank = 0;
    if (by_bits) {
        my_bits = wr.item_bits[lane];
        const uint32_t pc = (uint32_t)__popcll(my_bits);
        my_rank = wave_iscan(pc) - pc;
    }
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(wr.tile_ctr, 1u);
        t = first_u32(t);
        if ((uint64_t)t * 64 >= n_items) break;
        const uint32_t item = t * 64 + lane;
        uint32_t lo;
        if (by_bits) {
            const uint64_t bits = bcast_u64(my_bits, (int)t);
            lo = bcast_u32(my_rank, (int)t) + __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u)) +
                 (uint32_t)((bits >> lane) & 1ull) - 1u;
            if (item >= n_items) continue;
        } else {
            if (item >= n_items) continue;
            // run of the item: last r with run_item[r] <= item (a run without items shares its successor's index: the successor wins)
            lo = lo_hint;
            uint32_t hi = nruns;                     // invariant: run_item[lo] <= item < run_item[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (run_item[mid] <= item) lo = mid; else hi = mid;
            }
            lo_hint = lo;
        }
        const uint32_t g = item - run_item[lo];
        const uint64_t r_end = run_out[lo + 1];
        const uint64_t v0 = (uint64_t)run_out[lo] + (uint64_t)g * RLE_VPT;
        if (v0 >= r_end || v0 >= oe) continue;
        const uint32_t cnt = r_end - v0 < RLE_VPT ? (uint32_t)(r_end - v0) : (uint32_t)RLE_VPT;
        const uint32_t first = v0 < ob ? (ob - v0 < RLE_VPT ? (uint32_t)(ob - v0) : (uint32_t)RLE_VPT) : 0u;
        if (first >= cnt) continue;
        const uint32_t info = run_info[lo];
        const uint32_t rel = info >> 1;
        uint32_t x[RLE_VPT];
        if (info & 1u) {
            const uint64_t byte = (uint64_t)rel + (uint64_t)g * w;       // window offset of the item's w bytes
            if (byte + w <= (uint64_t)wr.lim) {
                uint32_t at = wr.origin + (uint32_t)byte;
                if constexpr (SH::RINGSIZE != 0) at = at >= (uint32_t)SH::RINGSIZE ? at - (uint32_t)SH::RINGSIZE : at;
                rle_unpack8_w(w, win32, at, x);
            } else {
                // the part of a round's first run that lies past the window
                const uint64_t bit0 = ((uint64_t)(wlo + (int64_t)rel) + (uint64_t)g * w) * 8;
#pragma unroll
                for (int j = 0; j < RLE_VPT; ++j)
                    x[j] = ((uint32_t)j >= first && (uint32_t)j < cnt) ? rle_field_global(pv, bit0 + (uint64_t)j * w, w) : 0u;
            }
        } else {
            // ceil(w / 8) raw bytes, not masked to w (bit_stream_utils.hh:368-393)
            uint32_t rep_val = 0;
            if (vbytes) {
                if (rel + vbytes <= wr.lim) {
                    uint32_t at = wr.origin + rel;
                    if constexpr (SH::RINGSIZE != 0) at = at >= (uint32_t)SH::RINGSIZE ? at - (uint32_t)SH::RINGSIZE : at;
                    rep_val = (uint32_t)rle_win40(win32, at) & vmask;
                } else {
                    rep_val = rle_field_global(pv, (uint64_t)(wlo + (int64_t)rel) * 8, 8 * vbytes);
                }
            }
#pragma unroll
            for (int j = 0; j < RLE_VPT; ++j) x[j] = rep_val;
        }
        if (count_levels) {
#pragma unroll
            for (int j = 0; j < RLE_VPT; ++j) {
                if ((uint32_t)j >= first && (uint32_t)j < cnt) {
                    if (x[j] == max_level) ++hits;
                    else if (x[j] > max_level) ++above;
                }
            }
        }
        sink.store_vals(v0, first, cnt, x, sh);
    }
}

// Pass 4 of a ring round whose items are described by item_bits (RleShared).  The same items as in
// rle_expand_round, but: an item's run comes from a population count instead of a search; the next tile is claimed,
// and its bit word read, while the current one is looked up and unpacked; output indices are 32-bit (a page has fewer
// than 2^32 values); and the values do not leave from the lane that unpacked them.  The items of a tile follow each
// other in the output without a gap, so the wave puts them side by side in its staging area and then stores the
// range in groups of four values per lane, consecutive lanes consecutive groups: every store instruction writes
// whole cache lines.
template <typename Sink>
__device__ __forceinline__ void rle_expand_ring(RleShared &sh, uint32_t b, const PageView &pv, const RleWinRef wr, uint32_t ob, uint32_t oe,
                                                uint32_t w, const Sink &sink, bool count_levels, uint32_t max_level,
                                                uint32_t &hits, uint32_t &above) {
    const uint32_t lane = lane_id();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(wr.base);
    const uint32_t *run_out = sh.run_out[b];
    const uint32_t *run_item = sh.run_item[b];
    const uint16_t *run_info = sh.run_info[b];
    uint32_t *stg = sh.stage[first_u32(threadIdx.x >> 6)];
    const uint32_t n_items = sh.n_items[b];
    const uint32_t ntiles = (n_items + 63u) >> 6;
    const uint32_t vbytes = (w + 7) >> 3;
    const uint32_t vmask = vbytes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * vbytes)) - 1u);
    uint32_t tv = 0;
    if (lane == 0) tv = atomicAdd(wr.tile_ctr, 1u);
    // lane j: the bit word of tile j and the runs that start before it
    const uint64_t my_bits = wr.item_bits[lane];
    const uint32_t my_pc = (uint32_t)__popcll(my_bits);
    const uint32_t my_rank = wave_iscan(my_pc) - my_pc;
    uint32_t t = first_u32(tv);
    if (t >= ntiles) return;
    for (;;) {
        const uint64_t bits = bcast_u64(my_bits, (int)t);
        const uint32_t rank = bcast_u32(my_rank, (int)t);
        uint32_t tnv = 0;
        if (lane == 0) tnv = atomicAdd(wr.tile_ctr, 1u);          // the tile after this one
        const uint32_t item = t * 64u + lane;
        // runs that start at or before the item, minus one
        const uint32_t lo = rank + __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u)) +
                            (uint32_t)((bits >> lane) & 1ull) - 1u;
        const uint32_t ri = run_item[lo], ro = run_out[lo], re = run_out[lo + 1];
        const uint32_t info = run_info[lo];
        const uint32_t tn = first_u32(tnv);
        const uint32_t g = item - ri;
        const uint32_t v0 = ro + g * RLE_VPT;
        const uint32_t cnt = re - v0 < RLE_VPT ? re - v0 : (uint32_t)RLE_VPT;
        const uint32_t first = v0 < ob ? (ob - v0 < RLE_VPT ? ob - v0 : (uint32_t)RLE_VPT) : 0u;
        const bool act = item < n_items && v0 < re && v0 < oe && first < cnt;
        const uint64_t am = __ballot(act);
        if (am) {
            // the tile's part of the output: from the first value of its first item to the last of its last
            const uint32_t p0 = bcast_u32(v0 + first, (int)__builtin_ctzll(am));
            const uint32_t pe = bcast_u32(v0 + cnt, 63 - (int)__builtin_clzll(am));
            if (act) {
                const uint32_t rel = info >> 1;
                uint32_t x[RLE_VPT];
                if (info & 1u) {
                    const uint32_t byte = rel + g * w;                       // window offset of the item's w bytes (< 2^16 * 32)
                    if (byte + w <= wr.lim) {
                        uint32_t at = wr.origin + byte;
                        at = at >= (uint32_t)RLE_RING ? at - (uint32_t)RLE_RING : at;
                        rle_unpack8_w(w, win32, at, x);
                    } else {
                        // the part of a run that lies past the tiles in LDS
                        const uint64_t bit0 = ((uint64_t)(wr.wlo + (int64_t)rel) + (uint64_t)g * w) * 8;
#pragma unroll
                        for (int j = 0; j < RLE_VPT; ++j)
                            x[j] = ((uint32_t)j >= first && (uint32_t)j < cnt) ? rle_field_global(pv, bit0 + (uint64_t)j * w, w) : 0u;
                    }
                } else {
                    // ceil(w / 8) raw bytes, not masked to w (bit_stream_utils.hh:368-393)
                    uint32_t rep_val = 0;
                    if (vbytes) {
                        if (rel + vbytes <= wr.lim) {
                            uint32_t at = wr.origin + rel;
                            at = at >= (uint32_t)RLE_RING ? at - (uint32_t)RLE_RING : at;
                            rep_val = (uint32_t)rle_win40(win32, at) & vmask;
                        } else {
                            rep_val = rle_field_global(pv, (uint64_t)(wr.wlo + (int64_t)rel) * 8, 8 * vbytes);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < RLE_VPT; ++j) x[j] = rep_val;
                }
                if (count_levels) {
#pragma unroll
                    for (int j = 0; j < RLE_VPT; ++j) {
                        if ((uint32_t)j >= first && (uint32_t)j < cnt) {
                            if (x[j] == max_level) ++hits;
                            else if (x[j] > max_level) ++above;
                        }
                    }
                }
                uint32_t *dst = stg + (int32_t)(v0 - p0);          // (v0 + first >= p0; with first > 0 the slots below p0 are not written)
                if (first == 0 && cnt == RLE_VPT) {
                    __builtin_memcpy(dst, x, 4 * RLE_VPT);
                } else if (!(info & 1u)) {
                    // part of a repeated run (where nearly all items that are not whole come from: a literal run is
                    // whole groups of eight): 4, 2, 1 equal values by the bits of the length
                    const uint32_t len = cnt - first;             // 1 .. 7
                    uint32_t *q = dst + first;
                    if (len & 4u) { __builtin_memcpy(q, x, 16); q += 4; }
                    if (len & 2u) { __builtin_memcpy(q, x, 8); q += 2; }
                    if (len & 1u) *q = x[0];
                } else {
#pragma unroll
                    for (int j = 0; j < RLE_VPT; ++j)
                        if ((uint32_t)j >= first && (uint32_t)j < cnt) dst[j] = x[j];
                }
            }
            // out of the staging area: values p0 .. pe - 1, four per lane and step
            const uint32_t nv = pe - p0;                  // <= 512
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t idx = (uint32_t)q * 256u + lane * 4u;
                if (idx < nv) {
                    uint32_t y[4];
                    __builtin_memcpy(y, stg + idx, 16);
                    sink.store4((uint64_t)p0 + idx, nv - idx < 4u ? nv - idx : 4u, y, sh);
                }
            }
        }
        if (tn >= ntiles) break;
        t = tn;
    }
}

// Pass 4 of a round of the dense layout whose items are described by item_bits (streams of up to 4 bits, linear
// window).  As rle_expand_ring: an item's run from a population count, the tile's values side by side in the wave's
// staging area, stores of 16 contiguous bytes per lane.  A value is staged as ONE BYTE (a literal of up to 4 bits; the
// unmasked byte of a repeated run): an item is 8 bytes, written with one LDS store when it is whole and with three
// (4, 2, 1 bytes by the bits of its length) when it is not, and levels are counted four at a time on the staged bytes.
template <typename Sink, typename SH>
__device__ __forceinline__ void rle_expand_dense(SH &sh, uint32_t b, const PageView &pv, const RleWinRef wr, uint32_t ob, uint32_t oe,
                                                 uint32_t w, const Sink &sink, bool count_levels, uint32_t max_level,
                                                 uint32_t &hits, uint32_t &above) {
    const uint32_t lane = lane_id();
    const uint8_t *win = wr.base;
    const uint32_t *run_out = sh.run_out[b];
    const uint32_t *run_item = sh.run_item[b];
    const uint16_t *run_info = sh.run_info[b];
    uint8_t *stg = sh.stage8[first_u32(threadIdx.x >> 6)];
    const uint32_t n_items = sh.n_items[b];
    const uint32_t ntiles = (n_items + 63u) >> 6;
    uint32_t tv = 0;
    if (lane == 0) tv = atomicAdd(wr.tile_ctr, 1u);
    // lane j: the bit word of tile j and the runs that start before it
    const uint64_t my_bits = wr.item_bits[lane];
    const uint32_t my_pc = (uint32_t)__popcll(my_bits);
    const uint32_t my_rank = wave_iscan(my_pc) - my_pc;
    const uint32_t lvl4 = (max_level & 0x7Fu) * 0x01010101u;
    const uint32_t gt4 = (0x7Fu - (max_level & 0x7Fu)) * 0x01010101u;
    uint32_t t = first_u32(tv);
    if (t >= ntiles) return;
    for (;;) {
        const uint64_t bits = bcast_u64(my_bits, (int)t);
        const uint32_t rank = bcast_u32(my_rank, (int)t);
        uint32_t tnv = 0;
        if (lane == 0) tnv = atomicAdd(wr.tile_ctr, 1u);          // the tile after this one
        const uint32_t item = t * 64u + lane;
        const uint32_t lo = rank + __builtin_amdgcn_mbcnt_hi((uint32_t)(bits >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bits, 0u)) +
                            (uint32_t)((bits >> lane) & 1ull) - 1u;
        const uint32_t ri = run_item[lo], ro = run_out[lo], re = run_out[lo + 1];
        const uint32_t info = run_info[lo];
        const uint32_t tn = first_u32(tnv);
        const uint32_t g = item - ri;
        const uint32_t v0 = ro + g * RLE_VPT;
        const uint32_t cnt = re - v0 < RLE_VPT ? re - v0 : (uint32_t)RLE_VPT;
        const uint32_t first = v0 < ob ? (ob - v0 < RLE_VPT ? ob - v0 : (uint32_t)RLE_VPT) : 0u;
        const bool act = item < n_items && v0 < re && v0 < oe && first < cnt;
        const uint64_t am = __ballot(act);
        if (am) {
            // the tile's part of the output: from the first value of its first item to the last of its last
            const uint32_t p0 = bcast_u32(v0 + first, (int)__builtin_ctzll(am));
            const uint32_t pe = bcast_u32(v0 + cnt, 63 - (int)__builtin_clzll(am));
            if (act) {
                const uint32_t rel = info >> 1;
                uint32_t plo, phi;                                 // the item's values, a byte each
                if (info & 1u) {
                    const uint32_t byte = rel + g * w;             // window offset of the item's w bytes
                    uint32_t d = 0;
                    if (byte + w <= wr.lim) {
                        __builtin_memcpy(&d, win + byte, 4);       // (the window keeps RLE_SLACK readable bytes past its end)
                    } else {
                        // the part of a round's first run that lies past the window
                        const uint64_t bit0 = ((uint64_t)(wr.wlo + (int64_t)rel) + (uint64_t)g * w) * 8;
                        for (uint32_t j = first; j < cnt; ++j) d |= rle_field_global(pv, bit0 + (uint64_t)j * w, w) << (j * w);
                    }
                    const uint32_t m = (1u << w) - 1u;
                    plo = (d & m) | (((d >> w) & m) << 8) | (((d >> (2 * w)) & m) << 16) | (((d >> (3 * w)) & m) << 24);
                    const uint32_t e = d >> (4 * w);
                    phi = (e & m) | (((e >> w) & m) << 8) | (((e >> (2 * w)) & m) << 16) | (((e >> (3 * w)) & m) << 24);
                } else {
                    // one raw byte, not masked to w (bit_stream_utils.hh:368-393)
                    uint32_t rep_val;
                    if (rel + 1u <= wr.lim) rep_val = win[rel];
                    else rep_val = rle_field_global(pv, (uint64_t)(wr.wlo + (int64_t)rel) * 8, 8);
                    plo = phi = rep_val * 0x01010101u;
                }
                uint8_t *dst = stg + (int32_t)(v0 - p0);          // (with first > 0 the bytes below p0 are not written)
                if (first == 0 && cnt == RLE_VPT) {
                    const uint64_t pk = ((uint64_t)phi << 32) | plo;
                    __builtin_memcpy(dst, &pk, 8);
                } else {
                    uint64_t pk = (((uint64_t)phi << 32) | plo) >> (8 * first);
                    const uint32_t len = cnt - first;              // 1 .. 7
                    uint8_t *q = dst + first;
                    if (len & 4u) { const uint32_t x4 = (uint32_t)pk; __builtin_memcpy(q, &x4, 4); pk >>= 32; q += 4; }
                    if (len & 2u) { const uint16_t x2 = (uint16_t)pk; __builtin_memcpy(q, &x2, 2); pk >>= 16; q += 2; }
                    if (len & 1u) *q = (uint8_t)pk;
                }
            }
            // out of the staging area: values p0 .. pe - 1, four per lane and step
            const uint32_t nv = pe - p0;                  // <= 512
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const uint32_t idx = (uint32_t)q * 256u + lane * 4u;
                if (idx < nv) {
                    const uint32_t k = nv - idx < 4u ? nv - idx : 4u;
                    const uint32_t x4 = *reinterpret_cast<const uint32_t *>(stg + idx);
                    if (count_levels) {
                        // one flag bit per byte: the byte equals the maximum level / lies above it (a maximum level is below 128)
                        const uint32_t valid = 0x80808080u >> (8u * (4u - k));
                        const uint32_t x = x4 ^ lvl4;
                        const uint32_t eq = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x) & valid;
                        const uint32_t gt = (((x4 & 0x7F7F7F7Fu) + gt4) | x4) & valid;
                        hits += (uint32_t)__popc(eq);
                        if (gt) above += (uint32_t)__popc(gt);
                    }
                    uint32_t y[4] = {x4 & 0xFFu, (x4 >> 8) & 0xFFu, (x4 >> 16) & 0xFFu, x4 >> 24};
                    sink.store4((uint64_t)p0 + idx, k, y, sh);
                }
            }
        }
        if (tn >= ntiles) break;
        t = tn;
    }
}

// Shared body: rounds of (window, find headers, place runs, expand) over one hybrid stream `pv`.
// `chunk` = batch size in which the reference's caller drains the decoder (dict: 1000, else 0).
// Returns the number of values produced (uniform across the workgroup).
template <typename Sink, typename SH>
__device__ __forceinline__ uint64_t rle_decode_stream(SH &sh, const PageView &pv, uint64_t n, uint32_t w,
                                                      uint32_t tbytes, uint32_t chunk, const Sink &sink,
                                                      bool count_levels, uint32_t max_level,
                                                      const RleSpan span = RleSpan{0, 0, 0, ~0ull}) {
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = first_u32(tid >> 6);
    uint64_t cur = span.cur0;         // walker state (wave 0)
    uint64_t outp = span.out0;
    uint32_t hits = 0, above = 0;
    if (tid == 0) sh.irregular = 0;
    if (n <= span.out0) { __syncthreads(); return span.out0; }
    constexpr int NPIECE = SH::DWIN / 16;                                  // 16-byte pieces of a window
    constexpr int PIECES = (NPIECE + RLE_THREADS - 1) / RLE_THREADS;       // per thread
    static_assert(SH::DWIN % 16 == 0, "whole 16-byte pieces");
    struct WinRegs { uint4 v[PIECES]; };
    // pieces tid, tid + 256, ... of the window that covers stream offset pos (16-byte aligned global addresses; bytes
    // past the input read 0)
    auto fetch = [&](uint64_t pos, int64_t &lo) {
        const uint32_t mis = (uint32_t)((uintptr_t)(pv.base + pos) & 15u);
        lo = (int64_t)pos - (int64_t)mis;
        WinRegs r;
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int64_t o = lo + ((int64_t)i * RLE_THREADS + tid) * 16;
            const int64_t room = (int64_t)pv.room - o;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (i * RLE_THREADS + (int)tid < NPIECE) {
                if (room >= 16) {
                    v = P4S_LD(reinterpret_cast<const uint4 *>(pv.base + o), 66);
                } else if (room > 0) {
                    uint8_t tmp[16];
                    for (int c = 0; c < 16; ++c) tmp[c] = c < room ? P4S_LD(pv.base + o + c, 67) : 0;
                    v = *reinterpret_cast<uint4 *>(tmp);
                }
            }
            r.v[i] = v;
        }
        return r;
    };
    auto stage = [&](uint32_t b, const WinRegs &r) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i)
            if (i * RLE_THREADS + (int)tid < NPIECE) *reinterpret_cast<uint4 *>(sh.win[b] + (i * RLE_THREADS + tid) * 16) = r.v[i];
        if (tid < RLE_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win[b] + SH::DWIN)[tid] = 0;
    };
    // ---- prologue: round 0 staged, walked and placed
    int64_t wlo[2] = {0, 0};
    stage(0, fetch(span.cur0, wlo[0]));
    __syncthreads();
    // passes 2 and 3 of a round: the walking wave alone when the round's runs are all of the common kind
    auto walk = [&](uint32_t b, uint64_t out_before) {
        bool walked;
        if constexpr (SH::DENSE) walked = rle_walk_dense(sh, b, pv, wlo[b], cur, out_before, n, w, tbytes, span.in_limit);
        else walked = rle_walk_fast(sh, b, pv, wlo[b], cur, out_before, n, w, tbytes, span.in_limit);
        if (!walked) {
#ifdef P4S_EXP_TIMING
            if (lane_id() == 0) atomicAdd(&sh.cnt_fallback, 1u);
#endif
            rle_find_headers(sh, b, pv, wlo[b], cur, w, tbytes, span.in_limit);
            if (lane_id() == 0) {
                sh.need_place = 1u;
                sh.bits_ok[b] = 0u;      // the placing pass's table is searched
            }
        }
    };
    // the bit words of a round that rle_walk_dense has walked (and cleared them for): which items start a run.  Done
    // by all threads behind the barrier that ends the walk, not in the walk, whose chain would wait for the atomics
    auto mark_items = [&](uint32_t b) {
        {
            if (sh.bits_ok[b]) {
                const uint32_t nr = sh.nruns[b];
                for (uint32_t k = tid; k < nr; k += RLE_THREADS) {
                    const uint32_t it0 = sh.run_item[b][k];
                    atomicOr(reinterpret_cast<uint32_t *>(sh.item_bits[b]) + (it0 >> 5), 1u << (it0 & 31u));
                }
            }
        }
    };
    auto place = [&](uint32_t b, uint64_t out_before) -> uint64_t {
        if (sh.need_place) return rle_place_runs(sh, b, pv, wlo[b], out_before, n, w, tbytes, chunk);
        return sh.out_end;
    };
    if (wave == 0) walk(0, outp);
    __syncthreads();
    uint64_t oe = place(0, outp);
    mark_items(0);
    bool finished = sh.finished != 0;
    if (!finished) stage(1, fetch(sh.next_cur, wlo[1]));
    if (tid == 0) sh.tile_next = 0;
    __syncthreads();
#ifdef P4S_EXP_TIMING      // cycles per page: walk (wave 0) | expand, barrier after expand | place, stage (window wait)
    uint64_t racc[5] = {0, 0, 0, 0, 0}, rlast = __builtin_readcyclecounter();
#define RLE_TICK(i) do { const uint64_t n_ = __builtin_readcyclecounter(); racc[i] += n_ - rlast; rlast = n_; } while (0)
#else
#define RLE_TICK(i) do { } while (0)
#endif
    for (uint32_t k = 0;; ++k) {
        const uint32_t b = k & 1u, nb = b ^ 1u;
        RLE_TICK(4);
#ifdef P4S_EXP_TIMING
        if (tid == 0) atomicAdd(&sh.cnt_rounds, 1u);
#endif
        // wave 0 walks round k+1 while the others already expand round k, then joins them
        if (wave == 0 && !finished) {
            // the chain is one long dependency: let its instructions issue ahead of the expanders sharing the SIMD
            __builtin_amdgcn_s_setprio(3);
            walk(nb, oe);
            __builtin_amdgcn_s_setprio(0);
        }
        RLE_TICK(0);
#ifndef P4S_EXP_RLE_NOEXPAND     // timing experiment: header chain and run placement only
        if constexpr (!Sink::COUNT_ONLY) {
            const uint64_t ob = outp > span.out_begin ? outp : span.out_begin;
            if (ob < oe) {
                bool searched = true;
                if constexpr (SH::DENSE) {
                    if (sh.bits_ok[b]) {
                        searched = false;
                        rle_expand_dense(sh, b, pv, RleWinRef{sh.win[b], 0u, (uint32_t)SH::DWIN, wlo[b], sh.item_bits[b], &sh.tile_next}, (uint32_t)ob, (uint32_t)oe, w, sink, count_levels, max_level, hits, above);
                    }
                }
                if (searched) {
                    // (the search is for tables of the placing pass and rounds of more items than the bit words hold)
                    const uint64_t *ib = (!SH::DENSE && sh.bits_ok[b]) ? sh.item_bits[b] : nullptr;
                    rle_expand_round(sh, b, pv, RleWinRef{sh.win[b], 0u, (uint32_t)SH::DWIN, wlo[b], ib, &sh.tile_next}, ob, oe, w, sink, count_levels, max_level, hits, above);
                }
            }
        }
#endif
        outp = oe;
        RLE_TICK(1);
        __syncthreads();
        RLE_TICK(2);
        if (finished) break;
        // the window after next is fetched now (its position is known) and staged once round k+1 is placed
        const bool more_stream = sh.stream_end == 0;
        int64_t lo2 = 0;
        WinRegs v2{};
        if (more_stream) v2 = fetch(sh.next_cur, lo2);
        oe = place(nb, outp);
        mark_items(nb);
        finished = sh.finished != 0;
        RLE_TICK(3);
        if (!finished) { stage(b, v2); wlo[b] = lo2; }
        if (tid == 0) sh.tile_next = 0;
        __syncthreads();
    }
#ifdef P4S_EXP_TIMING
    if (tid == P4S_EXP_TIMING) {
        sh.dbg[0] = (racc[0] & 0xFFFFFFFFull) | (racc[1] << 32);
        sh.dbg[1] = (racc[2] & 0xFFFFFFFFull) | (racc[3] << 32);
        sh.dbg[2] = (uint64_t)sh.cnt_iter | ((uint64_t)sh.cnt_fallback << 24) | ((uint64_t)sh.cnt_rounds << 44);
    }
    __syncthreads();
#endif
#undef RLE_TICK
    if (count_levels) {
        __syncthreads();
        if (hits) atomicAdd(&sh.level_hits, hits);
        if (above) atomicAdd(&sh.level_above, above);
        __syncthreads();
    }
    return outp;
}

// The same for the wider values: rounds over the tile ring (rle_walk_ring), and ONE linear round at a time over the
// ring's memory wherever the ring walk stops (rle_find_headers + rle_place_runs decide there); after a linear round
// that does not finish the decode the ring starts again where it ended.
template <typename Sink>
__device__ __forceinline__ uint64_t rle_decode_stream_ring(RleShared &sh, const PageView &pv, uint64_t n, uint32_t w,
                                                           uint32_t tbytes, uint32_t chunk, const Sink &sink,
                                                           bool count_levels, uint32_t max_level,
                                                           const RleSpan span = RleSpan{0, 0, 0, ~0ull}) {
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = first_u32(tid >> 6);
    uint64_t cur = span.cur0;
    uint64_t outp = span.out0;
    uint32_t hits = 0, above = 0;
    if (tid == 0) sh.irregular = 0;
    if (n <= span.out0) { __syncthreads(); return span.out0; }
    const uint64_t end_at = span.in_limit < pv.size ? span.in_limit : pv.size;
    const bool ring_ok = ((w + 7) >> 3) <= tbytes;        // else every repeated run is an error (bit_stream_utils.hh:370-377)
    // 16 bytes of the stream at offset o (a 16-byte aligned global address; bytes past the input read 0)
    auto load16 = [&](int64_t o) {
        const int64_t room = (int64_t)pv.room - o;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (room >= 16) {
            v = P4S_LD(reinterpret_cast<const uint4 *>(pv.base + o), 66);
        } else if (room > 0) {
            uint8_t tmp[16];
            for (int c = 0; c < 16; ++c) tmp[c] = c < room ? P4S_LD(pv.base + o + c, 67) : 0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        return v;
    };
#ifdef P4S_EXP_TIMING      // cycles per page: walk (wave 0) | expand | barrier after expand | stage + barrier | linear rounds and ring starts
    uint64_t racc[5] = {0, 0, 0, 0, 0}, rlast = __builtin_readcyclecounter();
#define RLE_TICK(i) do { const uint64_t n_ = __builtin_readcyclecounter(); racc[i] += n_ - rlast; rlast = n_; } while (0)
#else
#define RLE_TICK(i) do { } while (0)
#endif
    RleWalkTicks wticks{0, 0, 0};
    for (;;) {
        // ---- ring rounds, while the stream ahead is long enough to be worth it
        if (ring_ok && cur + 2 * RLE_TILE + 2048 <= end_at) {
            // Wave 0 walks, and it alone brings in the tiles, with loads that go from global memory straight to LDS
            // (no registers held while the tile is on its way).  The counter a wave waits on for its loads also counts
            // its stores, in issue order: a wave that streams output would have to wait for all of its stores to be
            // acknowledged before it could be sure of a tile.  So the other waves expand and never wait for memory;
            // wave 0 joins them once its walk is done and the tile has landed: its own stores of a round are then a whole
            // walk older than the next wait, which they no longer hold up.
            // A tile is asked for right after the barrier that ends a round (the slot it goes to is dead from then on),
            // waited for by wave 0 before the next round's barrier, and used from the round after that.
            const int64_t base = (int64_t)cur - (int64_t)((uintptr_t)(pv.base + cur) & 15u);      // stream offset of tile 0
            const uint32_t lane = tid & 63u;
            auto issue_tile = [&](int32_t t) {                 // wave 0
                uint8_t *slot = sh.win[0] + (uint32_t)(t % RLE_RING_TILES) * (uint32_t)RLE_TILE;
#pragma unroll
                for (int j = 0; j < RLE_TILE / 1024; ++j) {
                    const int64_t off = base + (int64_t)t * RLE_TILE + (int64_t)(j * 64 + lane) * 16;
                    const int64_t room = (int64_t)pv.room - off;
                    if (room >= 16) {
                        // (written as assembly so that the compiler does not know of a load that writes LDS: it would
                        // make every wave wait for ALL of its outstanding stores, vmcnt(0), in front of the first LDS
                        // store of every expansion tile, since it cannot tell which wave issued the load)
                        if (P4S_INB(pv.base + off, 16, 66)) {
                            const uint32_t lds_at = first_u32((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)(slot + j * 1024));
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"      // (m0 is reserved: named all the same, the load takes its LDS address from it)
                            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off"
                                         :: "s"(lds_at), "v"(pv.base + off) : "memory", "m0");
#pragma clang diagnostic pop
                        }
                    } else if (room > 0) {                       // the piece that holds the end of the input
                        *reinterpret_cast<uint4 *>(slot + (j * 64 + lane) * 16) = load16(off);
                    }
                }
            };
            auto tile_arrived = [&](int32_t t) {               // wave 0: the tile's loads have landed; the ring's mirror
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (t % RLE_RING_TILES == 0 && lane < RLE_MIRROR / 16)
                    *reinterpret_cast<uint4 *>(sh.win[0] + RLE_RING + lane * 16) = *reinterpret_cast<const uint4 *>(sh.win[0] + lane * 16);
            };
            if (wave == 0) {
                issue_tile(0);
                issue_tile(1);
                tile_arrived(0);
                issue_tile(2);
            }
            int32_t staged = 1;
            bool pending = true;                  // tile staged + 1 is on its way
            uint64_t wcur = cur;                  // the walk's state (wave 0)
            uint32_t wout = (uint32_t)outp;
            if (tid == 0) { sh.tile_ctr[0] = 0; sh.tile_ctr[1] = 0; }
            __syncthreads();
            if (wave == 0) rle_walk_ring(sh, 0, base, staged, end_at, wcur, wout, n, w, wticks);
            lds_barrier();
            uint32_t idle = 0;
            RLE_TICK(4);
            for (uint32_t i = 0;; ++i) {
                const uint32_t b = i & 1u, nb = b ^ 1u;
                const RleRoundRec rr = sh.rnd[b];
                // a round that neither produced anything nor moved on waits for a tile; it must not wait for ever
                idle = (rr.out_end == (uint32_t)outp && rr.next_cur == cur) ? idle + 1 : 0;
                const bool stop = rr.stop != 0 || idle >= 6;
                if (wave == 0) {
                    // the chain is one long dependency: let its instructions issue ahead of the expanders sharing the SIMD
                    __builtin_amdgcn_s_setprio(3);
                    if (!stop) rle_walk_ring(sh, nb, base, staged, end_at, wcur, wout, n, w, wticks);
                    __builtin_amdgcn_s_setprio(0);
                    if (lane == 0) sh.tile_ctr[nb] = 0;
                    if (pending) tile_arrived(staged + 1);
                }
                RLE_TICK(0);
#ifndef P4S_EXP_RLE_NOEXPAND     // timing experiment: header chain and run placement only
                if constexpr (!Sink::COUNT_ONLY) {
                    const uint64_t ob = outp > span.out_begin ? outp : span.out_begin;
                    const uint64_t oe = rr.out_end;
                    if (ob < oe) {
                        const int32_t last = staged < rr.tile + 1 ? staged : rr.tile + 1;
                        const RleWinRef wr{sh.win[0], (uint32_t)(rr.tile % RLE_RING_TILES) * (uint32_t)RLE_TILE,
                                           (uint32_t)(last + 1 - rr.tile) * (uint32_t)RLE_TILE, base + (int64_t)rr.tile * RLE_TILE,
                                           rr.bits ? sh.item_bits[b] : nullptr, &sh.tile_ctr[b]};
                        if (rr.bits) rle_expand_ring(sh, b, pv, wr, (uint32_t)ob, (uint32_t)oe, w, sink, count_levels, max_level, hits, above);
                        else rle_expand_round(sh, b, pv, wr, ob, oe, w, sink, count_levels, max_level, hits, above);
                    }
                }
#endif
                outp = rr.out_end;
                cur = rr.next_cur;
                RLE_TICK(1);
                lds_barrier();
                RLE_TICK(2);
                if (pending) { ++staged; pending = false; }
                if (stop) break;
                // the round walked just now is expanded next: tiles up to its own + 2 fit the ring beside it
                const RleRoundRec rn = sh.rnd[nb];
                if (!rn.stop && staged + 1 <= rn.tile + 2 && (uint64_t)(base + (int64_t)(staged + 1) * RLE_TILE) < end_at) {
                    if (wave == 0) issue_tile(staged + 1);
                    pending = true;
                }
                RLE_TICK(3);
            }
            if (wave == 0 && pending) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // nothing lands in the ring after this
            __syncthreads();
        }
        // ---- one linear round where the walk stands
        {
            const int64_t wlo = (int64_t)cur - (int64_t)((uintptr_t)(pv.base + cur) & 15u);
#pragma unroll
            for (int i = 0; i < (RLE_RING / 16 + RLE_THREADS - 1) / RLE_THREADS; ++i)
                if (i * RLE_THREADS + (int)tid < RLE_RING / 16)
                    *reinterpret_cast<uint4 *>(sh.win[0] + (i * RLE_THREADS + tid) * 16) = load16(wlo + (int64_t)(i * RLE_THREADS + tid) * 16);
            if (tid < RLE_MIRROR / 4) reinterpret_cast<uint32_t *>(sh.win[0] + RLE_RING)[tid] = 0;
            __syncthreads();
            if (wave == 0) {
                uint64_t c = cur;
                rle_find_headers(sh, 0, pv, wlo, c, w, tbytes, span.in_limit);
                if (lane_id() == 0) sh.need_place = 1u;
            }
            __syncthreads();
            const uint64_t oe = rle_place_runs(sh, 0, pv, wlo, outp, n, w, tbytes, chunk);
            const bool finished = sh.finished != 0;
            if (tid == 0) sh.tile_next = 0;
            __syncthreads();
            if constexpr (!Sink::COUNT_ONLY) {
                const uint64_t ob = outp > span.out_begin ? outp : span.out_begin;
                if (ob < oe)
                    rle_expand_round(sh, 0, pv, RleWinRef{sh.win[0], 0u, (uint32_t)RLE_RING, wlo, nullptr, &sh.tile_next}, ob, oe, w, sink, count_levels, max_level,
                                     hits, above);
            }
            outp = oe;
            cur = sh.next_cur;
            __syncthreads();
            RLE_TICK(4);
            if (finished) break;
        }
    }
#ifdef P4S_EXP_TIMING
    if (tid == P4S_EXP_TIMING) {
#ifdef P4S_EXP_TIMING_WALK      // the walk's phases instead: tile bytes to registers | chain | placement
        racc[0] = wticks.t0; racc[1] = wticks.t1; racc[2] = wticks.t2; racc[3] = 0;
#endif
        sh.dbg[0] = (racc[0] & 0xFFFFFFFFull) | (racc[1] << 32);
        sh.dbg[1] = (racc[2] & 0xFFFFFFFFull) | (racc[3] << 32);
        sh.dbg[2] = racc[4];
    }
    __syncthreads();
#endif
#undef RLE_TICK
    if (count_levels) {
        if (hits) atomicAdd(&sh.level_hits, hits);
        if (above) atomicAdd(&sh.level_above, above);
        __syncthreads();
    }
    return outp;
}

// ---------------------------------------------------------------------------
// The three entry points built on the hybrid decoder differ in how a page descriptor turns into a stream:
//   RAW     p4s_decode_rle: the bare stream, param0 = bit width (rle_decoder_boolean with width 1);
//   LEVELS  p4s_decode_levels: level_decoder::reset_v1 / reset_v2 + one read_batch (encoding.hh:42-82, encoding.cc:26-79);
//   DICT    p4s_decode_dictionary: width byte + indices, gathered through the dictionary (encoding.cc:468-508).
// rle_setup does that part, for the decode kernel and for the passes that split a page over several workgroups.
// ---------------------------------------------------------------------------
enum RleKind : int { RLE_KIND_RAW = 0, RLE_KIND_LEVELS = 1, RLE_KIND_DICT = 2 };

struct RleSetup {
    PageView pv;          // the hybrid stream
    uint64_t n;           // values one read_batch is asked for
    uint64_t consumed;    // levels: what reset_v1 returns
    uint32_t w;
    uint32_t max_level;   // levels
    int32_t code;         // outcome known without decoding: P4S_E_* (with aux0 / aux1) ...
    uint64_t aux0, aux1;
    uint32_t mode;        // 0: decode the stream; 1: finished (status = code, values = n, aux0 = n when code == 0);
                          // 2: BIT_PACKED levels (no run headers)
};

template <int KIND>
__device__ __forceinline__ RleSetup rle_setup(const uint8_t *d_in, uint64_t in_bytes, const p4s_page_desc &pd) {
    RleSetup su;
    su.pv = page_view(d_in, in_bytes, pd);
    su.n = pd.num_values;
    su.consumed = 0; su.w = 0; su.max_level = 0; su.code = 0; su.aux0 = 0; su.aux1 = 0; su.mode = 0;
    if constexpr (KIND == RLE_KIND_RAW) {
        su.w = pd.param0;
        if (su.w > 32) { su.code = P4S_E_BAD_ENCODING; su.aux0 = su.w; su.n = 0; su.mode = 1; }
    } else if constexpr (KIND == RLE_KIND_DICT) {
        if (su.pv.size == 0) { su.code = P4S_E_DICT_EMPTY_PAGE; su.n = 0; su.mode = 1; return su; }
        su.w = P4S_LD(su.pv.base, 70);
        if (su.w > 32) { su.code = P4S_E_DICT_BAD_WIDTH; su.aux0 = su.w; su.n = 0; su.mode = 1; return su; }   // encoding.cc:473-477
        su.pv.base += 1; su.pv.room -= 1; su.pv.size -= 1;
    } else {
        const uint32_t max_level = pd.param0;
        su.max_level = max_level;
        su.w = max_level ? 32u - (uint32_t)__builtin_clz(max_level) : 0u;                    // encoding.hh:47-49
        su.n = pd.num_values < pd.param1 ? pd.num_values : pd.param1;                       // encoding.hh:63
        if (su.w == 0) { su.mode = 1; su.aux0 = su.n; return su; }                          // encoding.hh:64-68: zeros, all equal to max level 0
        const uint32_t kind = pd.flags & 3u;
        PageView &pv = su.pv;
        if (kind == P4S_LEVELS_V1_RLE) {                                                    // encoding.cc:35-50
            if (pv.size < 4) { su.code = P4S_E_LEVELS_EOF; su.aux0 = 4; su.aux1 = pv.size; su.n = 0; su.mode = 1; return su; }
            const uint32_t raw = (uint32_t)P4S_LD(pv.base, 69) | ((uint32_t)P4S_LD(pv.base + 1, 69) << 8) |
                                 ((uint32_t)P4S_LD(pv.base + 2, 69) << 16) | ((uint32_t)P4S_LD(pv.base + 3, 69) << 24);
            const int32_t len = (int32_t)raw;
            if (len < 0) { su.code = P4S_E_LEVELS_NEGATIVE; su.aux0 = (uint64_t)(int64_t)len; su.n = 0; su.mode = 1; return su; }
            if ((uint64_t)len > pv.size) { su.code = P4S_E_LEVELS_EOF; su.aux0 = (uint64_t)len; su.aux1 = pv.size; su.n = 0; su.mode = 1; return su; }
            // the reference checks len against size(), not size()-4; clamp instead of running past the page
            const uint64_t use = (uint64_t)len < pv.size - 4 ? (uint64_t)len : pv.size - 4;
            su.consumed = 4 + use;
            pv.base += 4; pv.room -= 4; pv.size = use;
        } else if (kind == P4S_LEVELS_V1_BIT_PACKED) {                                      // encoding.cc:51-62
            const uint64_t byte_len = ((uint64_t)pd.param1 * su.w + 7) >> 3;
            if (byte_len > 0x7FFFFFFFull) { su.code = P4S_E_LEVELS_BITPACKED_LEN; su.aux0 = byte_len; su.n = 0; su.mode = 1; return su; }
            if (byte_len > pv.size) { su.code = P4S_E_LEVELS_EOF; su.aux0 = byte_len; su.aux1 = pv.size; su.n = 0; su.mode = 1; return su; }
            su.consumed = byte_len;
            su.mode = 2;
        } else {                                                                            // v2: encoding.cc:68-79
            if (pv.size > 0x7FFFFFFFull) { su.code = P4S_E_LEVELS_V2_LEN; su.aux0 = pv.size; su.n = 0; su.mode = 1; return su; }
            su.consumed = pv.size;
        }
    }
    return su;
}

// One part of a page that is split over several workgroups (written by k_rle_plan_parts).
struct RlePart {
    RleSpan span;
    uint64_t n;           // the decode ends at this output index
};

// Per-page record of the split decode (device scratch).
struct RleSplitPage {
    unsigned long long errkey;   // dict: min over (position << 32 | bad index), ~0: none
    uint32_t hits, above;        // levels: values equal to / above the maximum level
    uint32_t win0, nwin;         // the page's windows in the per-window arrays
    uint32_t nparts;             // parts the page is decoded in; 0: the page is decoded whole by one workgroup
    uint32_t mis;                // bytes by which the stream starts past a 16-byte aligned address
    uint32_t narrow;             // the page's values are at most RLE_SPLIT_NARROW bits wide
    uint32_t pad;
};

template <int KIND, int ELEM> struct RleSinkOf { using type = RleSinkPlain<uint32_t>; };
template <> struct RleSinkOf<RLE_KIND_RAW, 1> { using type = RleSinkPlain<uint8_t>; };
template <> struct RleSinkOf<RLE_KIND_RAW, 2> { using type = RleSinkPlain<uint16_t>; };
template <> struct RleSinkOf<RLE_KIND_RAW, 4> { using type = RleSinkPlain<uint32_t>; };
template <> struct RleSinkOf<RLE_KIND_LEVELS, 2> { using type = RleSinkPlain<uint16_t>; };
template <> struct RleSinkOf<RLE_KIND_LEVELS, 4> { using type = RleSinkPlain<uint32_t>; };
template <int ELEM> struct RleSinkOf<RLE_KIND_DICT, ELEM> { using type = RleSinkDict<ELEM>; };

template <int KIND, int ELEM>
__device__ __forceinline__ typename RleSinkOf<KIND, ELEM>::type rle_make_sink(uint8_t *d_out, const p4s_page_desc &pd,
                                                                               const uint8_t *d_dict, uint64_t dict_size) {
    if constexpr (KIND == RLE_KIND_DICT) {
        return RleSinkDict<ELEM>{d_out + pd.out_offset * ELEM, d_dict, dict_size > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dict_size};
    } else {
        using OutT = decltype(RleSinkOf<KIND, ELEM>::type::out);
        return typename RleSinkOf<KIND, ELEM>::type{reinterpret_cast<OutT>(d_out) + pd.out_offset};
    }
}

// The decode kernel of the three entry points.  With `parts` null: one workgroup per page (blockIdx.x = page).
// With `parts` set (a batch of few pages, split so that the whole chip works on them): blockIdx.x = page * fan + s;
// part s of the page is decoded like a page of its own, what it counts goes to the page's record, and the page's
// status is written afterwards by k_rle_finish; a page whose record says nparts == 0 is decoded whole by s == 0.
template <int KIND, int ELEM>
__global__ __launch_bounds__(RLE_THREADS, 8) void k_rle(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        const uint8_t *__restrict__ d_dict, uint64_t dict_size,
        uint8_t *__restrict__ d_out, p4s_page_status *__restrict__ status,
        const RlePart *__restrict__ parts, RleSplitPage *__restrict__ split, uint32_t fan) {
    __shared__ union { RleShared wide; RleSharedDense dense; RleSharedDict dict; } lds;      // same scalar members at the same offsets
    RleShared &sh = lds.wide;
    const uint32_t tid = threadIdx.x;
    uint32_t page = blockIdx.x;
    RleSpan span{0, 0, 0, ~0ull};
    bool part_mode = false;
    uint64_t part_n = 0;
    if (parts) {
        page = blockIdx.x / fan;
        const uint32_t s = blockIdx.x % fan;
        const uint32_t np = split[page].nparts;
        if (np == 0) { if (s != 0) return; }
        else {
            if (s >= np) return;
            part_mode = true;
            const RlePart pt = parts[(uint64_t)page * fan + s];
            span = pt.span;
            part_n = pt.n;
        }
    }
    const p4s_page_desc pd = pages[page];
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pd);
    if (tid == 0) { sh.level_hits = 0; sh.level_above = 0; sh.errkey = ~0ull; }
#ifdef P4S_EXP_TIMING
    if (tid == 0) { sh.cnt_iter = 0; sh.cnt_fallback = 0; sh.cnt_rounds = 0; }
#endif
    if (su.mode == 1) {
        if constexpr (KIND == RLE_KIND_LEVELS) {
            if (su.code == 0) {      // bit width 0: zeros
                using OutT = decltype(RleSinkOf<KIND, ELEM>::type::out);
                OutT out = reinterpret_cast<OutT>(d_out) + pd.out_offset;
                for (uint64_t i = tid; i < su.n; i += RLE_THREADS) P4S_ST(out + i, 0, 68);
            }
        }
        if (tid == 0) write_status(&status[page], su.code, su.n, su.aux0, su.aux1, 0, 0);
        return;
    }
    auto sink = rle_make_sink<KIND, ELEM>(d_out, pd, d_dict, dict_size);
    if constexpr (KIND == RLE_KIND_LEVELS) {
        if (su.mode == 2) {
            // plain bit-packed values, LSB first (BitReader::GetBatch), no run headers: one literal run starting
            // at byte 0, expanded like any other (what lies past the first window comes from global memory)
            const int64_t wlo = window_load<RleShared::DWIN, RLE_THREADS>(sh.win[0], su.pv, 0);
            if (tid < RLE_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win[0] + RleShared::DWIN)[tid] = 0;
            if (tid == 0) {
                sh.nruns[0] = 1;
                sh.run_out[0][0] = 0;
                sh.run_out[0][1] = (uint32_t)su.n;
                sh.run_info[0][0] = (uint16_t)(((uint32_t)(0 - wlo) << 1) | 1u);
                sh.run_item[0][0] = 0;
                sh.run_item[0][1] = sh.n_items[0] = (uint32_t)((su.n + RLE_VPT - 1) / RLE_VPT);
                sh.tile_next = 0;
            }
            __syncthreads();
            uint32_t hits = 0, above = 0;
            rle_expand_round(sh, 0, su.pv, RleWinRef{sh.win[0], 0u, (uint32_t)RleShared::DWIN, wlo, nullptr, &sh.tile_next}, 0, su.n, su.w, sink, true, su.max_level, hits, above);
            __syncthreads();
            if (hits) atomicAdd(&sh.level_hits, hits);
            if (above) atomicAdd(&sh.level_above, above);
            __syncthreads();
            if (tid == 0) write_status(&status[page], P4S_E_NONE, su.n, sh.level_hits, sh.level_above, su.consumed, 0);
            return;
        }
    }
    __syncthreads();
    const uint32_t tbytes = KIND == RLE_KIND_DICT ? 4u : (uint32_t)ELEM;
    const uint32_t chunk = KIND == RLE_KIND_DICT ? 1000u : 0u;
    const uint64_t n = part_mode ? part_n : su.n;
    // level streams (a header every three bytes where one value in ten is null): the layout with the large run table
    // and the walk that finds all headers of a 64-byte span at once by pointer jumping, whose cost does not depend on
    // how many there are (definition levels with 10 % nulls, 2048 pages: 4.2 ms, 6.2 ms through the ring, whose
    // walker goes header by header); other streams of narrow values (8 to 107 values a run) are faster through the ring
    uint64_t got;
    if (KIND == RLE_KIND_LEVELS && su.w <= RLE_DENSE_MAX_WIDTH)
        got = rle_decode_stream(lds.dense, su.pv, n, su.w, tbytes, chunk, sink, true, su.max_level, span);
    else if constexpr (KIND == RLE_KIND_DICT)
        got = rle_decode_stream(lds.dict, su.pv, n, su.w, tbytes, chunk, sink, false, 0u, span);
    else
        got = rle_decode_stream_ring(sh, su.pv, n, su.w, tbytes, chunk, sink, KIND == RLE_KIND_LEVELS, su.max_level, span);
    __syncthreads();
    if (tid != 0) return;
    if (part_mode) {
        if constexpr (KIND == RLE_KIND_LEVELS) {
            if (sh.level_hits) atomicAdd(&split[page].hits, sh.level_hits);
            if (sh.level_above) atomicAdd(&split[page].above, sh.level_above);
        }
        if constexpr (KIND == RLE_KIND_DICT) {
            if (sh.errkey != ~0ull) atomicMin(&split[page].errkey, sh.errkey);
        }
        return;
    }
    if constexpr (KIND == RLE_KIND_DICT) {
        const unsigned long long k = sh.errkey;
        if (k != ~0ull) { write_status(&status[page], P4S_E_DICT_INDEX, 0, dict_size, k & 0xFFFFFFFFull, 0, 0); return; }
    }
#ifdef P4S_EXP_TIMING
    write_status(&status[page], P4S_E_NONE, got, sh.dbg[0], sh.dbg[1], sh.dbg[2], 0);
#else
    if constexpr (KIND == RLE_KIND_LEVELS) write_status(&status[page], P4S_E_NONE, got, sh.level_hits, sh.level_above, su.consumed, 0);
    else write_status(&status[page], P4S_E_NONE, got, 0, 0, 0, 0);
#endif
}

// values_to_read of column_chunk_reader.hh:134-140: copy the level hit count into a value page's n.
static __global__ void k_set_num_values(p4s_page_desc *pages, const p4s_page_status *level_status, uint32_t n_pages) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pages) pages[i].num_values = level_status[i].code == 0 ? (uint32_t)level_status[i].aux0 : 0u;
}

}  // namespace p4s
