// Splitting hybrid-RLE pages over many workgroups, for batches with too few pages to fill the chip with one
// workgroup per page (one page of 1Mi values otherwise takes milliseconds: its run-header chain is one long
// dependency, RleDecoder::NextCounts of include/parquet4seastar/rle_encoding.hh:321-348, walked by one wave).
//
// The only serial fact of the format is where the next run header starts.  The passes below resolve that chain
// in parallel and cut every page into parts that k_rle decodes like pages of their own:
//   k_rle_split_init   per page: the stream (rle_setup) and how many 4 KiB windows it spans;
//   k_rle_split_exits  per window (one wave): every byte position is parsed as if a header started there (where the
//                      header after it would start), from the window's last 64 bytes to its first; a position whose
//                      successor lies further on in the window takes over that one's exit, which is already known
//                      (within the 64 positions of a step: a few rounds of pointer jumping), so that exits[p] =
//                      "where the chain that enters the window at p leaves it", for all p at once;
//   k_rle_split_entry  per page: the true chain through the windows, one table lookup per window;
//   k_rle_split_count  per window (one lane): the true chain through the window, header by header, adding up the
//                      values of its runs, and whether anything but a plain run of a one- or two-byte header was met;
//   k_rle_split_plan   per page: prefix sums of the window counts give every window its first output index; pages
//                      whose chain yields the n values asked for without anything irregular are cut into parts of
//                      equal output size, each starting at the window that holds its first value;
//   k_rle<..>(parts)   the decode proper; k_rle_split_finish writes the statuses of the pages decoded in parts.
// Anything irregular (unreadable header, truncated run, fewer values than asked for, a run longer than 60 KiB)
// sends the page to the whole-page decode, whose results define parity: the split never decides an outcome.
#pragma once
#include "k_rle.cuh"

namespace p4s {

constexpr uint32_t RLE_NEXT_FAIL = 0xFFFFu;     // the header at this position cannot be read (NextCounts returns false)
constexpr uint32_t RLE_NEXT_FAR = 0xFFFEu;      // the next header lies further than a 16-bit window offset reaches
constexpr uint32_t RLE_NO_ENTRY = 0xFFFFFFFFu;  // the chain does not enter this window

// Device scratch of one split decode, carved by the host (p4s_abi_rle.inc).
struct RleSplitArrays {
    RleSplitPage *pages;      // [n_pages]
    uint32_t *total_windows;  // [1]
    uint32_t *win_page;       // [cap]   page of a window
    uint32_t *entry;          // [cap]   window offset at which the chain enters the window
    uint32_t *wflag;          // [cap]   the window's runs hold something irregular
    uint64_t *wcount;         // [cap]   values of the runs that start in the window
    uint64_t *wbase;          // [cap]   output index of the window's first run
    uint16_t *exits;          // [cap * RLE_WIN]
    uint16_t *ck_pos;         // [cap * RLE_CKS]  window offset of the first header of the chain in every 256 bytes of the window
    uint32_t *ck_cnt;         // [cap * RLE_CKS]  values of the window's runs before that header
    RlePart *parts;           // [n_pages * fan]
    uint32_t cap;             // windows the arrays hold
};

constexpr uint32_t RLE_CKS = RLE_WIN / 256;      // places per window at which a part can start
constexpr uint32_t RLE_CK_NONE = 0xFFFFu;        // no header of the chain in these 256 bytes
constexpr uint32_t RLE_SPLIT_NARROW = 4;         // bit widths up to this make short runs: hundreds of headers in a window
constexpr uint32_t RLE_SPLIT_MAX_PAGES = 1024;   // batches of more pages fill the chip with one workgroup per page

// One workgroup per page: windows per page and their place in the per-window arrays (n_pages <=
// RLE_SPLIT_MAX_PAGES).  Every workgroup adds up the windows of all pages by itself (a few loads per thread and one
// scan) and then fills in its own page.  Pages from the first one whose windows no longer fit the arrays on are
// decoded whole.
template <int KIND>
__global__ __launch_bounds__(RLE_THREADS) void k_rle_split_init(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, uint32_t n_pages,
        RleSplitArrays a) {
    constexpr uint32_t PER = RLE_SPLIT_MAX_PAGES / RLE_THREADS;
    __shared__ uint64_t s_wave[RLE_THREADS / 64];
    __shared__ uint32_t s_mine[4], s_top[RLE_THREADS / 64];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    uint32_t nwin[PER], mis[PER], narrow[PER];
    uint64_t sum = 0;
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
        const uint32_t p = tid * PER + i;
        nwin[i] = 0; mis[i] = 0; narrow[i] = 0;
        if (p < n_pages) {
            const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[p]);
            narrow[i] = su.w <= RLE_SPLIT_NARROW ? 1u : 0u;
            if (su.mode == 0 && su.n > 0 && su.pv.size > 0) {
                mis[i] = (uint32_t)((uintptr_t)su.pv.base & 15u);
                const uint64_t w = (su.pv.size + mis[i] + RLE_WIN - 1) / RLE_WIN;
                nwin[i] = w > a.cap ? a.cap + 1u : (uint32_t)w;
            }
        }
        sum += nwin[i];
    }
    uint64_t incl = wave_iscan(sum);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    for (uint32_t v = 0; v < wave; ++v) incl += s_wave[v];
    uint64_t run = incl - sum;                               // windows of the pages before this thread's
    uint32_t held = 0;                                       // windows of the pages that fit, up to this thread's
#pragma unroll
    for (uint32_t i = 0; i < PER; ++i) {
        const uint32_t p = tid * PER + i;
        const bool fits = run + nwin[i] <= a.cap;
        if (p == blockIdx.x) {
            s_mine[0] = fits ? nwin[i] : 0u;
            s_mine[1] = fits ? (uint32_t)run : 0u;
            s_mine[2] = mis[i];
            s_mine[3] = narrow[i];
        }
        if (fits) held = (uint32_t)(run + nwin[i]);           // page sizes only add up: the pages that fit come first
        run += nwin[i];
    }
    if (blockIdx.x == 0) {
        held = wave_imaxscan(held);
        if (lane == 63) s_top[wave] = held;
    }
    __syncthreads();
    const uint32_t my_nwin = s_mine[0], win0 = s_mine[1];
    if (tid == 0) {
        if (blockIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t v = 0; v < RLE_THREADS / 64; ++v) total = s_top[v] > total ? s_top[v] : total;
            *a.total_windows = total;
        }
        RleSplitPage sp;
        sp.errkey = ~0ull; sp.hits = 0; sp.above = 0; sp.win0 = win0; sp.nwin = my_nwin; sp.nparts = 0; sp.mis = (uint8_t)s_mine[2];
        sp.narrow = s_mine[3]; sp.pad = 0;
        a.pages[blockIdx.x] = sp;
    }
    for (uint32_t k = tid; k < my_nwin; k += RLE_THREADS) a.win_page[win0 + k] = blockIdx.x;
}

// A run header for the split passes: headers of up to three bytes (runs of up to 2^20 values or groups; x = the four
// bytes at the header position, little endian; `avail` = stream bytes from the header on).  `skip` = bytes to the
// header behind the run, `vals` = values of the run; plain = the header is complete, of at most three bytes, and
// its repeated value fits the output type.  Whether the stream holds the whole run is the caller's question
// (skip <= avail).  Anything else is left to the whole-page decode.
struct RleCheapHeader { bool plain; uint32_t skip, vals; };
__device__ __forceinline__ RleCheapHeader rle_parse_cheap(uint32_t x, int64_t avail, uint32_t w, uint32_t vbytes, uint32_t tbytes) {
    RleCheapHeader h;
    const bool two = (x & 0x80u) != 0, three = two && (x & 0x8000u), four = three && (x & 0x800000u);
    const uint32_t v = (x & 0x7Fu) | (two ? (x >> 1) & 0x3F80u : 0u) | (three ? (x >> 2) & 0x1FC000u : 0u);
    const uint32_t hl = 1u + (two ? 1u : 0u) + (three ? 1u : 0u), cnt = v >> 1;
    const bool lit = (v & 1u) != 0;
    h.plain = !four && (int64_t)hl <= avail && (lit || vbytes <= tbytes);
    h.skip = hl + (lit ? cnt * w : vbytes);          // <= 3 + 2^20 * 32
    h.vals = lit ? cnt * 8u : cnt;
    return h;
}

// Per window (one wave): exits[p] = where the chain that enters the window at offset p leaves it.
//   Step 1: every lane takes 64 bytes of the window (in registers) and goes through them from the last to the first:
//           a position whose successor lies among the same 64 bytes takes over the successor's value, so that
//           afterwards every position knows where its chain leaves the lane's 64 bytes.
//   Step 2: groups of 64 positions from the end of the window to its start, a lane per position: a chain that
//           left its 64 bytes inside the window landed on a position that is final already.
// The table is kept with one dword of padding after every 64 entries, so that the lanes of step 1 (128 bytes
// apart) use different LDS banks.
constexpr int RLE_EXIT_THREADS = 64;
__device__ __forceinline__ uint32_t rle_exit_slot(uint32_t p) { return p + ((p >> 6) << 1); }

// A window as one wave holds it: 64 bytes per lane in registers.
struct RleWaveWindow {
    uint32_t seg[16];      // this lane's 64 bytes
    uint32_t bnext;        // the four bytes behind them
    uint32_t w, vbytes;
    bool rep_fits;         // the repeated value of a run fits the output type
    uint32_t left;         // stream bytes from the window's byte 0 on (0 .. 1 MiB)
    uint32_t lim;          // offsets at or past it lie outside the window or the stream: a chain that reaches one has left
    uint32_t first;        // offsets before it lie in front of the stream (a page that starts off the 16-byte grid)
};

template <int KIND, int ELEM>
__device__ __forceinline__ void rle_wave_window_load(RleWaveWindow &ww, const RleSetup &su, const RleSplitPage &sp, uint32_t k, uint32_t lane) {
    const uint32_t tbytes = KIND == RLE_KIND_DICT ? 4u : (uint32_t)ELEM;
    ww.w = su.w; ww.vbytes = (su.w + 7) >> 3;
    ww.rep_fits = ww.vbytes <= tbytes;
    const int64_t so = (int64_t)k * RLE_WIN - (int64_t)sp.mis;          // stream offset of the window's byte 0
    // this lane's 64 bytes (16-byte aligned global address), zero past the input allocation
    {
        const int64_t o = so + (int64_t)lane * 64;
        const int64_t room = (int64_t)su.pv.room - o;
        if (room >= 64) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 v = P4S_LD(reinterpret_cast<const uint4 *>(su.pv.base + o) + j, 110);
                ww.seg[4 * j] = v.x; ww.seg[4 * j + 1] = v.y; ww.seg[4 * j + 2] = v.z; ww.seg[4 * j + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                uint32_t d = 0;
                for (int i = 0; i < 4; ++i)
                    if (4 * j + i < room) d |= (uint32_t)P4S_LD(su.pv.base + o + 4 * j + i, 111) << (8 * i);
                ww.seg[j] = d;
            }
        }
    }
    // the four bytes behind this lane's 64: the next lane's first, and the first of the next window for the last lane
    ww.bnext = __shfl_down(ww.seg[0], 1);
    if (lane == RLE_EXIT_THREADS - 1) {
        const int64_t o = so + RLE_WIN;
        uint32_t d = 0;
        for (int i = 0; i < 4; ++i)
            if (o + i < (int64_t)su.pv.room) d |= (uint32_t)P4S_LD(su.pv.base + o + i, 112) << (8 * i);
        ww.bnext = d;
    }
    const int64_t left64 = (int64_t)su.pv.size - so;
    ww.left = left64 <= 0 ? 0u : (left64 > (1 << 20) ? (1u << 20) : (uint32_t)left64);
    ww.lim = ww.left < (uint32_t)RLE_WIN ? ww.left : (uint32_t)RLE_WIN;
    ww.first = so < 0 ? (uint32_t)(-so) : 0u;
}

// The header that would start at window offset p (x = the four bytes there).
struct RleLaneHeader { bool plain; uint32_t to, vals; };
__device__ __forceinline__ RleLaneHeader rle_lane_header(const RleWaveWindow &ww, uint32_t p, uint32_t x) {
    RleLaneHeader h;
    const bool two = (x & 0x80u) != 0, three = two && (x & 0x8000u), four = three && (x & 0x800000u);
    const uint32_t v = (x & 0x7Fu) | (two ? (x >> 1) & 0x3F80u : 0u) | (three ? (x >> 2) & 0x1FC000u : 0u);
    const uint32_t hl = 1u + (two ? 1u : 0u) + (three ? 1u : 0u), cnt = v >> 1;
    const bool lit = (v & 1u) != 0;
    // positions in front of the stream and headers that the stream does not hold completely are no headers
    h.plain = !four && p + hl <= ww.left && p >= ww.first && (lit || ww.rep_fits);
    h.to = p + hl + (lit ? cnt * ww.w : ww.vbytes);
    h.vals = lit ? cnt * 8u : cnt;
    return h;
}
// The four bytes at byte J of the lane's 64.
template <int J>
__device__ __forceinline__ uint32_t rle_lane_bytes(const RleWaveWindow &ww) {
    const uint32_t lo = ww.seg[J >> 2], hi = (J >> 2) + 1 < 16 ? ww.seg[((J >> 2) + 1) & 15] : ww.bnext;
    return (J & 3) ? __funnelshift_r(lo, hi, (J & 3) * 8) : lo;
}

// Every lane goes through its 64 bytes from the last to the first: a position whose successor lies among the same 64
// bytes takes over the successor's value, so that afterwards ex[] says of every position where its chain leaves the
// lane's 64 bytes (RLE_NEXT_FAIL: at a header that is none).
template <int J>
__device__ __forceinline__ void rle_wave_segment_exits_from(const RleWaveWindow &ww, uint32_t lane, uint16_t *ex) {
    const uint32_t seg_end = lane * 64u + 64u;
    const uint32_t p = lane * 64u + (uint32_t)J;
    const RleLaneHeader h = rle_lane_header(ww, p, rle_lane_bytes<J>(ww));
    const uint32_t inside = ex[rle_exit_slot(h.to < seg_end ? h.to : seg_end - 1)];
    const uint32_t nx = !h.plain ? RLE_NEXT_FAIL : (h.to < seg_end ? inside : (h.to >= RLE_NEXT_FAR ? RLE_NEXT_FAR : h.to));
    ex[rle_exit_slot(p)] = (uint16_t)nx;
    if constexpr (J > 0) rle_wave_segment_exits_from<J - 1>(ww, lane, ex);
}
__device__ __forceinline__ void rle_wave_segment_exits(const RleWaveWindow &ww, uint32_t lane, uint16_t *ex) {
    rle_wave_segment_exits_from<63>(ww, lane, ex);
}

template <int KIND, int ELEM>
__global__ __launch_bounds__(RLE_EXIT_THREADS) void k_rle_split_exits(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a) {
    static_assert(RLE_WIN == 64 * RLE_EXIT_THREADS, "a lane per 64 bytes of the window");
    __shared__ __align__(16) uint16_t ex[RLE_WIN + 2 * (RLE_WIN / 64)];
    const uint32_t widx = blockIdx.x;
    if (widx >= *a.total_windows) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t page = a.win_page[widx];
    const RleSplitPage sp = a.pages[page];
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    RleWaveWindow ww;
    rle_wave_window_load<KIND, ELEM>(ww, su, sp, widx - sp.win0, lane);
    rle_wave_segment_exits(ww, lane, ex);
    __syncthreads();
    for (int s = RLE_WIN / 64 - 2; s >= 0; --s) {
        const uint32_t p = (uint32_t)s * 64u + lane;
        uint32_t e = ex[rle_exit_slot(p)];
        if (e < ww.lim) {
            e = ex[rle_exit_slot(e)];
            ex[rle_exit_slot(p)] = (uint16_t)e;
        }
    }
    __syncthreads();
    uint32_t *dst = reinterpret_cast<uint32_t *>(a.exits + (uint64_t)widx * RLE_WIN);
    for (uint32_t i = lane; i < RLE_WIN / 2; i += RLE_EXIT_THREADS)
        dst[i] = *reinterpret_cast<const uint32_t *>(&ex[rle_exit_slot(2 * i)]);
}

// Per page (one wave; one lane follows the chain, all lanes write the entries out): the chain through the windows.
// Every step is one dependent table lookup, so nothing else is in the loop: the entries are collected in LDS and
// written out in coalesced pieces.
constexpr uint32_t RLE_ENTRY_CHUNK = 2048;
template <int KIND>
__global__ __launch_bounds__(64) void k_rle_split_entry(const uint8_t *__restrict__ d_in, uint64_t in_bytes,
                                                        const p4s_page_desc *__restrict__ pages, uint32_t n_pages, RleSplitArrays a) {
    __shared__ uint32_t buf[RLE_ENTRY_CHUNK];
    __shared__ uint32_t s_k, s_e, s_far;
    const uint32_t page = blockIdx.x, lane = threadIdx.x;
    if (page >= n_pages) return;
    RleSplitPage &sp = a.pages[page];
    const uint32_t nwin = sp.nwin, win0 = sp.win0, mis = sp.mis;
    if (nwin == 0) return;
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    const uint64_t size = su.pv.size;
    if (lane == 0) { s_k = 0; s_e = mis; s_far = 0; }
    for (uint32_t k0 = 0; k0 < nwin; k0 += RLE_ENTRY_CHUNK) {
        const uint32_t k1 = k0 + RLE_ENTRY_CHUNK < nwin ? k0 + RLE_ENTRY_CHUNK : nwin;
        for (uint32_t i = lane; i < RLE_ENTRY_CHUNK; i += 64) buf[i] = RLE_NO_ENTRY;
        __syncthreads();
        if (lane == 0) {
            uint64_t k = s_k;            // window the chain enters next (nwin or more: it has ended)
            uint32_t e = s_e;
            while (k < k1) {
                buf[k - k0] = e;
                const uint32_t x = a.exits[(win0 + k) * (uint64_t)RLE_WIN + e];
                if (x == RLE_NEXT_FAIL) { k = ~0u; break; }              // the count pass meets the same header and flags the window
                if (x == RLE_NEXT_FAR) { s_far = 1; k = ~0u; break; }    // where the chain goes on is not in the table
                const uint64_t at = k * RLE_WIN + x;                       // window coordinate of the next header
                if (at - mis >= size) { k = ~0u; break; }                // the chain ends with the stream (or overshoots it: flagged by the count pass)
                k = at / RLE_WIN;
                e = (uint32_t)(at % RLE_WIN);
            }
            s_k = k > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)k;
            s_e = e;
        }
        __syncthreads();
        for (uint32_t i = lane; i < k1 - k0; i += 64) a.entry[win0 + k0 + i] = buf[i];
        __syncthreads();
    }
    if (lane == 0 && s_far) sp.pad = 1;       // decode the page whole
}

// Per window (one lane): the values of the runs whose headers lie in the window, on the true chain, and whether the
// chain met anything but a plain run that the stream holds completely (such a page is decoded whole).  The first
// header in every 256 bytes of the window is kept with the values before it: a part starts at the last of these at
// or before its first value (a window of many short runs would otherwise be walked from its start by every part).
template <int KIND, int ELEM>
__global__ __launch_bounds__(256) void k_rle_split_count(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a,
        uint32_t leave_narrow) {
    const uint32_t widx = blockIdx.x * blockDim.x + threadIdx.x;
    if (widx >= *a.total_windows) return;
    if (leave_narrow && a.pages[a.win_page[widx]].narrow) return;       // k_rle_split_count_wave's
    uint64_t count = 0;
    uint32_t flag = 0, slot = 0;                // slot: checkpoints written so far
    uint16_t *ck_pos = a.ck_pos + (uint64_t)widx * RLE_CKS;
    uint32_t *ck_cnt = a.ck_cnt + (uint64_t)widx * RLE_CKS;
    const uint32_t e = a.entry[widx];
    if (e != RLE_NO_ENTRY) {
        const uint32_t page = a.win_page[widx];
        const RleSplitPage sp = a.pages[page];
        const uint32_t k = widx - sp.win0;
        const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
        const uint32_t tbytes = KIND == RLE_KIND_DICT ? 4u : (uint32_t)ELEM;
        const uint32_t w = su.w, vbytes = (w + 7) >> 3;
        const int64_t wstart = (int64_t)k * RLE_WIN - sp.mis;
        int64_t pos = wstart + e;
        const int64_t wend = wstart + RLE_WIN;
        const int64_t end = wend < (int64_t)su.pv.size ? wend : (int64_t)su.pv.size;
        // every load is a round trip to the L2: 16 bytes at a time, and all the headers that start among the first 13
        while (pos < end && !flag) {
            uint64_t lo = 0, hi = 0;
            const bool many = pos + 16 <= (int64_t)su.pv.room;
            if (many) {
                lo = P4S_LD(reinterpret_cast<const uint64_t *>(su.pv.base + pos), 113);
                hi = P4S_LD(reinterpret_cast<const uint64_t *>(su.pv.base + pos + 8), 113);
            } else {
                for (int i = 0; i < 4; ++i)
                    if (pos + i < (int64_t)su.pv.size) lo |= (uint64_t)P4S_LD(su.pv.base + pos + i, 114) << (8 * i);
            }
            uint32_t o = 0;
            do {
                const int64_t avail = (int64_t)su.pv.size - pos;
                const uint32_t x = o < 8 ? (uint32_t)(lo >> (8 * o)) | (o > 4 ? (uint32_t)(hi << (64 - 8 * o)) : 0u)
                                         : (uint32_t)(hi >> (8 * (o - 8)));
                const RleCheapHeader h = rle_parse_cheap(x, avail, w, vbytes, tbytes);
                if (!h.plain || (int64_t)h.skip > avail) { flag = 1; break; }
                const uint32_t off = (uint32_t)(pos - wstart), at = off >> 8;
                if (at >= slot) {
                    for (; slot < at; ++slot) ck_pos[slot] = (uint16_t)RLE_CK_NONE;
                    ck_pos[slot] = (uint16_t)(count >> 32 ? RLE_CK_NONE : off);
                    ck_cnt[slot] = (uint32_t)count;
                    ++slot;
                }
                count += h.vals;
                pos += h.skip;
                o += h.skip;
            } while (many && o <= 12 && pos < end);
        }
    }
    for (; slot < RLE_CKS; ++slot) ck_pos[slot] = (uint16_t)RLE_CK_NONE;
    a.wcount[widx] = count;
    a.wflag[widx] = flag;
}

template <int J>
__device__ __forceinline__ void rle_wave_segment_count(const RleWaveWindow &ww, uint32_t lane, uint32_t &cur, uint32_t &cnt, uint32_t &flag) {
    const uint32_t p = lane * 64u + (uint32_t)J;
    const RleLaneHeader h = rle_lane_header(ww, p, rle_lane_bytes<J>(ww));
    const bool here = cur == p && p < ww.lim;            // the chain ends where the stream does
    const bool good = h.plain && h.to <= ww.left;        // the stream holds the whole run
    flag |= (here && !good) ? 1u : 0u;
    cnt += (here && good) ? h.vals : 0u;
    cur = here ? (good ? h.to : RLE_CK_NONE) : cur;
    if constexpr (J < 63) rle_wave_segment_count<J + 1>(ww, lane, cur, cnt, flag);
}

// The same per window by one wave, for batches of few windows and for the windows of pages of narrow values (a
// window of a thousand short runs keeps a single lane busy for a quarter of a millisecond): where the chain leaves
// every 64 bytes (as k_rle_split_exits does), lane 0 follows it from 64 bytes to 64 bytes, and every lane adds up the
// runs that start in its own.
template <int KIND, int ELEM>
__global__ __launch_bounds__(RLE_EXIT_THREADS) void k_rle_split_count_wave(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a,
        uint32_t only_narrow) {
    __shared__ __align__(16) uint16_t ex[RLE_WIN + 2 * (RLE_WIN / 64)];
    __shared__ uint16_t s_ent[64];
    const uint32_t widx = blockIdx.x;
    if (widx >= *a.total_windows) return;
    const uint32_t lane = threadIdx.x;
    const uint32_t page = a.win_page[widx];
    const RleSplitPage sp = a.pages[page];
    if (only_narrow && !sp.narrow) return;                               // k_rle_split_count's
    uint16_t *ck_pos = a.ck_pos + (uint64_t)widx * RLE_CKS;
    uint32_t *ck_cnt = a.ck_cnt + (uint64_t)widx * RLE_CKS;
    const uint32_t e = a.entry[widx];
    if (e == RLE_NO_ENTRY) {
        if (lane < RLE_CKS) ck_pos[lane] = (uint16_t)RLE_CK_NONE;
        if (lane == 0) { a.wcount[widx] = 0; a.wflag[widx] = 0; }
        return;
    }
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    RleWaveWindow ww;
    rle_wave_window_load<KIND, ELEM>(ww, su, sp, widx - sp.win0, lane);
    rle_wave_segment_exits(ww, lane, ex);
    s_ent[lane] = (uint16_t)RLE_CK_NONE;
    __syncthreads();
    if (lane == 0) {
        uint32_t pos = e;
        while (pos < ww.lim) {
            s_ent[pos >> 6] = (uint16_t)pos;
            pos = ex[rle_exit_slot(pos)];
        }
    }
    __syncthreads();
    // the runs of the chain that start among this lane's 64 bytes (at most 32 of at most 2^23 values)
    uint32_t cur = s_ent[lane], cnt = 0, flag = 0;
    const uint32_t mine = cur;
    rle_wave_segment_count<0>(ww, lane, cur, cnt, flag);
    const uint64_t incl = wave_iscan((uint64_t)cnt);
    // a part can start at the first header of every 256 bytes
    const unsigned long long entered = __ballot(mine != RLE_CK_NONE);
    const uint32_t four = lane < RLE_CKS ? (uint32_t)(entered >> (4 * lane)) & 0xFu : 0u;
    const uint32_t src = four ? 4 * lane + (uint32_t)__builtin_ctz(four) : 0u;
    const uint64_t before = incl - cnt;
    const uint32_t ck_p = __shfl(mine, src), ck_c = __shfl((uint32_t)before, src), ck_h = __shfl((uint32_t)(before >> 32), src);
    if (lane < RLE_CKS) {
        ck_pos[lane] = (uint16_t)(four && ck_h == 0 ? ck_p : RLE_CK_NONE);
        ck_cnt[lane] = ck_c;
    }
    const unsigned long long flagged = __ballot(flag != 0);
    if (lane == 63) {
        a.wcount[widx] = incl;
        a.wflag[widx] = flagged ? 1u : 0u;
    }
}

// Per page (one workgroup): first output index of every window; the parts.
template <int KIND>
__global__ __launch_bounds__(RLE_THREADS) void k_rle_split_plan(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a,
        uint32_t fan) {
    __shared__ uint64_t wsum[4];
    __shared__ uint64_t carry;
    __shared__ uint32_t bad;
    const uint32_t page = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    RleSplitPage &sp = a.pages[page];
    const uint32_t nwin = sp.nwin, win0 = sp.win0;
    if (nwin == 0 || sp.pad) return;
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    const uint64_t n = su.n;
    if (tid == 0) { carry = 0; bad = 0; }
    __syncthreads();
    for (uint32_t k0 = 0; k0 < nwin; k0 += RLE_THREADS) {
        const uint32_t k = k0 + tid;
        const uint64_t c = k < nwin ? a.wcount[win0 + k] : 0;
        const uint64_t inc = wave_iscan(c);
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint64_t before = carry + inc - c;
        for (uint32_t i = 0; i < wave; ++i) before += wsum[i];
        if (k < nwin) {
            a.wbase[win0 + k] = before;
            if (before < n && a.wflag[win0 + k]) bad = 1;      // something irregular before the n-th value
        }
        __syncthreads();
        if (tid == RLE_THREADS - 1) carry = before + c;
        __syncthreads();
    }
    if (bad || carry < n) return;                               // decoded whole: nparts stays 0
    // parts of equal output size (whole tiles of the expander), at most `fan` of them
    uint64_t T = (n + fan - 1) / fan;
    T = (T + 511) / 512 * 512;
    if (T < 4096) T = 4096;
    const uint32_t nparts = (uint32_t)((n + T - 1) / T);
    for (uint32_t s = tid; s < nparts; s += RLE_THREADS) {
        const uint64_t X = (uint64_t)s * T;
        // last window whose first run starts at or before value X ...
        uint32_t lo = 0, hi = nwin;                             // wbase[lo] <= X < wbase[hi] (wbase[nwin] = total > X)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (a.wbase[win0 + mid] <= X) lo = mid; else hi = mid;
        }
        // ... that the chain enters (a long run skips whole windows)
        while (lo > 0 && a.entry[win0 + lo] == RLE_NO_ENTRY) --lo;
        // ... and in it the last kept header at or before value X
        uint32_t from = a.entry[win0 + lo], before = 0;
        const uint64_t rel = X - a.wbase[win0 + lo];
        for (uint32_t j = 0; j < RLE_CKS; ++j) {
            const uint32_t cp = a.ck_pos[(uint64_t)(win0 + lo) * RLE_CKS + j], cc = a.ck_cnt[(uint64_t)(win0 + lo) * RLE_CKS + j];
            if (cp != RLE_CK_NONE && cc <= rel) { from = cp; before = cc; }
        }
        RlePart pt;
        pt.span.cur0 = (uint64_t)lo * RLE_WIN + from - sp.mis;
        pt.span.out0 = a.wbase[win0 + lo] + before;
        pt.span.out_begin = X;
        pt.span.in_limit = ~0ull;
        pt.n = X + T < n ? X + T : n;
        a.parts[(uint64_t)page * fan + s] = pt;
    }
    __syncthreads();
    if (tid == 0) sp.nparts = nparts;
}

// Statuses of the pages that were decoded in parts (every part found all of its values: that is what made the
// page eligible), with what the parts counted.
template <int KIND>
__global__ void k_rle_split_finish(const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
                                   uint32_t n_pages, uint64_t dict_size, RleSplitArrays a, p4s_page_status *__restrict__ status) {
    const uint32_t page = blockIdx.x * blockDim.x + threadIdx.x;
    if (page >= n_pages) return;
    const RleSplitPage sp = a.pages[page];
    if (sp.nparts == 0) return;
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    if constexpr (KIND == RLE_KIND_DICT) {
        if (sp.errkey != ~0ull) { write_status(&status[page], P4S_E_DICT_INDEX, 0, dict_size, sp.errkey & 0xFFFFFFFFull, 0, 0); return; }
        write_status(&status[page], P4S_E_NONE, su.n, 0, 0, 0, 0);
    } else if constexpr (KIND == RLE_KIND_LEVELS) {
        write_status(&status[page], P4S_E_NONE, su.n, sp.hits, sp.above, su.consumed, 0);
    } else {
        write_status(&status[page], P4S_E_NONE, su.n, 0, 0, 0, 0);
    }
}

}  // namespace p4s
