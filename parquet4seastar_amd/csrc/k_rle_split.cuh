// Splitting hybrid-RLE pages over many workgroups, for batches with too few pages to fill the chip with one
// workgroup per page (one page of 1Mi values otherwise takes milliseconds: its run-header chain is one long
// dependency, RleDecoder::NextCounts of include/parquet4seastar/rle_encoding.hh:321-348, walked by one wave).
//
// The only serial fact of the format is where the next run header starts.  The passes below resolve that chain
// in parallel and cut every page into parts that k_rle decodes like pages of their own:
//   k_rle_split_init   per page: the stream (rle_setup) and how many 4 KiB windows it spans;
//   k_rle_split_jump   per window: every byte position is parsed as if a header started there (next[p] = where
//                      the header after it would start), then pointer jumping (next[p] = next[next[p]], log steps)
//                      turns next[p] into "where the chain that enters the window at p leaves it", for all p at once;
//   k_rle_split_entry  per page: the true chain through the windows, one table lookup per window;
//   k_rle_split_count  per window: the values the runs that start in the window produce (the decode kernel's own
//                      header walk and run placement, nothing stored), and whether anything irregular was met;
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
    RlePart *parts;           // [n_pages * fan]
    uint32_t cap;             // windows the arrays hold
};

constexpr uint32_t RLE_SPLIT_MAX_PAGES = 1024;   // batches of more pages fill the chip with one workgroup per page

// One workgroup: windows per page and their place in the per-window arrays (n_pages <= RLE_SPLIT_MAX_PAGES).
template <int KIND>
__global__ __launch_bounds__(RLE_THREADS) void k_rle_split_init(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, uint32_t n_pages,
        RleSplitArrays a) {
    __shared__ uint32_t s_nwin[RLE_SPLIT_MAX_PAGES], s_win0[RLE_SPLIT_MAX_PAGES];
    __shared__ uint8_t s_mis[RLE_SPLIT_MAX_PAGES];
    const uint32_t tid = threadIdx.x;
    for (uint32_t p = tid; p < n_pages; p += RLE_THREADS) {
        uint32_t nwin = 0, mis = 0;
        const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[p]);
        if (su.mode == 0 && su.n > 0 && su.pv.size > 0) {
            mis = (uint32_t)((uintptr_t)su.pv.base & 15u);
            const uint64_t w = (su.pv.size + mis + RLE_WIN - 1) / RLE_WIN;
            nwin = w > a.cap ? a.cap + 1u : (uint32_t)w;
        }
        s_nwin[p] = nwin;
        s_mis[p] = (uint8_t)mis;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t pos = 0;
        for (uint32_t p = 0; p < n_pages; ++p) {
            // a page whose windows do not fit the arrays is decoded whole
            if ((uint64_t)pos + s_nwin[p] > a.cap) s_nwin[p] = 0;
            s_win0[p] = pos;
            pos += s_nwin[p];
        }
        *a.total_windows = pos;
    }
    __syncthreads();
    for (uint32_t p = 0; p < n_pages; ++p) {
        const uint32_t nwin = s_nwin[p], win0 = s_win0[p];
        if (tid == 0) {
            RleSplitPage sp;
            sp.errkey = ~0ull; sp.hits = 0; sp.above = 0; sp.win0 = win0; sp.nwin = nwin; sp.nparts = 0; sp.mis = s_mis[p];
            sp.lim_last = 0; sp.pad = 0;
            a.pages[p] = sp;
        }
        for (uint32_t k = tid; k < nwin; k += RLE_THREADS) a.win_page[win0 + k] = p;
    }
}

// Per window: exits[p] = where the chain that enters the window at offset p leaves it.
template <int KIND, int ELEM>
__global__ __launch_bounds__(RLE_THREADS) void k_rle_split_jump(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a) {
    __shared__ __align__(16) uint8_t win[RLE_WIN + RLE_SLACK];
    __shared__ __align__(16) uint16_t nxt[RLE_WIN];
    const uint32_t widx = blockIdx.x;
    if (widx >= *a.total_windows) return;
    const uint32_t tid = threadIdx.x;
    const uint32_t page = a.win_page[widx];
    const RleSplitPage sp = a.pages[page];
    const uint32_t k = widx - sp.win0;
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    const uint32_t tbytes = KIND == RLE_KIND_DICT ? 4u : (uint32_t)ELEM;
    const uint32_t w = su.w, vbytes = (w + 7) >> 3;
    const int64_t so = (int64_t)k * RLE_WIN - (int64_t)sp.mis;          // stream offset of the window's byte 0
    // the window (16-byte aligned global address), zero past the input allocation
    {
        const int64_t o = so + (int64_t)tid * 16;
        const int64_t room = (int64_t)su.pv.room - o;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (room >= 16) {
            v = P4S_LD(reinterpret_cast<const uint4 *>(su.pv.base + o), 110);
        } else if (room > 0) {
            uint8_t tmp[16];
            for (int i = 0; i < 16; ++i) tmp[i] = i < room ? P4S_LD(su.pv.base + o + i, 111) : 0;
            v = *reinterpret_cast<uint4 *>(tmp);
        }
        *reinterpret_cast<uint4 *>(win + tid * 16) = v;
        if (tid == 0) {
            // the bytes a header at the window's last offsets runs into
            uint8_t tmp[16];
            const int64_t o2 = so + RLE_WIN, room2 = (int64_t)su.pv.room - o2;
            for (int i = 0; i < 16; ++i) tmp[i] = i < room2 ? P4S_LD(su.pv.base + o2 + i, 112) : 0;
            *reinterpret_cast<uint4 *>(win + RLE_WIN) = *reinterpret_cast<uint4 *>(tmp);
        }
    }
    __syncthreads();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(win);
    // offsets at or past `lim` lie outside the window or the stream: a chain that reaches one has left
    const int64_t left = (int64_t)su.pv.size - so;
    const uint32_t lim = left >= (int64_t)RLE_WIN ? (uint32_t)RLE_WIN : (left > 0 ? (uint32_t)left : 0u);
    constexpr int PER = RLE_WIN / RLE_THREADS;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const uint32_t p = tid + (uint32_t)i * RLE_THREADS;
        const int64_t sp_off = so + (int64_t)p;
        uint32_t nx = RLE_NEXT_FAIL;
        if (sp_off >= 0 && p < lim) {
            const RleHeader h = rle_parse(rle_win40(win32, p), rle_avail(su.pv, sp_off), w, vbytes, tbytes);
            if (!h.fail) {
                const uint64_t to = (uint64_t)p + h.skip;
                nx = to >= RLE_NEXT_FAR ? RLE_NEXT_FAR : (uint32_t)to;
            }
        }
        nxt[p] = (uint16_t)nx;
    }
    __syncthreads();
    // pointer jumping, in place: next[p] only ever moves further along p's own chain, so reading an entry that
    // another thread has already advanced is as good as reading the old one; 12 rounds cover 4096 steps
    for (int round = 0; round < 12; ++round) {
        bool changed = false;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t p = tid + (uint32_t)i * RLE_THREADS;
            const uint32_t n1 = nxt[p];
            if (n1 < lim) {
                nxt[p] = nxt[n1];
                changed = true;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    __syncthreads();
    uint4 *dst = reinterpret_cast<uint4 *>(a.exits + (uint64_t)widx * RLE_WIN);
    const uint4 *src = reinterpret_cast<const uint4 *>(nxt);
    for (uint32_t c = tid; c < RLE_WIN * 2 / 16; c += RLE_THREADS) dst[c] = src[c];
    if (tid == 0) {
        a.wcount[widx] = 0;
        a.wflag[widx] = 0;
    }
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

struct RleSinkNone {
    static constexpr bool COUNT_ONLY = true;
    template <typename SH>
    __device__ __forceinline__ void store_vals(uint64_t, uint32_t, uint32_t, const uint32_t[RLE_VPT], SH &) const {}
};

// Per window: values of the runs whose headers lie in the window, on the true chain.
template <int KIND, int ELEM>
__global__ __launch_bounds__(RLE_THREADS, 8) void k_rle_split_count(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages, RleSplitArrays a) {
    __shared__ RleShared sh;
    const uint32_t widx = blockIdx.x;
    if (widx >= *a.total_windows) return;
    const uint32_t e = a.entry[widx];
    if (e == RLE_NO_ENTRY) return;
    const uint32_t page = a.win_page[widx];
    const RleSplitPage sp = a.pages[page];
    const uint32_t k = widx - sp.win0;
    const RleSetup su = rle_setup<KIND>(d_in, in_bytes, pages[page]);
    const uint32_t tbytes = KIND == RLE_KIND_DICT ? 4u : (uint32_t)ELEM;
    RleSpan span;
    span.cur0 = (uint64_t)k * RLE_WIN + e - sp.mis;
    span.out0 = 0;
    span.out_begin = 0;
    span.in_limit = (uint64_t)(k + 1) * RLE_WIN - sp.mis;
    RleSinkNone sink;
    // a count that reaches 2^32 - 1 already covers any n a page can be asked for
    const uint64_t got = rle_decode_stream_ring(sh, su.pv, 0xFFFFFFFFull, su.w, tbytes, 0, sink, false, 0, span);
    if (threadIdx.x == 0) {
        a.wcount[widx] = got;
        a.wflag[widx] = sh.irregular;
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
        RlePart pt;
        pt.span.cur0 = (uint64_t)lo * RLE_WIN + a.entry[win0 + lo] - sp.mis;
        pt.span.out0 = a.wbase[win0 + lo];
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
