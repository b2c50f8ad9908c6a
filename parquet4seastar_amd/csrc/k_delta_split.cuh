// DELTA_BINARY_PACKED pages split over several workgroups (small batches: with one workgroup per page a batch of
// few pages leaves most of the chip idle and a page takes the latency of its 512-block chain plus its decode).
//
//   k_delta_split_walk    one workgroup per page: the block-header walk of the decode kernel without the decode
//                         (windows through LDS, wave 0 follows the chain of block headers).  Every B-th block start
//                         becomes the start of a part.  Pages the fast chain cannot walk to their last whole block
//                         (odd shapes, errors, reads that stop before the stream's end) are not split.
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

template <typename T>
__global__ __launch_bounds__(DELTA_THREADS, 8) void k_delta_split_walk(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        DeltaPartRec *__restrict__ parts, DeltaSplitPage *__restrict__ split, uint32_t fan) {
    __shared__ DeltaShared sh;
    __shared__ uint32_t s_bad;
    const uint32_t page = blockIdx.x;
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
        int64_t wlo = window_load<DELTA_WIN, DELTA_THREADS>(sh.win, pv, p);
        if (tid < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DELTA_WIN)[tid] = 0;
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
            wlo = window_load<DELTA_WIN, DELTA_THREADS>(sh.win, pv, next);
            if (tid < DELTA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + DELTA_WIN)[tid] = 0;
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
