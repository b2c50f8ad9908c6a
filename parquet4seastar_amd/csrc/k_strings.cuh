// BYTE_ARRAY decoders producing (offsets, bytes):
//   DELTA_LENGTH_BYTE_ARRAY  (ref: src/encoding.cc:252-294)
//   DELTA_BYTE_ARRAY         (ref: src/encoding.cc:296-362, single whole-page read_batch)
//   PLAIN BYTE_ARRAY         (ref: src/encoding.cc:405-408,430-450)
//   PLAIN FIXED_LEN_BYTE_ARRAY (ref: src/encoding.cc:410-413,452-466)
// One 256-thread workgroup per page.  Length streams reuse the DELTA_BINARY_PACKED
// machinery (k_delta.cuh) with sinks that turn lengths into offsets by a second
// prefix sum in the same tile.
#pragma once
#include "k_delta.cuh"

namespace p4s {

// PLAIN BYTE_ARRAY walks its length chain through an LDS window of its own
constexpr int PBA_THREADS = 256;
constexpr int PBA_WIN = 4096;
constexpr int PBA_SLACK = 16;
constexpr int PBA_MAXSTR = 512;

// Cooperative byte copy by the whole workgroup; dst/src arbitrary alignment.
__device__ __forceinline__ void wg_copy(uint8_t *dst, const uint8_t *src, uint64_t n) {
    const uint32_t tid = threadIdx.x;
    uint64_t head = (16 - ((uintptr_t)dst & 15u)) & 15u;
    if (head > n) head = n;
    for (uint64_t i = tid; i < head; i += blockDim.x) P4S_ST(dst + i, P4S_LD(src + i, 20), 21);
    dst += head; src += head; n -= head;
    const uint64_t n16 = n >> 4;
    uint64_t i = tid;
    const uint64_t step = blockDim.x;
#ifndef P4S_DEBUG_BOUNDS
    // eight pieces in flight per thread: with one, every load also waits for the store before it (loads and stores
    // share the wave's memory counter, in issue order) and a wave moves 1 KiB per round trip
    constexpr int U = 8;
    for (; i + (U - 1) * step < n16; i += U * step) {
        uint4 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) __builtin_memcpy(&v[k], src + ((i + k * step) << 4), 16);
#pragma unroll
        for (int k = 0; k < U; ++k) *reinterpret_cast<uint4 *>(dst + ((i + k * step) << 4)) = v[k];
    }
#endif
    for (; i < n16; i += step) {
        uint4 v;
#ifdef P4S_DEBUG_BOUNDS
        if (!dbg_ok(src + (i << 4), 16, 22) || !dbg_ok(dst + (i << 4), 16, 23)) continue;
#endif
        __builtin_memcpy(&v, src + (i << 4), 16);
        *reinterpret_cast<uint4 *>(dst + (i << 4)) = v;
    }
    for (uint64_t i = (n16 << 4) + tid; i < n; i += blockDim.x) P4S_ST(dst + i, P4S_LD(src + i, 24), 25);
}

// Workgroup exclusive prefix over one uint64 per thread (4 waves), with a running carry.
struct WgScan64 {
    uint64_t wtot[2][4];
};

__device__ __forceinline__ uint64_t wg_exclusive_scan(WgScan64 &s, uint32_t &buf, uint64_t x, uint64_t &total) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint64_t inc = wave_iscan(x);
    if (lane == 63) s.wtot[buf][wave] = inc;
    __syncthreads();
    uint64_t base = inc - x;
    total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t t = s.wtot[buf][i];
        if ((uint32_t)i < wave) base += t;
        total += t;
    }
    buf ^= 1u;
    return base;
}

// ---------------------------------------------------------------------------
// DELTA_LENGTH_BYTE_ARRAY
// ---------------------------------------------------------------------------
struct DlbaSink {
    int64_t *offsets;     // page slot: offsets[0] = 0
    uint64_t limit;       // strings to emit
    WgScan64 *scan;
    uint32_t buf;
    uint64_t carry;       // bytes of the strings emitted so far
    __device__ __forceinline__ void first(uint32_t v) {
        if (limit > 0) {
            if (threadIdx.x == 0) { P4S_ST(offsets, 0, 26); P4S_ST(offsets + 1, (int64_t)(uint64_t)v, 27); }
            carry = v;
        } else if (threadIdx.x == 0) {
            P4S_ST(offsets, 0, 28);
        }
    }
    static constexpr bool PREFETCH_WINDOW = true;
    static constexpr bool SHIFTED_STORES = false;
    __device__ __forceinline__ bool full_ok(uint64_t, uint32_t) const { return true; }
    __device__ __forceinline__ void tile_full(uint64_t base, uint32_t lane_off, const uint32_t (&r)[4]) { tile(base + lane_off, r, 4u); }
    __device__ __forceinline__ void tile(uint64_t idx0, const uint32_t (&r)[4], uint32_t valid) {
        uint64_t l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) l[j] = ((uint32_t)j < valid && idx0 + j < limit) ? (uint64_t)r[j] : 0;   // (uint32_t)len, ref :262
        l[1] += l[0]; l[2] += l[1]; l[3] += l[2];
        uint64_t total;
        const uint64_t base = carry + wg_exclusive_scan(*scan, buf, l[3], total);
        if (valid == 4u && idx0 + 4 <= limit) {
            // a lane's four offsets as two 16-byte stores (element by element they leave as four 8-byte stores, each
            // of which touches every 32 bytes of the tile's 2 KiB)
            typedef uint32_t v4u32_ __attribute__((ext_vector_type(4), aligned(8)));
            const uint64_t o[4] = {base + l[0], base + l[1], base + l[2], base + l[3]};
            v4u32_ t[2];
            __builtin_memcpy(t, o, 32);
            if (P4S_INB(offsets + idx0 + 1, 32, 29)) {
                v4u32_ *q = reinterpret_cast<v4u32_ *>(offsets + idx0 + 1);
                q[0] = t[0];
                q[1] = t[1];
            }
        } else {
            // (three at most: see above for why the fourth must not appear in this loop)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                if ((uint32_t)j < valid && idx0 + j < limit) P4S_ST(offsets + idx0 + j + 1, (int64_t)(base + l[j]), 29);
        }
        carry += total;
    }
};

struct StringsShared {
    DeltaShared delta;
    WgScan64 scan;
    WgScan64 scan2;
    int32_t err;
};

#ifndef P4S_DLBA_MIN_WG
#define P4S_DLBA_MIN_WG 1     // workgroups per CU the register budget is sized for
#endif

__global__ __launch_bounds__(DELTA_THREADS, P4S_DLBA_MIN_WG) void k_delta_length_byte_array(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        int64_t *__restrict__ d_offsets, uint8_t *__restrict__ d_bytes, uint64_t bytes_cap,
        p4s_page_status *__restrict__ status) {
    __shared__ StringsShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    DlbaSink sink{d_offsets + pd.out_offset, 0, &sh.scan, 0, 0};
    // the sink may only emit min(n, total) strings; total is in the stream header, which the decode parses
    // first, so give it n and let the decode's own count cap the indices it hands over
    sink.limit = pd.num_values;
    DeltaResult res;
    delta_decode_stream<uint32_t>(sh.delta, pv, pd.num_values, true, sink, res);
    const uint32_t tid = threadIdx.x;
    if (res.err) {
        if (tid == 0) write_status(&status[page], res.err, 0, res.aux0, res.aux1, 0, 0);
        return;
    }
    const uint64_t count = res.count;
    if (count == 0 && tid == 0) d_offsets[pd.out_offset] = 0;
    const uint64_t data_pos = res.end_pos;                      // ref :289-290
    const uint64_t data_len = pv.size - data_pos;
    const uint64_t nbytes = sink.carry;
    if (nbytes > data_len) {                                    // ref :262-266
        if (tid == 0) write_status(&status[page], P4S_E_DLBA_VALUES_EOF, 0, 0, 0, 0, 0);
        return;
    }
    const uint64_t cap = pd.param0;
    if (nbytes > cap || pd.aux_offset + nbytes > bytes_cap) {
        if (tid == 0) write_status(&status[page], E_OUT_CAPACITY, count, nbytes, cap, 0, nbytes);
        return;
    }
    wg_copy(d_bytes + pd.aux_offset, pv.base + data_pos, nbytes);
    if (tid == 0) write_status(&status[page], P4S_E_NONE, count, 0, 0, data_pos + nbytes, nbytes);
}

// ---------------------------------------------------------------------------
// DELTA_BYTE_ARRAY: sinks of the two nested length streams (the kernels are in k_dba.cuh)
// ---------------------------------------------------------------------------
// scratch: two uint32 per value slot (same slot numbering as d_offsets), as two arrays of scratch_slots words: the
// prefix lengths, then the suffix lengths / offsets inside the suffix data section.  The prefix lengths are stored by the
// plain decoder's sink (DeltaStoreSink<uint32_t>).

constexpr uint32_t DBA_TILE_SHIFT = 8;      // the string assembly works on tiles of 256 strings

// For a walk to a stream's end (delta_decode_stream<.., WALK_ONLY>): nothing is handed over but the first value.
struct DbaNullSink {
    static constexpr bool PREFETCH_WINDOW = true;
    static constexpr bool SHIFTED_STORES = false;
    __device__ __forceinline__ void first(uint32_t) const {}
    __device__ __forceinline__ bool full_ok(uint64_t, uint32_t) const { return true; }
    __device__ __forceinline__ void tile_full(uint64_t, uint32_t, const uint32_t (&)[4]) const {}
    __device__ __forceinline__ void tile(uint64_t, const uint32_t (&)[4], uint32_t) const {}
};

// Suffix lengths: kept as they come (the offsets are made of them afterwards by dba_scan_page), and the bytes of ALL
// suffixes added up (the reference slices every one of them in reset()).
struct DbaSuffixSink {
    uint32_t *out;        // suffix length of string i -> out[i]
    uint64_t limit;       // strings to emit = min(n, number of suffixes)
    uint64_t all_suffix;  // this thread's share of the sum
    __device__ __forceinline__ void first(uint32_t v) {
        if (threadIdx.x == 0) {
            all_suffix += v;
            if (limit > 0) P4S_ST(out, v, 13);
        }
    }
    static constexpr bool PREFETCH_WINDOW = true;
    static constexpr bool SHIFTED_STORES = false;      // (every value has to pass exactly once: they are added up)
    __device__ __forceinline__ bool full_ok(uint64_t base, uint32_t n) const { return base + n <= limit; }
    __device__ __forceinline__ void tile_full(uint64_t base, uint32_t lane_off, const uint32_t (&r)[4]) {
        all_suffix += (uint64_t)r[0] + r[1] + r[2] + r[3];
        if (P4S_INB(out + base + lane_off, 16, 18)) __builtin_memcpy(out + base + lane_off, r, 16);
    }
    __device__ __forceinline__ void tile(uint64_t idx0, const uint32_t (&r)[4], uint32_t valid) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((uint32_t)j < valid) {
                all_suffix += r[j];
                if (idx0 + j < limit) P4S_ST(out + idx0 + j, r[j], 18);
            }
        }
    }
};

// ---------------------------------------------------------------------------
// PLAIN FIXED_LEN_BYTE_ARRAY
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_plain_flba(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_bytes, uint64_t bytes_cap, p4s_page_status *__restrict__ status) {
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint64_t L = pd.param0;
    const uint64_t n = pd.num_values;
    uint64_t count;
    if (L == 0) {
        count = pv.size == 0 ? 0 : n;                            // ref :452-466 with zero-length slices
    } else {
        const uint64_t whole = pv.size / L;
        if (n <= whole) count = n;
        else if (pv.size % L == 0) count = whole;
        else {
            if (threadIdx.x == 0) write_status(&status[page], P4S_E_PLAIN_FLBA_EOF, 0, L, pv.size % L, 0, 0);
            return;
        }
    }
    const uint64_t nbytes = count * L;
    if (pd.aux_offset + nbytes > bytes_cap) {
        if (threadIdx.x == 0) write_status(&status[page], E_OUT_CAPACITY, count, nbytes, 0, 0, nbytes);
        return;
    }
    wg_copy(d_bytes + pd.aux_offset, pv.base, nbytes);
    if (threadIdx.x == 0) write_status(&status[page], P4S_E_NONE, count, 0, 0, nbytes, nbytes);
}

// ---------------------------------------------------------------------------
// PLAIN BYTE_ARRAY: [u32 length][bytes] ...  The length chain is serial; wave 0 follows it
// with the same speculate-at-every-byte + readlane chain used for run headers, then all
// waves copy the strings' bytes (4 output bytes per lane, string found by binary search).
// ---------------------------------------------------------------------------
struct PlainBaShared {
    alignas(16) uint8_t win[PBA_WIN + PBA_SLACK];
    uint32_t str_out[PBA_MAXSTR + 1];   // first output byte of each non-empty string (+ end sentinel)
    uint32_t str_pos[PBA_MAXSTR];       // page offset of its first byte
    uint32_t nstr;
    uint32_t out_begin, out_end;
    uint32_t more;
    uint32_t next_cur;
    int32_t err;
    uint64_t aux0, aux1;
};

struct PlainBaWalk {
    uint32_t cur;        // page offset of the next length field
    uint64_t strings;    // strings emitted
    uint64_t outp;       // output bytes so far
    bool finished;
};

__device__ __forceinline__ void plain_ba_walk_round(PlainBaShared &sh, const PageView &pv, int64_t wlo, PlainBaWalk &st,
                                                    uint64_t n, int64_t *offsets) {
    const uint32_t lane = lane_id();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win);
    uint32_t nstr = 0;
    const uint64_t out_begin = st.outp;
    int32_t err = 0;
    uint64_t a0 = 0, a1 = 0;
    while (!st.finished && st.strings < n) {
        if (st.cur >= pv.size) { st.finished = true; break; }                   // clean end (ref :432-434)
        if ((int64_t)st.cur + 72 > wlo + PBA_WIN) break;
        if (nstr + 64 > PBA_MAXSTR) break;
        const uint32_t p = st.cur + lane;
        const uint32_t rel = (uint32_t)((int64_t)p - wlo);
        const uint32_t d0 = win32[rel >> 2], d1 = win32[(rel >> 2) + 1];
        const uint32_t len = __funnelshift_r(d0, d1, (rel & 3u) * 8);
        const int64_t avail = (int64_t)pv.size - (int64_t)p;
        const bool short_len = avail < 4;                                       // includes avail <= 0
        const bool short_body = !short_len && (uint64_t)len > (uint64_t)(avail - 4);
        const bool fail = short_len || short_body;
        const uint64_t skip = 4ull + len;
        const uint32_t skip32 = skip > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)skip;
        const uint64_t failmask = __ballot(fail);
        uint64_t M = 0;
        uint32_t r = 0;
        uint32_t taken = 0;
        bool hit_fail = false, clean_end = false;
        while (r < 64 && st.strings + taken < n) {
            if ((uint64_t)st.cur + r == pv.size) { clean_end = true; break; }
            if ((failmask >> r) & 1ull) { hit_fail = true; break; }
            M |= 1ull << r;
            ++taken;
            r += bcast_u32(skip32, (int)r);
        }
        if (hit_fail) {
            const int64_t av = (int64_t)pv.size - (int64_t)(st.cur + r);
            if (av < 4) { err = P4S_E_PLAIN_BA_LEN_EOF; a0 = 4; a1 = (uint64_t)av; }          // ref :435-438
            else { err = P4S_E_PLAIN_BA_EOF; a0 = bcast_u32(len, (int)r); a1 = (uint64_t)(av - 4); }   // ref :442-445
            st.finished = true;
        }
        const bool inM = (M >> lane) & 1ull;
        const uint64_t l_eff = inM ? (uint64_t)len : 0;
        const uint64_t inc = wave_iscan(l_eff);
        const uint64_t start = st.outp + (inc - l_eff);
        const uint32_t rank = (uint32_t)__popcll(M & ((1ull << lane) - 1ull));
        if (inM) P4S_ST(offsets + st.strings + rank + 1, (int64_t)(start + len), 80);
        const bool emit = inM && len > 0;
        const uint64_t emitmask = __ballot(emit);
        if (emit) {
            const uint32_t idx = nstr + (uint32_t)__popcll(emitmask & ((1ull << lane) - 1ull));
            sh.str_out[idx] = (uint32_t)start;
            sh.str_pos[idx] = p + 4;
        }
        nstr += (uint32_t)__popcll(emitmask);
        st.outp += bcast_u64(inc, 63);
        st.strings += taken;
        if (hit_fail || clean_end || st.strings >= n) st.finished = true;
        else st.cur += r;
    }
    if (lane == 0) {
        sh.nstr = nstr;
        sh.str_out[nstr] = (uint32_t)st.outp;
        sh.out_begin = (uint32_t)out_begin;
        sh.out_end = (uint32_t)st.outp;
        sh.more = (!st.finished && st.strings < n) ? 1u : 0u;
        sh.next_cur = st.cur;
        if (err) { sh.err = err; sh.aux0 = a0; sh.aux1 = a1; }
    }
}

__global__ __launch_bounds__(PBA_THREADS) void k_plain_byte_array(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        int64_t *__restrict__ d_offsets, uint8_t *__restrict__ d_bytes, uint64_t bytes_cap,
        p4s_page_status *__restrict__ status) {
    __shared__ PlainBaShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    int64_t *offsets = d_offsets + pd.out_offset;
    uint8_t *obytes = d_bytes + pd.aux_offset;
    uint64_t cap = pd.param0;
    if (pd.aux_offset >= bytes_cap) cap = 0;
    else if (cap > bytes_cap - pd.aux_offset) cap = bytes_cap - pd.aux_offset;
    if (tid == 0) { sh.err = 0; P4S_ST(offsets, 0, 81); }
    PlainBaWalk st;
    st.cur = 0; st.strings = 0; st.outp = 0; st.finished = false;
    uint32_t wpos = 0;
    uint64_t total = 0, nstrings = 0;
    if (pd.num_values > 0) {
        for (;;) {
            const int64_t wlo = window_load<PBA_WIN, PBA_THREADS>(sh.win, pv, wpos);
            if (tid < PBA_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + PBA_WIN)[tid] = 0;
            __syncthreads();
            if (wave == 0) plain_ba_walk_round(sh, pv, wlo, st, pd.num_values, offsets);
            __syncthreads();
            const uint32_t nstr = sh.nstr;
            const uint32_t ob = sh.out_begin, oe = sh.out_end;
            for (uint32_t v0 = ob + tid * 4; v0 < oe; v0 += PBA_THREADS * 4) {
                uint32_t lo = 0, hi = nstr;
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh.str_out[mid] <= v0) lo = mid; else hi = mid;
                }
                uint32_t r = lo;
                uint32_t r_begin = sh.str_out[r], r_end = sh.str_out[r + 1], pos = sh.str_pos[r];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const uint32_t v = v0 + j;
                    if (v < oe) {
                        while (v >= r_end) {
                            ++r;
                            r_begin = r_end;
                            r_end = sh.str_out[r + 1];
                            pos = sh.str_pos[r];
                        }
                        if ((uint64_t)v < cap) P4S_ST(obytes + v, P4S_LD(pv.base + pos + (v - r_begin), 82), 83);
                    }
                }
            }
            total = oe;
            if (!sh.more) break;
            wpos = sh.next_cur;
            __syncthreads();
        }
    }
    __syncthreads();
    if (tid == 0) {
        // wave 0, lane 0 owns the walker state
        nstrings = st.strings;
        if (sh.err) write_status(&status[page], sh.err, 0, sh.aux0, sh.aux1, 0, 0);
        else if (total > cap) write_status(&status[page], E_OUT_CAPACITY, nstrings, total, cap, 0, total);
        else write_status(&status[page], P4S_E_NONE, nstrings, 0, 0, 0, total);
    }
}

}  // namespace p4s
