// RLE / bit-packed hybrid decode (ref: RleDecoder::GetBatch / NextCounts,
// include/parquet4seastar/rle_encoding.hh:282-348) and the three decoders built
// on it: level_decoder (encoding.hh:42-82, encoding.cc:26-79), dict_decoder
// (encoding.cc:468-508) and rle_decoder_boolean (encoding.cc:510-516).
//
// One 256-thread workgroup per page, in rounds:
//   1. all waves stage a 16 KiB window of the stream in LDS;
//   2. wave 0 locates the run headers: every lane parses the ULEB128 header that
//      WOULD start at byte cur+lane (speculatively, all 64 offsets at once), then
//      the true chain of headers inside the 64-byte span is followed with scalar
//      readlane steps; a DPP exclusive scan of the run lengths gives every run its
//      first output index.  Runs go to an LDS table {first index, payload position};
//   3. all waves expand the round's runs: each lane takes 4 consecutive output
//      values, finds their run by binary search in the LDS table, then either
//      broadcasts the repeated value or extracts the bit-packed literal from the
//      stream.  Dictionary pages gather through the index on the spot.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int RLE_THREADS = 256;
constexpr int RLE_WIN = 16384;
constexpr int RLE_SLACK = 16;
constexpr int RLE_MAXRUNS = 1024;
constexpr int RLE_VPT = 4;

enum RleMode { RLE_RAW = 0, RLE_LEVELS = 1, RLE_DICT = 2 };

struct RleShared {
    alignas(16) uint8_t win[RLE_WIN + RLE_SLACK];
    uint32_t run_out[RLE_MAXRUNS + 1];   // first output index of each run (+ end sentinel)
    uint32_t run_info[RLE_MAXRUNS];      // (stream byte offset of payload) << 1 | is_literal
    uint32_t nruns;
    uint32_t out_begin, out_end;         // output range covered by this round
    uint32_t more;                       // another round follows
    uint32_t next_cur;                   // stream offset of the next header
    unsigned long long errkey;           // dict: min over (position << 32 | bad index)
    uint32_t level_hits;                 // levels: values equal to max level
    uint32_t level_above;                // levels: values above max level (column_chunk_reader.hh:124-133 rejects them)
};

// 8 bytes at stream offset `off`, never touching memory past the input allocation.
__device__ __forceinline__ uint64_t ld_u64_guarded(const PageView &pv, uint64_t off) {
    if (off + 8 <= pv.room) {
        uint64_t v;
        __builtin_memcpy(&v, pv.base + off, 8);
        return v;
    }
    uint64_t v = 0;
    for (int i = 0; i < 8; ++i)
        if (off + i < pv.room) v |= (uint64_t)pv.base[off + i] << (8 * i);
    return v;
}

// Walk state, uniform in wave 0.
struct RleWalk {
    uint32_t cur;        // stream offset of the next run header
    uint64_t outp;       // values produced so far
    bool finished;       // no further run contributes
};

// One round of header location (wave 0, 64 lanes, uniform control flow).
// `n` = values wanted, `w` = bit width, `tbytes` = sizeof(output element),
// `chunk` = batch size in which the reference's caller drains the decoder (dict: 1000, else 0).
__device__ __forceinline__ void rle_walk_round(RleShared &sh, const PageView &pv, int64_t wlo, RleWalk &st,
                                               uint64_t n, uint32_t w, uint32_t tbytes, uint32_t chunk) {
    const uint32_t lane = lane_id();
    const uint32_t *win32 = reinterpret_cast<const uint32_t *>(sh.win);
    const uint32_t vbytes = (w + 7) >> 3;
    uint32_t nruns = 0;
    const uint64_t out_begin = st.outp;

    while (!st.finished && st.outp < n) {
        if (st.cur >= pv.size) { st.finished = true; break; }                 // clean end of stream
        if ((int64_t)st.cur + 72 > wlo + RLE_WIN) break;                       // span not staged: next window
        if (nruns + 64 > RLE_MAXRUNS) break;
        // ---- speculative parse at every byte offset of the span
        const uint32_t p = st.cur + lane;
        const uint32_t rel = (uint32_t)((int64_t)p - wlo);
        const uint32_t d0 = win32[rel >> 2], d1 = win32[(rel >> 2) + 1];
        const uint64_t x = (((uint64_t)d1 << 32) | d0) >> ((rel & 3u) * 8);    // bytes p .. p+4 (40 bits)
        const int64_t avail = (int64_t)pv.size - (int64_t)p;                   // stream bytes from p on
        const uint64_t stopbits = ~x & 0x8080808080ull;
        const uint32_t t = stopbits ? ((uint32_t)__builtin_ctzll(stopbits) >> 3) : 5u;
        const uint32_t hl = t + 1;
        bool fail = (t >= 5) || ((int64_t)hl > avail);                         // over-long or cut header (bit_stream_utils.hh:405-422)
        uint32_t h = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            if ((uint32_t)i <= t) h |= (uint32_t)((x >> (8 * i)) & 0x7Fu) << (7 * i);
        const uint32_t count = h >> 1;
        const bool lit = h & 1u;
        uint64_t vals, skip;
        if (lit) {
            if (count > 0x0FFFFFFFu) fail = true;                              // rle_encoding.hh:332-334
            vals = (uint64_t)count * 8;
            skip = (uint64_t)hl + (uint64_t)count * w;
        } else {
            if (vbytes > tbytes || (int64_t)(hl + vbytes) > avail) fail = true; // bit_stream_utils.hh:370-377
            vals = count;
            skip = hl + vbytes;
        }
        const uint32_t skip32 = skip > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)skip;
        // ---- follow the real chain through the span
        uint64_t M = 0;
        uint32_t r = 0;
        const uint32_t failmask_lo = 0;
        (void)failmask_lo;
        const uint64_t failmask = __ballot(fail);
        bool hit_fail = false;
        while (r < 64) {
            M |= 1ull << r;
            if ((failmask >> r) & 1ull) { hit_fail = true; break; }
            r += bcast_u32(skip32, (int)r);
        }
        // ---- output index of every real run
        const bool inM = (M >> lane) & 1ull;
        const bool good = inM && !fail;
        const uint64_t v_eff = good ? vals : 0;
        const uint64_t inc = wave_iscan(v_eff);
        const uint64_t start = st.outp + (inc - v_eff);
        const bool needed = good && start < n;
        const uint64_t want = needed ? (vals < n - start ? vals : n - start) : 0;
        // a literal run that cannot supply `want` values ends the decode (rle_encoding.hh:299-304)
        bool trunc = false;
        if (needed && lit && w > 0) {
            const uint64_t avail_vals = ((uint64_t)(avail - hl) * 8) / w;
            trunc = want > avail_vals;
        }
        const uint64_t truncmask = __ballot(trunc);
        const uint64_t stopmask = truncmask | (hit_fail ? (failmask & M) : 0);
        const uint32_t S = stopmask ? (uint32_t)__builtin_ctzll(stopmask) : 64u;   // first lane that ends the stream
        uint64_t contrib = (needed && lane < S) ? want : 0;
        if (lane == S && trunc && chunk) {
            // the caller drains in `chunk`-sized batches: whole leading pieces still succeed
            const uint64_t avail_vals = ((uint64_t)(avail - hl) * 8) / w;
            uint64_t pos = start, got = 0;
            while (got < want) {
                uint64_t piece = chunk - (pos % chunk);
                if (piece > want - got) piece = want - got;
                if (got + piece > avail_vals) break;
                got += piece; pos += piece;
            }
            contrib = got;
        }
        const bool emit = contrib > 0;
        const uint64_t emitmask = __ballot(emit);
        if (emit) {
            const uint32_t idx = nruns + (uint32_t)__popcll(emitmask & ((1ull << lane) - 1ull));
            sh.run_out[idx] = (uint32_t)start;
            sh.run_info[idx] = ((p + hl) << 1) | (lit ? 1u : 0u);
        }
        nruns += (uint32_t)__popcll(emitmask);
        // total produced by this span = last emitting lane's start + contrib - outp
        uint64_t produced = 0;
        if (emitmask) {
            const int last = 63 - __builtin_clzll(emitmask);
            produced = bcast_u64(start, last) + bcast_u64(contrib, last) - st.outp;
        }
        st.outp += produced;
        if (S < 64 || st.outp >= n) st.finished = true;
        else st.cur += r;
    }
    if (lane == 0) {
        sh.nruns = nruns;
        sh.run_out[nruns] = (uint32_t)st.outp;
        sh.out_begin = (uint32_t)out_begin;
        sh.out_end = (uint32_t)st.outp;
        sh.more = (!st.finished && st.outp < n) ? 1u : 0u;
        sh.next_cur = st.cur;
    }
}

// Output-side policy: what happens to a decoded hybrid value.
template <typename OutT>
struct RleSinkPlain {
    OutT *out;
    __device__ __forceinline__ void store4(uint32_t v0, uint32_t cnt, const uint32_t x[RLE_VPT], RleShared &) const {
        OutT tmp[RLE_VPT];
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) tmp[j] = (OutT)x[j];
        if (cnt == RLE_VPT) {
            __builtin_memcpy(out + v0, tmp, sizeof(tmp));
        } else {
#pragma unroll
            for (int j = 0; j < RLE_VPT; ++j)
                if ((uint32_t)j < cnt) out[v0 + j] = tmp[j];
        }
    }
};

template <int ELEM>
struct RleSinkDict {
    uint8_t *out;
    const uint8_t *dict;
    uint64_t dict_size;
    __device__ __forceinline__ void store4(uint32_t v0, uint32_t cnt, const uint32_t x[RLE_VPT], RleShared &sh) const {
#pragma unroll
        for (int j = 0; j < RLE_VPT; ++j) {
            if ((uint32_t)j >= cnt) continue;
            const uint32_t idx = x[j];
            if ((uint64_t)idx > dict_size) {            // encoding.cc:489 (idx == dict_size passes)
                atomicMin(&sh.errkey, ((unsigned long long)(v0 + j) << 32) | idx);
                continue;
            }
            if constexpr (ELEM == 1) {
                out[v0 + j] = dict[idx];
            } else if constexpr (ELEM == 4) {
                reinterpret_cast<uint32_t *>(out)[v0 + j] = reinterpret_cast<const uint32_t *>(dict)[idx];
            } else if constexpr (ELEM == 8) {
                reinterpret_cast<uint64_t *>(out)[v0 + j] = reinterpret_cast<const uint64_t *>(dict)[idx];
            } else {
                const uint32_t *s = reinterpret_cast<const uint32_t *>(dict) + (uint64_t)idx * 3;
                uint32_t *d = reinterpret_cast<uint32_t *>(out) + (uint64_t)(v0 + j) * 3;
                d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
            }
        }
    }
};

// Shared body: rounds of (window, walk, expand) over one hybrid stream `pv`.
// Returns the number of values produced (uniform across the workgroup).
template <typename Sink>
__device__ __forceinline__ uint64_t rle_decode_stream(RleShared &sh, const PageView &pv, uint64_t n, uint32_t w,
                                                      uint32_t tbytes, uint32_t chunk, const Sink &sink,
                                                      bool count_levels, uint32_t max_level) {
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = tid >> 6;
    RleWalk st;
    st.cur = 0;
    st.outp = 0;
    st.finished = false;
    uint32_t wpos = 0;
    uint32_t hits = 0, above = 0;
    const uint32_t vbytes = (w + 7) >> 3;
    const uint32_t vmask = vbytes >= 4 ? 0xFFFFFFFFu : ((1u << (8 * vbytes)) - 1u);
    uint64_t total = 0;
    if (n == 0) return 0;
    for (;;) {
        const int64_t wlo = window_load<RLE_WIN, RLE_THREADS>(sh.win, pv, wpos);
        if (tid < RLE_SLACK / 4) reinterpret_cast<uint32_t *>(sh.win + RLE_WIN)[tid] = 0;
        __syncthreads();
        if (wave == 0) rle_walk_round(sh, pv, wlo, st, n, w, tbytes, chunk);
        __syncthreads();
        const uint32_t nruns = sh.nruns;
        const uint32_t ob = sh.out_begin, oe = sh.out_end;
        for (uint32_t v0 = ob + tid * RLE_VPT; v0 < oe; v0 += RLE_THREADS * RLE_VPT) {
            // run holding v0: last r with run_out[r] <= v0
            uint32_t lo = 0, hi = nruns;           // invariant: run_out[lo] <= v0 < run_out[hi]
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (sh.run_out[mid] <= v0) lo = mid; else hi = mid;
            }
            uint32_t r = lo;
            uint32_t r_begin = sh.run_out[r], r_end = sh.run_out[r + 1];
            uint32_t info = sh.run_info[r];
            uint32_t x[RLE_VPT];
            const uint32_t cnt = oe - v0 < RLE_VPT ? oe - v0 : RLE_VPT;
            uint32_t rep_val = 0;
            bool rep_loaded = false;
#pragma unroll
            for (int j = 0; j < RLE_VPT; ++j) {
                const uint32_t v = v0 + j;
                uint32_t val = 0;
                if ((uint32_t)j < cnt) {
                    while (v >= r_end) {           // advance to the run holding v
                        ++r;
                        r_begin = r_end;
                        r_end = sh.run_out[r + 1];
                        info = sh.run_info[r];
                        rep_loaded = false;
                    }
                    const uint32_t pos = info >> 1;
                    if (info & 1u) {
                        const uint64_t bit = (uint64_t)(v - r_begin) * w;
                        const uint64_t raw = ld_u64_guarded(pv, (uint64_t)pos + (bit >> 3));
                        val = mask32((uint32_t)(raw >> (bit & 7u)), w);
                    } else {
                        if (!rep_loaded) {
                            rep_val = (uint32_t)ld_u64_guarded(pv, pos) & vmask;   // raw bytes, not masked to w
                            if (vbytes == 0) rep_val = 0;
                            rep_loaded = true;
                        }
                        val = rep_val;
                    }
                    if (count_levels) {
                        if (val == max_level) ++hits;
                        else if (val > max_level) ++above;
                    }
                }
                x[j] = val;
            }
            sink.store4(v0, cnt, x, sh);
        }
        total = oe;
        if (!sh.more) break;
        wpos = sh.next_cur;
        __syncthreads();
    }
    if (count_levels) {
        __syncthreads();
        if (hits) atomicAdd(&sh.level_hits, hits);
        if (above) atomicAdd(&sh.level_above, above);
        __syncthreads();
    }
    return total;
}

// p4s_decode_rle: bare hybrid stream, param0 = bit width.
template <typename OutT>
__global__ __launch_bounds__(RLE_THREADS) void k_rle_raw(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        OutT *__restrict__ d_out, p4s_page_status *__restrict__ status) {
    __shared__ RleShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    if (pd.param0 > 32) {
        if (threadIdx.x == 0) write_status(&status[page], P4S_E_BAD_ENCODING, 0, pd.param0, 0, 0, 0);
        return;
    }
    RleSinkPlain<OutT> sink{d_out + pd.out_offset};
    const uint64_t got = rle_decode_stream(sh, pv, pd.num_values, pd.param0, sizeof(OutT), 0, sink, false, 0);
    if (threadIdx.x == 0) write_status(&status[page], P4S_E_NONE, got, 0, 0, 0, 0);
}

// p4s_decode_levels: level_decoder::reset_v1 / reset_v2 + one read_batch.
template <typename OutT>
__global__ __launch_bounds__(RLE_THREADS) void k_levels(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        OutT *__restrict__ d_out, p4s_page_status *__restrict__ status) {
    __shared__ RleShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    const uint32_t max_level = pd.param0;
    const uint32_t w = max_level ? 32u - (uint32_t)__builtin_clz(max_level) : 0u;   // encoding.hh:47-49
    uint64_t n = pd.num_values < pd.param1 ? pd.num_values : pd.param1;           // encoding.hh:63
    OutT *out = d_out + pd.out_offset;
    if (tid == 0) { sh.level_hits = 0; sh.level_above = 0; }
    if (w == 0) {                                                                  // encoding.hh:64-68
        for (uint64_t i = tid; i < n; i += RLE_THREADS) out[i] = 0;
        if (tid == 0) write_status(&status[page], P4S_E_NONE, n, n, 0, 0, 0);      // every level equals max level 0
        return;
    }
    uint64_t consumed = 0;
    const uint32_t kind = pd.flags & 3u;
    if (kind == P4S_LEVELS_V1_RLE) {                                               // encoding.cc:35-50
        int32_t err = 0;
        uint64_t a0 = 0, a1 = 0;
        int32_t len = 0;
        if (pv.size < 4) { err = P4S_E_LEVELS_EOF; a0 = 4; a1 = pv.size; }
        else {
            uint32_t raw = (uint32_t)pv.base[0] | ((uint32_t)pv.base[1] << 8) | ((uint32_t)pv.base[2] << 16) | ((uint32_t)pv.base[3] << 24);
            len = (int32_t)raw;
            if (len < 0) { err = P4S_E_LEVELS_NEGATIVE; a0 = (uint64_t)(int64_t)len; }
            else if ((uint64_t)len > pv.size) { err = P4S_E_LEVELS_EOF; a0 = (uint64_t)len; a1 = pv.size; }
        }
        if (err) {
            if (tid == 0) write_status(&status[page], err, 0, a0, a1, 0, 0);
            return;
        }
        // the reference checks len against size(), not size()-4; clamp instead of running past the page
        uint64_t use = (uint64_t)len < pv.size - 4 ? (uint64_t)len : pv.size - 4;
        consumed = 4 + use;
        pv.base += 4;
        pv.room -= 4;
        pv.size = use;
    } else if (kind == P4S_LEVELS_V1_BIT_PACKED) {                                 // encoding.cc:51-62
        const uint64_t byte_len = ((uint64_t)pd.param1 * w + 7) >> 3;
        if (byte_len > 0x7FFFFFFFull) {
            if (tid == 0) write_status(&status[page], P4S_E_LEVELS_BITPACKED_LEN, 0, byte_len, 0, 0, 0);
            return;
        }
        if (byte_len > pv.size) {
            if (tid == 0) write_status(&status[page], P4S_E_LEVELS_EOF, 0, byte_len, pv.size, 0, 0);
            return;
        }
        // plain bit-packed values, LSB first (BitReader::GetBatch), no run headers
        uint32_t hits = 0, above = 0;
        for (uint64_t i = tid; i < n; i += RLE_THREADS) {
            const uint64_t bit = i * w;
            const uint64_t raw = ld_u64_guarded(pv, bit >> 3);
            const uint32_t val = mask32((uint32_t)(raw >> (bit & 7u)), w);
            out[i] = (OutT)val;
            if (val == max_level) ++hits;
            else if (val > max_level) ++above;
        }
        __syncthreads();
        if (hits) atomicAdd(&sh.level_hits, hits);
        if (above) atomicAdd(&sh.level_above, above);
        __syncthreads();
        if (tid == 0) write_status(&status[page], P4S_E_NONE, n, sh.level_hits, sh.level_above, byte_len, 0);
        return;
    } else {                                                                       // v2: encoding.cc:68-79
        if (pv.size > 0x7FFFFFFFull) {
            if (tid == 0) write_status(&status[page], P4S_E_LEVELS_V2_LEN, 0, pv.size, 0, 0, 0);
            return;
        }
        consumed = pv.size;
    }
    __syncthreads();
    RleSinkPlain<OutT> sink{out};
    const uint64_t got = rle_decode_stream(sh, pv, n, w, sizeof(OutT), 0, sink, true, max_level);
    if (tid == 0) write_status(&status[page], P4S_E_NONE, got, sh.level_hits, sh.level_above, consumed, 0);
}

// p4s_decode_dictionary: width byte + hybrid indices, gathered through the dictionary.
template <int ELEM>
__global__ __launch_bounds__(RLE_THREADS) void k_dict(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        const uint8_t *__restrict__ d_dict, uint64_t dict_size,
        uint8_t *__restrict__ d_out, p4s_page_status *__restrict__ status) {
    __shared__ RleShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    if (pv.size == 0) {
        if (tid == 0) write_status(&status[page], P4S_E_DICT_EMPTY_PAGE, 0, 0, 0, 0, 0);
        return;
    }
    const uint32_t w = pv.base[0];
    if (w > 32) {                                                                  // encoding.cc:473-477
        if (tid == 0) write_status(&status[page], P4S_E_DICT_BAD_WIDTH, 0, w, 0, 0, 0);
        return;
    }
    if (tid == 0) sh.errkey = ~0ull;
    pv.base += 1;
    pv.room -= 1;
    pv.size -= 1;
    __syncthreads();
    RleSinkDict<ELEM> sink{d_out + pd.out_offset * ELEM, d_dict, dict_size};
    const uint64_t got = rle_decode_stream(sh, pv, pd.num_values, w, 4, 1000, sink, false, 0);
    __syncthreads();
    if (tid == 0) {
        const unsigned long long k = sh.errkey;
        if (k != ~0ull) write_status(&status[page], P4S_E_DICT_INDEX, 0, dict_size, k & 0xFFFFFFFFull, 0, 0);
        else write_status(&status[page], P4S_E_NONE, got, 0, 0, 0, 0);
    }
}

// values_to_read of column_chunk_reader.hh:134-140: copy the level hit count into a value page's n.
__global__ void k_set_num_values(p4s_page_desc *pages, const p4s_page_status *level_status, uint32_t n_pages) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_pages) pages[i].num_values = level_status[i].code == 0 ? (uint32_t)level_status[i].aux0 : 0u;
}

}  // namespace p4s
