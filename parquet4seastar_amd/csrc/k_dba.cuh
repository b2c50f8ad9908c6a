// DELTA_BYTE_ARRAY kernel (ref: delta_byte_array_decoder, src/encoding.cc:296-362; parity is
// with a single whole-page read_batch, the only call shape the reference decodes correctly).
//
// value_i = value_{i-1}[0 : prefix_i] ++ suffix_i.  Phases, one workgroup per page:
//   A. prefix lengths  = nested DELTA_BINARY_PACKED INT32 stream  -> scratch
//   B. suffix lengths  = second nested stream; two prefix sums per tile give every string its
//      suffix position and its output offset (offsets[] is final after this phase)
//   C. 256 strings at a time.  The tile's suffix bytes are staged in LDS with coalesced 16-byte
//      loads.  Each wave then replays its 64 strings in order with one lane per BYTE: the lanes
//      hold the current string, string i overwrites bytes [prefix_i, len_i) with its suffix and
//      the whole string is written to the tile's LDS image - a fixed cost per string, whatever
//      the prefix structure.  Bytes below the running minimum of the prefix lengths are
//      inherited from before the wave's first string: they are patched afterwards from the
//      last string of the previous wave (the string before the tile for wave 0), which is why
//      the four waves can run side by side.  The assembled tile leaves LDS with coalesced
//      16-byte stores.  Tiles with strings over 256 bytes, or too large for the LDS buffers,
//      take a lane-per-string path through global memory.
#pragma once
#include "k_strings.cuh"

namespace p4s {

constexpr int DBA_TILE = 256;
static_assert(DBA_TILE == DELTA_THREADS, "one string per lane");
constexpr int DBA_OUT_CAP = 12288;    // output bytes of a tile staged in LDS
constexpr int DBA_SUF_CAP = 6144;     // suffix bytes of a tile staged in LDS
constexpr int DBA_PREV_CAP = 1024;    // bytes of the string before the tile kept in LDS
constexpr int DBA_WAVE_LEN = 256;     // longest string the byte-per-lane replay keeps in registers (4 per lane)

struct DbaTileShared {
    uint32_t prefix[DBA_TILE + 1];    // entry e (1-based) = string t0+e-1
    uint32_t sufoff[DBA_TILE + 2];    // suffix offset; [cnt+1] = end of the last suffix of the tile
    uint64_t last_start;              // output offset of the tile's last string
    uint64_t last_len;                // and its length
};

struct DbaStage {
    alignas(16) uint8_t obuf[DBA_OUT_CAP + 32];   // obuf[shift + r] = byte r of the tile's output, shift = output address & 15
    alignas(16) uint8_t sbuf[DBA_SUF_CAP + 32];   // sbuf[x - slo] = byte x of the page's suffix data
    alignas(16) uint8_t prev[DBA_PREV_CAP];       // the string before the tile (when it fits)
    alignas(16) uint8_t wlast[4][DBA_WAVE_LEN];   // last string of each wave's 64 strings
    uint32_t wlast_len[4], wlast_hole[4];         // its length; bytes [0, hole) are inherited from the wave before
};

struct DbaShared {
    union {
        DeltaShared delta;            // phases A and B
        DbaStage stage;               // phase C
    } u;
    WgScan64 scan, scan2;
    int32_t err;
    DbaTileShared ts;
};

// One lane's string: suffix bytes, then the shared prefix from the earlier strings of the tile (entries
// k .. 1 of ts), then from the string before the tile.  `suf(k)` = pointer to the suffix bytes of entry k.
template <typename SufAt>
__device__ __forceinline__ void dba_assemble(uint8_t *dst, const DbaTileShared &ts, uint32_t e, uint32_t pfx, uint32_t slen,
                                             const uint8_t *prev_bytes, SufAt suf) {
    const uint8_t *src = suf(e);
    for (uint32_t b = 0; b < slen; ++b) P4S_ST(dst + pfx + b, P4S_LD(src + b, 30), 31);
    uint32_t need = pfx;
    for (uint32_t k = e - 1; need > 0 && k >= 1; --k) {
        const uint32_t pk = ts.prefix[k];
        if (pk < need) {
            const uint8_t *sk = suf(k);
            for (uint32_t b = pk; b < need; ++b) P4S_ST(dst + b, P4S_LD(sk + (b - pk), 32), 33);
            need = pk;
        }
    }
    for (uint32_t b = 0; b < need; ++b) P4S_ST(dst + b, P4S_LD(prev_bytes + b, 34), 35);
}

#ifndef P4S_DBA_MIN_WG
#define P4S_DBA_MIN_WG 4      // workgroups per CU the register budget is sized for
#endif

template <bool SIZES_ONLY>
__global__ __launch_bounds__(DELTA_THREADS, P4S_DBA_MIN_WG) void k_delta_byte_array(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        int64_t *__restrict__ d_offsets, uint8_t *__restrict__ d_bytes, uint64_t bytes_cap,
        uint32_t *__restrict__ d_scratch, uint64_t scratch_slots, p4s_page_status *__restrict__ status) {
    __shared__ DbaShared sh;
    DbaTileShared &ts = sh.ts;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    const uint64_t slot0 = pd.out_offset;
    const uint64_t slots = slot0 < scratch_slots ? scratch_slots - slot0 : 0;
    uint32_t *scratch = d_scratch + 2 * slot0;
    int64_t *offsets = SIZES_ONLY ? nullptr : d_offsets + slot0;
    if (tid == 0) sh.err = 0;
#ifdef P4S_EXP_TIMING
    const uint64_t tk_a = __builtin_readcyclecounter();
#endif

    // ---- A: every prefix length of the first stream (ref :331-345).  Only the first num_values of them are used
    // (and kept: the scratch slots after them belong to the next page).
    DbaPrefixSink psink{scratch, slots < pd.num_values ? slots : pd.num_values};
    DeltaResult pres;
    delta_decode_stream<uint32_t>(sh.u.delta, pv, 0xFFFFFFFFull, true, psink, pres);
    if (pres.err) {
        if (tid == 0) write_status(&status[page], pres.err, 0, pres.aux0, pres.aux1, 0, 0);
        return;
    }
    __syncthreads();
#ifdef P4S_EXP_TIMING
    const uint64_t tk_b = __builtin_readcyclecounter();
#endif

    // ---- B: the suffixes are a DELTA_LENGTH_BYTE_ARRAY section (ref :347-358)
    PageView sv = pv;
    sv.base += pres.end_pos;
    sv.room -= pres.end_pos;
    sv.size -= pres.end_pos;
    uint64_t n_suffix = 0;       // third ULEB128 of the nested header: caps what is emitted
    {
        uint64_t tmp, pos = 0;
        uint32_t k = uleb_global(sv, pos, 10, &tmp);
        if (k) { pos += k; k = uleb_global(sv, pos, 10, &tmp); }
        if (k) { pos += k; k = uleb_global(sv, pos, 10, &n_suffix); }
        if (!k) n_suffix = 0;    // the decode below reports the header error
    }
    const uint64_t limit = n_suffix < pd.num_values ? n_suffix : pd.num_values;        // ref :306
    if (limit > slots) {
        if (tid == 0) write_status(&status[page], E_OUT_CAPACITY, limit, 0, 0, 0, 0);
        return;
    }
    DbaSuffixSink<SIZES_ONLY> ssink{scratch, offsets, limit, pres.total, &sh.scan, &sh.scan2, 0, 0, 0, 0, 0, false};
    DeltaResult sres;
    delta_decode_stream<uint32_t>(sh.u.delta, sv, 0xFFFFFFFFull, true, ssink, sres);
    if (sres.err) {
        if (tid == 0) write_status(&status[page], sres.err, 0, sres.aux0, sres.aux1, 0, 0);
        return;
    }
    if (!SIZES_ONLY && limit == 0 && tid == 0) P4S_ST(offsets, 0, 36);
    const uint64_t data_pos = pres.end_pos + sres.end_pos;       // first suffix byte in the page
    const uint64_t data_len = pv.size - data_pos;
    // only the thread that met a string without a prefix length has set the flag: the whole workgroup must leave
    const bool missing_prefix = __syncthreads_or(ssink.missing_prefix) != 0;
    if (ssink.all_suffix > data_len) {                            // ref :262-266, reached through :348-358
        if (tid == 0) write_status(&status[page], P4S_E_DLBA_VALUES_EOF, 0, 0, 0, 0, 0);
        return;
    }
    if (missing_prefix) {         // more suffixes than prefix lengths: the reference indexes past its array
        if (tid == 0) write_status(&status[page], P4S_E_DBA_BAD_PREFIX, 0, 0, 0, 0, 0);
        return;
    }
    const uint64_t out_bytes = ssink.out_carry;
    const uint64_t suffix_end = ssink.suf_carry;                  // end of the last emitted suffix
    bool can_write = !SIZES_ONLY;
    if (!SIZES_ONLY && (out_bytes > pd.param0 || pd.aux_offset + out_bytes > bytes_cap)) can_write = false;
    __syncthreads();              // the stream state in sh.u.delta is dead from here on

#ifdef P4S_EXP_TIMING
    const uint64_t tk_c = __builtin_readcyclecounter();
#endif
    // ---- C: validate prefix_i <= len(value_{i-1}) (ref :310-312) and rebuild the strings
    DbaStage &st = sh.u.stage;
    const uint8_t *sdata = pv.base + data_pos;
    const int64_t sroom = (int64_t)pv.room - (int64_t)data_pos;   // bytes readable from sdata
    uint8_t *obytes = SIZES_ONLY ? nullptr : d_bytes + pd.aux_offset;
    uint64_t prev_start = 0, prev_len = 0, prev_end = 0;   // the string before the tile
    bool prev_in_lds = true;                                // (the empty string before the page is "in LDS")
    uint32_t buf = 0;
    // a tile's (prefix length, suffix offset) pairs are fetched from scratch one tile ahead
    auto fetch_pair = [&](uint64_t i) -> uint2 {
        if (i < limit) return P4S_LD(reinterpret_cast<const uint2 *>(scratch + 2 * i), 37);
        return make_uint2(0u, (uint32_t)suffix_end);
    };
    uint2 mine = fetch_pair(tid), after = fetch_pair((uint64_t)tid + 1);   // string tid of the tile and the one after it
#ifdef P4S_EXP_TIMING
    uint64_t cacc[5] = {0, 0, 0, 0, 0}, clast = 0;
#define DBA_TICK(i) do { const uint64_t n_ = __builtin_readcyclecounter(); if ((i) > 0) cacc[(i) - 1] += n_ - clast; clast = n_; } while (0)
#else
#define DBA_TICK(i) do { } while (0)
#endif
    for (uint64_t t0 = 0; t0 < limit; t0 += DBA_TILE) {
        const uint32_t cnt = (uint32_t)(limit - t0 < DBA_TILE ? limit - t0 : DBA_TILE);
        DBA_TICK(0);
        if (tid < cnt) {
            ts.prefix[tid + 1] = mine.x;
            ts.sufoff[tid + 1] = mine.y;
            if (tid == cnt - 1) ts.sufoff[cnt + 1] = after.y;
        }
        mine = fetch_pair(t0 + DBA_TILE + tid);
        after = fetch_pair(t0 + DBA_TILE + tid + 1);
        __syncthreads();
        uint32_t pfx = 0, slen = 0;
        if (tid < cnt) {
            pfx = ts.prefix[tid + 1];
            slen = ts.sufoff[tid + 2] - ts.sufoff[tid + 1];
        }
        const uint64_t my_len = (uint64_t)pfx + slen;
        uint64_t tile_bytes;
        const uint64_t my_rel = wg_exclusive_scan(sh.scan, buf, my_len, tile_bytes);     // offset inside the tile's output
        const uint64_t my_start = prev_end + my_rel;
        if (tid < cnt) {
            const uint64_t before = tid == 0 ? prev_len
                                             : (uint64_t)ts.prefix[tid] + (ts.sufoff[tid + 1] - ts.sufoff[tid]);
            if ((uint64_t)pfx > before) atomicExch(&sh.err, (int32_t)P4S_E_DBA_BAD_PREFIX);
            if (tid == cnt - 1) { ts.last_start = my_start; ts.last_len = my_len; }
        }
        const bool long_strings = __syncthreads_or(my_len > (uint64_t)DBA_WAVE_LEN) != 0;
        if (sh.err) break;
        const uint32_t suf0 = ts.sufoff[1];
        const uint32_t suf_bytes = ts.sufoff[cnt + 1] - suf0;
        const uint64_t last_len = ts.last_len;
        DBA_TICK(1);
        if (can_write) {
            const uint8_t *prev_bytes = prev_in_lds ? st.prev : obytes + prev_start;   // complete since the previous iteration
            if (tile_bytes <= DBA_OUT_CAP && suf_bytes <= DBA_SUF_CAP) {
                // suffix bytes of the tile -> LDS (16-byte chunks from an aligned address; bytes past the input read as 0)
                const uint32_t smis = (uint32_t)((uintptr_t)(sdata + suf0) & 15u);
                const int64_t slo = (int64_t)suf0 - smis;               // suffix-data offset of sbuf[0] (may precede the section)
                const uint32_t schunks = (smis + suf_bytes + 15) >> 4;
                for (uint32_t c = tid; c < schunks; c += DELTA_THREADS) {
                    const int64_t o = slo + (int64_t)c * 16;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (o + 16 <= sroom) {
                        v = P4S_LD(reinterpret_cast<const uint4 *>(sdata + o), 38);
                    } else {
                        uint8_t tmp[16];
                        for (int b = 0; b < 16; ++b) tmp[b] = (o + b < sroom) ? P4S_LD(sdata + o + b, 39) : 0;
                        v = *reinterpret_cast<uint4 *>(tmp);
                    }
                    *reinterpret_cast<uint4 *>(st.sbuf + c * 16) = v;
                }
                __syncthreads();
                DBA_TICK(2);
                uint8_t *gdst = obytes + prev_end;                      // where the tile's first byte goes
                const uint32_t shift = (uint32_t)((uintptr_t)gdst & 15u);
                uint8_t *tile = st.obuf + shift;                       // tile[r] = byte r of the tile's output
                if (!long_strings) {
                    // ---- one lane per byte: every wave replays its 64 strings in order
                    const uint32_t wv = first_u32(tid >> 6), lane = tid & 63u;
                    const uint32_t rel32 = (uint32_t)my_rel, len32 = (uint32_t)my_len;
                    const uint32_t srel = tid < cnt ? (uint32_t)((int64_t)ts.sufoff[tid + 1] - slo) : 0u;
                    // bytes [0, hole) of a string are inherited from before the wave's first string
                    const uint32_t hole = ~wave_imaxscan(tid < cnt ? ~pfx : 0u);
                    const uint32_t wcnt = cnt > wv * 64 ? (cnt - wv * 64 < 64 ? cnt - wv * 64 : 64u) : 0u;
                    const uint32_t wmax = bcast_u32(wave_imaxscan(tid < cnt ? len32 : 0u), 63);     // longest string of the wave
                    uint32_t cur[DBA_WAVE_LEN / 64] = {0, 0, 0, 0};
                    if (wmax <= 64) {
                        // ---- short strings: one lane per STRING, one step per byte column.  Byte j of string i comes
                        // from the suffix of the last string k <= i of the wave with prefix_k <= j (a DPP max-scan of the
                        // qualifying lane numbers); no such k: the byte is inherited (patched below).
                        const uint32_t packed = (srel << 8) | pfx;                       // pfx <= 64 here
                        uint8_t *dst = tile + rel32;
                        // four columns per step: the four scan / permute / read chains are independent
                        for (uint32_t j0 = 0; j0 < wmax; j0 += 4) {
                            uint32_t src1[4];        // lane + 1 of the source, 0: none
#pragma unroll
                            for (int u = 0; u < 4; ++u) src1[u] = (tid < cnt && pfx <= j0 + u) ? lane + 1u : 0u;
                            wave_imaxscan_n(src1);
                            uint32_t pk[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                pk[u] = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((src1[u] - 1u) << 2), (int)packed);
                            uint8_t bt[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const bool on = src1[u] && j0 + u < len32;
                                bt[u] = on ? st.sbuf[(pk[u] >> 8) + j0 + u - (pk[u] & 0xFFu)] : (uint8_t)0;
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u)
                                if (src1[u] && j0 + u < len32) dst[j0 + u] = bt[u];
                        }
                        // the wave's last string, for the wave after it (bytes below its hole bound are patched in order below)
                        if (wcnt && lane == wcnt - 1) {
                            for (uint32_t j = hole; j < len32; ++j) st.wlast[wv][j] = dst[j];
                            st.wlast_len[wv] = len32;
                            st.wlast_hole[wv] = hole;
                        } else if (!wcnt && lane == 0) {
                            st.wlast_len[wv] = 0; st.wlast_hole[wv] = 0;
                        }
                    } else {
                    for (uint32_t s = 0; s < wcnt; ++s) {
                        const uint32_t p = bcast_u32(pfx, (int)s), n = bcast_u32(len32, (int)s);
                        const uint32_t sr = bcast_u32(srel, (int)s), orl = bcast_u32(rel32, (int)s), h = bcast_u32(hole, (int)s);
#pragma unroll
                        for (int c = 0; c < DBA_WAVE_LEN / 64; ++c) {
                            const uint32_t j = (uint32_t)c * 64 + lane;
                            if ((uint32_t)c * 64 < n) {
                                if (j >= p && j < n) cur[c] = st.sbuf[sr + j - p];
                                if (j >= h && j < n) tile[orl + j] = (uint8_t)cur[c];
                            }
                        }
                    }
                    if (wcnt) {
                        const uint32_t n = bcast_u32(len32, (int)(wcnt - 1));
#pragma unroll
                        for (int c = 0; c < DBA_WAVE_LEN / 64; ++c)
                            if ((uint32_t)c * 64 + lane < n) st.wlast[wv][c * 64 + lane] = (uint8_t)cur[c];
                        if (lane == 0) { st.wlast_len[wv] = n; st.wlast_hole[wv] = bcast_u32(hole, (int)(wcnt - 1)); }
                    } else if (lane == 0) {
                        st.wlast_len[wv] = 0; st.wlast_hole[wv] = 0;
                    }
                    }
                    __syncthreads();
                    DBA_TICK(3);
                    // ---- complete the waves' last strings in order, then every string's inherited bytes
                    // (one wave does all four in order: its LDS accesses are ordered, no barrier between the steps)
                    const uint32_t nw = (cnt + 63) >> 6;
                    if (wv == 0) {
                        for (uint32_t w2 = 0; w2 < nw; ++w2) {
                            const uint8_t *from = w2 == 0 ? prev_bytes : st.wlast[w2 - 1];
                            const uint32_t hh = st.wlast_hole[w2];
                            for (uint32_t b = lane; b < hh; b += 64) st.wlast[w2][b] = P4S_LD(from + b, 40);
                        }
                    }
                    __syncthreads();
                    if (tid < cnt && hole > 0) {
                        const uint8_t *from = wv == 0 ? prev_bytes : st.wlast[wv - 1];
                        uint8_t *dst = tile + rel32;
                        for (uint32_t b = 0; b < hole; ++b) dst[b] = P4S_LD(from + b, 41);
                    }
                } else if (tid < cnt) {
                    const uint8_t *sb = st.sbuf;
                    dba_assemble(tile + (uint32_t)my_rel, ts, tid + 1, pfx, slen, prev_bytes,
                                 [&](uint32_t k) { return sb + (uint32_t)((int64_t)ts.sufoff[k] - slo); });
                }
                __syncthreads();
                DBA_TICK(4);
                // the tile -> global memory: obuf and the destination agree modulo 16
                const uint32_t total = shift + (uint32_t)tile_bytes;
                uint8_t *g0 = gdst - shift;                              // 16-byte aligned
                const uint32_t chunks = (total + 15) >> 4;
                for (uint32_t c = tid; c < chunks; c += DELTA_THREADS) {
                    const uint32_t lo = c * 16, hi = lo + 16;
                    if (lo >= shift && hi <= total) {
                        P4S_ST(reinterpret_cast<uint4 *>(g0 + lo), *reinterpret_cast<const uint4 *>(st.obuf + lo), 42);
                    } else {
                        for (uint32_t b = lo < shift ? shift : lo; b < hi && b < total; ++b) P4S_ST(g0 + b, st.obuf[b], 43);
                    }
                }
                // keep the tile's last string for the next tile
                const bool keep = last_len <= DBA_PREV_CAP;
                if (keep) {
                    const uint32_t lrel = shift + (uint32_t)(ts.last_start - prev_end);
                    for (uint32_t b = tid; b < (uint32_t)last_len; b += DELTA_THREADS) st.prev[b] = st.obuf[lrel + b];
                }
                prev_in_lds = keep;
            } else {
                // direct path: every lane moves its own string through global memory
                if (tid < cnt) {
                    dba_assemble(obytes + my_start, ts, tid + 1, pfx, slen, prev_bytes,
                                 [&](uint32_t k) { return sdata + ts.sufoff[k]; });
                }
                prev_in_lds = last_len == 0;
            }
            __threadfence_block();    // the next tile may read this tile's last string back from global memory
        }
        prev_start = ts.last_start;
        prev_len = last_len;
        prev_end += tile_bytes;
        __syncthreads();      // the tile's bytes are written and ts / the stage can be reused
        DBA_TICK(5);
    }
    __syncthreads();
    if (tid == 0) {
        if (sh.err) write_status(&status[page], sh.err, 0, 0, 0, 0, 0);
        else if (!SIZES_ONLY && !can_write) write_status(&status[page], E_OUT_CAPACITY, limit, out_bytes, pd.param0, 0, out_bytes);
#ifdef P4S_EXP_TIMING      // cycles of phases A | B << 32, C
#ifdef P4S_EXP_TIMING_C     // phase C per part: scan | suffix load << 32, column loop | patch << 32 (store = C minus these)
        else write_status(&status[page], P4S_E_NONE, limit, (cacc[0] & 0xFFFFFFFFull) | (cacc[1] << 32),
                          (cacc[2] & 0xFFFFFFFFull) | (cacc[3] << 32), data_pos + ssink.all_suffix, out_bytes);
#else
        else write_status(&status[page], P4S_E_NONE, limit, ((tk_b - tk_a) & 0xFFFFFFFFull) | ((tk_c - tk_b) << 32),
                          __builtin_readcyclecounter() - tk_c, data_pos + ssink.all_suffix, out_bytes);
#endif
#else
        else write_status(&status[page], P4S_E_NONE, limit, 0, 0, data_pos + ssink.all_suffix, out_bytes);
#endif
    }
}

}  // namespace p4s
