// DELTA_BYTE_ARRAY kernels (ref: delta_byte_array_decoder, src/encoding.cc:296-362; parity is
// with a single whole-page read_batch, the only call shape the reference decodes correctly).
//
// value_i = value_{i-1}[0 : prefix_i] ++ suffix_i.  Two kernels:
//
// k_dba_lengths, one workgroup per page: the two nested DELTA_BINARY_PACKED length streams and what follows from
// them.
//   A. prefix lengths -> scratch
//   B. suffix lengths -> scratch
//   S. one pass over the (prefix length, suffix length) pairs: suffix offsets (in place of the suffix lengths), output
//      offsets (offsets[] is final after this phase), the check prefix_i <= len(value_{i-1}) (ref :310-312) and the
//      smallest prefix length of every tile of 256 strings.
//   This is all the sizes pass needs.
//
// k_dba_assemble, many wavefronts per page, each with a run of consecutive strings.
//   A string depends on the string before it.  A wavefront that starts in the middle of a page rebuilds the part of
//   that string it can inherit (as many bytes as the prefix length of its own first string) by walking backwards: a
//   string contributes the bytes between its own prefix length and the smallest prefix length met since, and whole
//   tiles whose smallest prefix length is not below that bound are skipped through the table phase S left.  The
//   walk ends at the latest at the page's first string, whose prefix length is 0.
//   Then 64 strings at a time, one lane per string, four byte columns per step: a copy-forward scan over the lanes
//   (see the kernel).  Steps with a string longer than DBA_PREV_CAP take a lane-per-string path through global memory.
#pragma once
#include "k_strings.cuh"

namespace p4s {

constexpr int DBA_TILE = 1 << DBA_TILE_SHIFT;
static_assert(DBA_TILE == DELTA_THREADS, "one string per lane");
constexpr int DBA_PREV_CAP = 1024;    // bytes of the string before a step kept in LDS; longer strings take the slow path
#ifndef P4S_DBA_BATCH
#define P4S_DBA_BATCH 4
#endif
constexpr int DBA_BATCH = P4S_DBA_BATCH;   // column groups whose loads are in flight together
constexpr uint32_t DBA_MIN_STEPS = 16; // a wavefront of the assembly takes at least this many steps of 64 strings

// What the length kernel leaves for the assembly kernel besides the per-string scratch.
struct DbaPageMeta {
    uint64_t data_pos;      // page offset of the first suffix byte
    uint64_t suffix_end;    // suffix-data offset after the last emitted string's suffix
};
struct DbaAux {
    DbaPageMeta *meta;      // [page]
    uint32_t *tile_min;     // [slot >> 8] of the first slot of a full tile: its smallest prefix length
};

// LDS of dba_scan_page: per-wave totals of the two running sums, the length of each wave's last string (prefix
// check across waves), double buffered by the chunk's parity.
struct DbaScanShared {
    uint64_t wtot[2][2][4];
    uint64_t wlast[2][4];
};

__device__ __forceinline__ uint32_t wave_iminscan(uint32_t v) { return ~wave_imaxscan(~v); }

// The page's strings in chunks of 1024 (four consecutive strings per thread, a tile of 256 per wave): from the
// (prefix length, suffix length) pairs in scratch to suffix offsets (in place of the suffix lengths), output
// offsets, the smallest prefix length of every full tile, and the check prefix_i <= len(value_{i-1}) (ref
// src/encoding.cc:310-312; the first string has nothing before it).  Returns through the references what the whole
// workgroup agrees on.
template <bool SIZES_ONLY>
__device__ __forceinline__ void dba_scan_page(DbaScanShared &sh, const uint32_t *pfxs, uint32_t *sufs, int64_t *offsets, uint32_t *tile_min,
                                              uint64_t limit, uint64_t n_prefix, uint64_t &suffix_bytes, uint64_t &out_bytes,
                                              bool &missing_prefix, bool &bad_prefix) {
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    uint64_t suf_carry = 0, out_carry = 0, len_carry = 0;     // sums before the chunk; length of the string before it
    bool missing = false, bad = false;
    uint32_t buf = 0;
    // a thread's four (prefix length, suffix length) pairs: two 16-byte loads where all four strings exist
    auto fetch4 = [&](uint64_t i, uint2 (&d)[4]) {
        if (i + 4 <= limit) {
            uint32_t a[4], c[4];
            if (P4S_INB(pfxs + i, 16, 17)) __builtin_memcpy(a, pfxs + i, 16); else a[0] = a[1] = a[2] = a[3] = 0;
            if (P4S_INB(sufs + i, 16, 17)) __builtin_memcpy(c, sufs + i, 16); else c[0] = c[1] = c[2] = c[3] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) d[j] = make_uint2(a[j], c[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                d[j] = i + j < limit ? make_uint2(P4S_LD(pfxs + i + j, 17), P4S_LD(sufs + i + j, 17)) : make_uint2(0u, 0u);
        }
    };
    uint2 nx[4];
    fetch4((uint64_t)tid * 4, nx);
    if (tid == 0 && !SIZES_ONLY) P4S_ST(offsets, 0, 14);
    for (uint64_t c0 = 0; c0 < limit; c0 += 4 * DELTA_THREADS) {
        const uint64_t idx0 = c0 + (uint64_t)tid * 4;
        uint32_t p[4];
        uint64_t s[4], o[4];
        bool emit[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            emit[j] = idx0 + j < limit;
            p[j] = nx[j].x;
            if (emit[j] && idx0 + j >= n_prefix) { missing = true; p[j] = 0; }
            s[j] = emit[j] ? (uint64_t)nx[j].y : 0;
            o[j] = emit[j] ? s[j] + p[j] : 0;              // the string's length
        }
        fetch4(idx0 + 4 * DELTA_THREADS, nx);
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (emit[j] && (uint64_t)p[j] > o[j - 1]) bad = true;
        uint64_t prev_len = wave_shr1(o[3]);               // the string before this thread's first
        if (lane == 63) sh.wlast[buf][wave] = o[3];
        // the smallest prefix length of the wave's tile
        uint32_t mn = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) if (emit[j] && p[j] < mn) mn = p[j];
        mn = wave_iminscan(mn);
        s[1] += s[0]; s[2] += s[1]; s[3] += s[2];
        o[1] += o[0]; o[2] += o[1]; o[3] += o[2];
        uint64_t inc[2] = {s[3], o[3]};
        wave_iscan_n(inc);
        if (lane == 63) { sh.wtot[buf][0][wave] = inc[0]; sh.wtot[buf][1][wave] = inc[1]; }
        __syncthreads();
        uint64_t base_s = suf_carry + inc[0] - s[3], base_o = out_carry + inc[1] - o[3], tot_s = 0, tot_o = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint64_t ts = sh.wtot[buf][0][i], to = sh.wtot[buf][1][i];
            if ((uint32_t)i < wave) { base_s += ts; base_o += to; }
            tot_s += ts; tot_o += to;
        }
        if (lane == 0) prev_len = wave > 0 ? sh.wlast[buf][wave - 1] : len_carry;
        if (emit[0] && (uint64_t)p[0] > prev_len) bad = true;
        len_carry = sh.wlast[buf][3];                      // (only a full chunk has a chunk after it)
        if (emit[3]) {
            uint32_t so[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint64_t before = base_s + (j ? s[j - 1] : 0);
                so[j] = before > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)before;
            }
            if (P4S_INB(sufs + idx0, 16, 18)) __builtin_memcpy(sufs + idx0, so, 16);
            if (!SIZES_ONLY) {
                int64_t oo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) oo[j] = (int64_t)(base_o + o[j]);
                if (P4S_INB(offsets + idx0 + 1, 32, 19)) __builtin_memcpy(offsets + idx0 + 1, oo, 32);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (emit[j]) {
                    const uint64_t before = base_s + (j ? s[j - 1] : 0);
                    P4S_ST(sufs + idx0 + j, before > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)before, 18);
                    if (!SIZES_ONLY) P4S_ST(offsets + idx0 + j + 1, (int64_t)(base_o + o[j]), 19);
                }
            }
        }
        if (!SIZES_ONLY && lane == 63 && emit[3])          // the tile is full: its last string exists
            tile_min[idx0 >> DBA_TILE_SHIFT] = mn;      // (the library's own table: not one of the caller's buffers a checked build knows)
        suf_carry += tot_s;
        out_carry += tot_o;
        buf ^= 1u;
    }
    missing_prefix = __syncthreads_or(missing) != 0;
    bad_prefix = __syncthreads_or(bad) != 0;
    suffix_bytes = suf_carry;
    out_bytes = out_carry;
}

// What the two length streams of a page came to (k_dba_streams -> k_dba_scan).
struct DbaStreams {
    int32_t a_err, b_err;       // page error of the prefix-length / suffix-length stream
    uint32_t b_skipped;         // the suffix stream was not read: the prefix stream does not walk to its end (a_err says why)
    uint32_t b_capacity;        // ... or the page has too few scratch slots for `limit` strings
    uint64_t a_aux0, a_aux1, a_end, a_total;
    uint64_t b_aux0, b_aux1, b_end, all_suffix, limit;
    uint64_t pad[3];
};
static_assert(sizeof(DbaStreams) == 112, "side table layout");

// Two workgroups per page, side by side: the even one decodes the prefix lengths (ref :331-345), the odd one the suffix
// lengths (a DELTA_LENGTH_BYTE_ARRAY section, ref :347-358).  The suffix stream starts where the prefix stream ends:
// its workgroup first follows the prefix stream's block headers to their end without decoding anything, which takes a
// small part of a decode.
#ifndef P4S_DBA_STREAMS_WG
#define P4S_DBA_STREAMS_WG 8
#endif
__global__ __launch_bounds__(DELTA_THREADS, P4S_DBA_STREAMS_WG) void k_dba_streams(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint32_t *__restrict__ d_scratch, uint64_t scratch_slots, DbaStreams *__restrict__ recs) {
    __shared__ DeltaShared sh;
    __shared__ uint64_t s_part[4];
    const uint32_t page = blockIdx.x >> 1;
    const bool suffix_role = (blockIdx.x & 1u) != 0;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    const uint64_t slot0 = pd.out_offset;
    const uint64_t slots = slot0 < scratch_slots ? scratch_slots - slot0 : 0;
    DbaStreams *rec = recs + page;
    if (!suffix_role) {
        // every prefix length of the first stream; only the first num_values of them are used (and kept: the scratch
        // slots after them belong to the next page)
        DeltaStoreSink<uint32_t> psink{d_scratch + slot0, slots < pd.num_values ? slots : pd.num_values};
        DeltaResult pres;
        delta_decode_stream<uint32_t>(sh, pv, 0xFFFFFFFFull, true, psink, pres);
        if (tid == 0) {
            rec->a_err = pres.err; rec->a_aux0 = pres.aux0; rec->a_aux1 = pres.aux1;
            rec->a_end = pres.end_pos; rec->a_total = pres.total;
        }
        return;
    }
    DbaNullSink nsink;
    DeltaResult pres;
    delta_decode_stream<uint32_t, DbaNullSink, true>(sh, pv, 0xFFFFFFFFull, true, nsink, pres);
    if (tid == 0) { rec->b_skipped = pres.err ? 1u : 0u; rec->b_capacity = 0; rec->b_err = 0; }
    if (pres.err) return;
    __syncthreads();
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
    if (tid == 0) rec->limit = limit;
    // the page's offsets take limit + 1 slots (same slot numbering as the scratch): one that ends at the last slot the
    // caller declared would write its closing offset behind them
    if (limit + 1 > slots) {
        if (tid == 0) rec->b_capacity = 1u;
        return;
    }
    DbaSuffixSink ssink{d_scratch + scratch_slots + slot0, limit, 0};
    DeltaResult sres;
    delta_decode_stream<uint32_t>(sh, sv, 0xFFFFFFFFull, true, ssink, sres);
    // bytes of all suffixes: the threads' shares added up
    const uint64_t wsum = wave_iscan(ssink.all_suffix);
    __syncthreads();
    if ((tid & 63u) == 63u) s_part[tid >> 6] = wsum;
    __syncthreads();
    if (tid == 0) {
        rec->b_err = sres.err; rec->b_aux0 = sres.aux0; rec->b_aux1 = sres.aux1;
        rec->b_end = sres.end_pos;
        rec->all_suffix = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    }
}

// One workgroup per page: the page's status from what the two streams came to, and for a page without an error one pass
// over the (prefix length, suffix length) pairs (dba_scan_page).
template <bool SIZES_ONLY>
__global__ __launch_bounds__(DELTA_THREADS, 4) void k_dba_scan(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        int64_t *__restrict__ d_offsets, uint64_t bytes_cap,
        uint32_t *__restrict__ d_scratch, uint64_t scratch_slots, const DbaStreams *__restrict__ recs, DbaAux aux,
        p4s_page_status *__restrict__ status) {
    __shared__ DbaScanShared sh;
    const uint32_t page = blockIdx.x;
    const uint32_t tid = threadIdx.x;
    const DbaStreams rec = recs[page];
    // the order in which one reader of the page would have met them
    if (rec.a_err || rec.b_skipped) {
        if (tid == 0) write_status(&status[page], rec.a_err ? rec.a_err : (int32_t)P4S_E_DELTA_DATA_EOF, 0, rec.a_aux0, rec.a_aux1, 0, 0);
        return;
    }
    if (rec.b_capacity) {
        if (tid == 0) write_status(&status[page], E_OUT_CAPACITY, rec.limit, 0, 0, 0, 0);
        return;
    }
    if (rec.b_err) {
        if (tid == 0) write_status(&status[page], rec.b_err, 0, rec.b_aux0, rec.b_aux1, 0, 0);
        return;
    }
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint64_t slot0 = pd.out_offset;
    int64_t *offsets = SIZES_ONLY ? nullptr : d_offsets + slot0;
    const uint64_t limit = rec.limit;
    uint64_t suffix_end = 0, out_bytes = 0;
    bool missing_prefix = false, bad_prefix = false;
    dba_scan_page<SIZES_ONLY>(sh, d_scratch + slot0, d_scratch + scratch_slots + slot0, offsets, SIZES_ONLY ? nullptr : aux.tile_min + (slot0 >> DBA_TILE_SHIFT),
                              limit, rec.a_total, suffix_end, out_bytes, missing_prefix, bad_prefix);
    if (tid != 0) return;
    const uint64_t data_pos = rec.a_end + rec.b_end;              // first suffix byte in the page
    const uint64_t data_len = pv.size - data_pos;
    if (rec.all_suffix > data_len) {                              // ref :262-266, reached through :348-358
        write_status(&status[page], P4S_E_DLBA_VALUES_EOF, 0, 0, 0, 0, 0);
        return;
    }
    if (missing_prefix || bad_prefix) {    // more suffixes than prefix lengths: the reference indexes past its array; ref :310-312
        write_status(&status[page], P4S_E_DBA_BAD_PREFIX, 0, 0, 0, 0, 0);
        return;
    }
    if (!SIZES_ONLY) {
        if (out_bytes > pd.param0 || pd.aux_offset + out_bytes > bytes_cap) {
            write_status(&status[page], E_OUT_CAPACITY, limit, out_bytes, pd.param0, 0, out_bytes);
            return;
        }
        DbaPageMeta m;
        m.data_pos = data_pos;
        m.suffix_end = suffix_end;                                // end of the last emitted suffix
        aux.meta[page] = m;
    }
    write_status(&status[page], P4S_E_NONE, limit, 0, 0, data_pos + rec.all_suffix, out_bytes);
}

// ---------------------------------------------------------------------------
// assembly
// ---------------------------------------------------------------------------
struct __attribute__((packed)) DbaU64 { uint64_t v; };   // eight bytes at any byte address
struct __attribute__((packed)) DbaU32 { uint32_t v; };
struct __attribute__((packed)) DbaU16 { uint16_t v; };
constexpr int DBA_GROUP = 8;          // byte columns per step of the assembly (one 64-bit register per lane)

// Bytes [0, need) of string i0 of the page, all of which it shares with string i0 - 1, written to dst (LDS or global).
// One wavefront; every lane calls.  pfxs / sufs = the page's prefix lengths and suffix offsets, tile_min = the page's
// table of smallest prefix lengths, sdata / sroom = the suffix data and the bytes readable from it.
__device__ __forceinline__ void dba_inherit(uint8_t *dst, uint32_t need, uint64_t i0, const uint32_t *pfxs, const uint32_t *sufs,
                                            const uint32_t *tile_min, const uint8_t *sdata, int64_t sroom) {
    const uint32_t lane = lane_id();
    uint32_t hi = need;          // bytes [0, hi) are still unknown
    uint64_t k = i0;             // strings below k have not been looked at
    while (hi > 0 && k > 0) {
        // the strings k-1 down to the first of their tile, 64 at a time (four loads in flight)
        const uint64_t tbeg = (k - 1) & ~(uint64_t)(DBA_TILE - 1);
        uint2 d[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t s = (int64_t)k - 1 - (int64_t)lane - 64 * c;
            d[c] = s >= (int64_t)tbeg ? make_uint2(P4S_LD(pfxs + s, 45), P4S_LD(sufs + s, 45)) : make_uint2(0xFFFFFFFFu, 0u);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (hi == 0) break;
            const uint32_t p = d[c].x;
            const uint32_t inc = wave_iminscan(p);
            uint32_t e = wave_shr1(inc);          // smallest prefix length among the later strings of this group
            if (lane == 0 || e > hi) e = hi;
            uint64_t m = __ballot(p < e);         // these strings own bytes [p, e)
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const uint32_t pl = bcast_u32(p, l), el = bcast_u32(e, l), so = bcast_u32(d[c].y, l);
                for (uint32_t b = lane; b < el - pl; b += 64) {
                    const int64_t at = (int64_t)so + b;
                    P4S_ST(dst + pl + b, at < sroom ? P4S_LD(sdata + at, 46) : (uint8_t)0, 47);
                }
            }
            const uint32_t mn = bcast_u32(inc, 63);
            if (mn < hi) hi = mn;
        }
        k = tbeg;
        if (hi == 0 || k == 0) break;
        // the nearest earlier tile with a prefix length below hi; the tiles between cannot contribute
        uint64_t u = k >> DBA_TILE_SHIFT;         // tiles 0 .. u-1 remain
        for (;;) {
            const int64_t t = (int64_t)u - 1 - (int64_t)lane;
            const uint32_t mn = t >= 0 ? tile_min[t] : 0xFFFFFFFFu;
            const uint64_t mm = __ballot(mn < hi);
            if (mm) { u = u - 1 - (uint64_t)__builtin_ctzll(mm); break; }
            if (u <= 64) { u = 0; break; }        // (a checked page always has one: its first string's prefix is 0)
            u -= 64;
        }
        k = (u + 1) << DBA_TILE_SHIFT;
    }
}

// One lane's string through global memory: suffix bytes, then the shared prefix from the earlier strings of the
// group (entries e-1 .. 0), then from the string before the group.  pfxs / sufs = the scratch words of the group's
// first string.
__device__ __forceinline__ void dba_assemble(uint8_t *dst, const uint32_t *pfxs, const uint32_t *sufs, uint32_t e, uint32_t pfx, uint32_t slen,
                                             const uint8_t *prev_bytes, const uint8_t *sdata) {
    const uint8_t *src = sdata + sufs[e];
    for (uint32_t b = 0; b < slen; ++b) P4S_ST(dst + pfx + b, P4S_LD(src + b, 30), 31);
    uint32_t need = pfx;
    for (uint32_t k = e; need > 0 && k >= 1; --k) {
        const uint32_t pk = P4S_LD(pfxs + (k - 1), 32);
        if (pk < need) {
            const uint8_t *sk = sdata + P4S_LD(sufs + (k - 1), 32);
            for (uint32_t b = pk; b < need; ++b) P4S_ST(dst + b, P4S_LD(sk + (b - pk), 32), 33);
            need = pk;
        }
    }
    for (uint32_t b = 0; b < need; ++b) P4S_ST(dst + b, P4S_LD(prev_bytes + b, 34), 35);
}

#ifndef P4S_DBA_ASM_WAVES
#define P4S_DBA_ASM_WAVES 7   // wavefronts per SIMD the register budget of the assembly kernel is sized for (measured: 8 spills, 6 is slower)
#endif

// One wavefront per workgroup, with a run of consecutive strings of one page; fan workgroups per page (the
// descriptors live on the device, so the host picks the fan-out and a workgroup finds its run from the page's string
// count).  Nothing is shared between wavefronts: no barriers.
//
// 64 strings per step, one lane per string, four byte columns at a time.  Byte j of string i is its own suffix byte when
// j >= prefix_i and byte j of string i-1 otherwise: per column a copy-forward scan over the lanes.  A lane loads the
// dword its own suffix puts in columns j0..j0+3 straight from the page, M marks the bytes of it that are its own (the
// upper ones: j >= prefix), six DPP steps carry (M, bytes) across the wave, the later string winning, and what no string
// of the step defines comes from the string before the step (kept in LDS).  A byte at or past a string's end is never
// asked for by a later string (its prefix cannot exceed that length), so what the scan carries there does no harm.
// Each lane stores its dword to the output itself.  Columns go from the top down: a dword stored at j0 >= 4 may run
// past its string's end into the first bytes of the strings behind it, which their lanes store afterwards (columns below
// 4 are stored last; one wavefront's stores to an address keep their order); past the step's own strings a lane stores
// exact bytes.  Steps with a string longer than DBA_PREV_CAP take a lane-per-string path through global memory.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(P4S_DBA_ASM_WAVES, P4S_DBA_ASM_WAVES))) void k_dba_assemble(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        const int64_t *__restrict__ d_offsets, uint8_t *__restrict__ d_bytes,
        const uint32_t *__restrict__ d_scratch, uint64_t scratch_slots, DbaAux aux, const p4s_page_status *__restrict__ status,
        uint32_t fan) {
    __shared__ __attribute__((aligned(16))) uint8_t prev[2][DBA_PREV_CAP];    // the string before the step, by the step's parity
    const uint32_t page = blockIdx.x / fan, part = blockIdx.x - page * fan;
    if (status[page].code != P4S_E_NONE) return;
    const uint64_t limit = status[page].values_decoded;
    const uint64_t nsteps = (limit + 63) >> 6;
    uint64_t per = (nsteps + fan - 1) / fan;
    if (per < DBA_MIN_STEPS) per = DBA_MIN_STEPS;
    const uint64_t step0 = (uint64_t)part * per;
    if (step0 >= nsteps) return;
    const uint64_t step1 = step0 + per < nsteps ? step0 + per : nsteps;
    const uint64_t i_begin = step0 << 6;
    const uint64_t i_end = step1 << 6 < limit ? step1 << 6 : limit;

    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t lane = threadIdx.x;
    const uint64_t slot0 = pd.out_offset;
    const uint32_t *pfxs = d_scratch + slot0, *sufs = d_scratch + scratch_slots + slot0;
    const DbaPageMeta meta = aux.meta[page];
    const uint8_t *sdata = pv.base + meta.data_pos;
    const int64_t sroom = (int64_t)pv.room - (int64_t)meta.data_pos;   // bytes readable from sdata
    const uint32_t suffix_end = meta.suffix_end > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)meta.suffix_end;
    uint8_t *obytes = d_bytes + pd.aux_offset;

    auto fetch_pair = [&](uint64_t i) -> uint2 {
        if (i < limit) return make_uint2(P4S_LD(pfxs + i, 37), P4S_LD(sufs + i, 37));
        return make_uint2(0u, suffix_end);
    };

    // ---- the string before the first step, as far as this run can inherit from it
    uint64_t out_pos = (uint64_t)P4S_LD(d_offsets + slot0 + i_begin, 49);     // output offset of the step's first byte
    uint32_t par = 0;
    const uint8_t *prev_global = nullptr;                  // set while the string before the step is not in LDS
    if (step0 > 0) {
        const uint32_t need = P4S_LD(pfxs + i_begin, 50);                    // prefix length of the run's first string
        // too long for LDS: the first string's own place in the output holds the same bytes
        if (need > (uint32_t)DBA_PREV_CAP) prev_global = obytes + out_pos;
        dba_inherit(prev_global ? obytes + out_pos : prev[0], need, i_begin, pfxs, sufs,
                    aux.tile_min + (slot0 >> DBA_TILE_SHIFT), sdata, sroom);
        __threadfence_block();
    }

    // The eight bytes a lane's own suffix puts in columns j0 .. j0+7 (0 when none of them is its own).
    auto gather = [&](int32_t j0, uint32_t pfx, uint32_t len, int64_t a0) -> uint64_t {
        uint64_t V = 0;
        if (j0 >= 0 && (int32_t)pfx - j0 < DBA_GROUP && (uint32_t)j0 < len) {
            const int64_t a = a0 + j0;
            if (a >= 0 && a + DBA_GROUP <= sroom) {
                V = P4S_LD(reinterpret_cast<const DbaU64 *>(sdata + a), 51).v;
            } else {
#pragma unroll
                for (int u = 0; u < DBA_GROUP; ++u)
                    if (a + u >= 0 && a + u < sroom) V |= (uint64_t)P4S_LD(sdata + a + u, 52) << (8 * u);
            }
        }
        return V;
    };
    // What a step needs of its strings: per lane the prefix length, the length, the offset in the step's output and
    // the first DBA_BATCH column groups of its own suffix (loads in flight); per step the count, the longest length and
    // the bytes.  A step with a string over DBA_PREV_CAP bytes only gets its count and longest length.
    struct Step {
        uint32_t pfx, len, rel;
        uint64_t g[DBA_BATCH];
        int64_t a0;
        uint32_t cnt, wmax, bytes;
    };
    constexpr uint32_t GMASK = ~(uint32_t)(DBA_GROUP - 1);
    auto setup = [&](uint64_t t0, const uint2 &m, const uint2 &a, Step &r) {
        r.cnt = t0 < i_end ? (uint32_t)(i_end - t0 < 64 ? i_end - t0 : 64) : 0u;
        const bool have = lane < r.cnt;
        r.pfx = have ? m.x : 0u;
        const uint32_t slen = have ? a.y - m.y : 0u;
        const uint64_t l64 = (uint64_t)r.pfx + slen;
        r.wmax = bcast_u32(wave_imaxscan(l64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)l64), 63);
        r.len = (uint32_t)l64;
        r.a0 = (int64_t)m.y - (int64_t)r.pfx;          // suffix-data offset of the string's byte 0 (were it all suffix)
        r.rel = 0; r.bytes = 0;
        if (r.wmax <= (uint32_t)DBA_PREV_CAP) {
            const uint32_t inc = wave_iscan(r.len);
            r.rel = inc - r.len;
            r.bytes = bcast_u32(inc, 63);
            const int32_t jtop = (int32_t)((r.wmax - 1u) & GMASK);
#pragma unroll
            for (int k = 0; k < DBA_BATCH; ++k) r.g[k] = gather(jtop - DBA_GROUP * k, r.pfx, r.len, r.a0);
        }
    };

    // (Issuing a step's first loads before the stores of the step before it, so that no load queues behind a store,
    // was measured and lost: the extra registers cost more than the queueing.)
    Step cur;
    // fetched one step ahead: a lane's own pair and, for its suffix length, the suffix offset of the string after it
    // (the lane after it holds that; the last lane needs the first string of the next step)
    auto fetch_after = [&](uint64_t t, const uint2 &own) -> uint2 {
        const uint64_t i = t + 64;
        const uint32_t edge = i < limit ? P4S_LD(sufs + i, 37) : suffix_end;      // (the same address in every lane)
        uint32_t nxt = dpp_mov<0x130, 0xF>(own.y);                                // wavefront shift left by one
        if (lane == 63) nxt = edge;
        return make_uint2(0u, nxt);
    };
    uint2 mine = fetch_pair(i_begin + lane), after = fetch_after(i_begin, mine);
    for (uint64_t t0 = i_begin; t0 < i_end; t0 += 64) {
        setup(t0, mine, after, cur);
        mine = fetch_pair(t0 + 64 + lane);
        after = fetch_after(t0 + 64, mine);
        const uint32_t cnt = cur.cnt, wmax = cur.wmax;
        const bool have = lane < cnt;
        if (wmax <= (uint32_t)DBA_PREV_CAP) {
            const uint32_t pfx = cur.pfx, len32 = cur.len;
            uint8_t *dst = obytes + out_pos + cur.rel;                // this lane's string in the output
            if (prev_global) {
                // (after a step of long strings) the bytes this step can inherit, into LDS
                const uint32_t need = bcast_u32(pfx, 0);
                for (uint32_t b = lane; b < need; b += 64) prev[par][b] = P4S_LD(prev_global + b, 58);
                prev_global = nullptr;
            }
            const uint8_t *pv_in = prev[par];
            uint8_t *pv_out = prev[par ^ 1u];
            const uint32_t room = cur.bytes - cur.rel;                // bytes up to the end of the step's strings
            // one group of eight columns: the scan, the string before the step, the store
            auto column = [&](int32_t j0, uint64_t V) {
                const int32_t t = (int32_t)pfx - j0;                  // own bytes start at byte t of the group
                uint64_t M = !have ? 0ull : t <= 0 ? ~0ull : t >= DBA_GROUP ? 0ull : ~0ull << (8 * t);
                // (when every string of the step owns the whole group, nothing is carried between lanes)
                if (__ballot(M != ~0ull && have) != 0) {
#define P4S_DBA_STEP(CTRL, MASK)                                                               \
                    {                                                                          \
                        const uint64_t va = dpp_mov64<CTRL, MASK>(V), ma = dpp_mov64<CTRL, MASK>(M); \
                        V = (V & M) | (va & ~M);                                               \
                        M |= ma;                                                               \
                    }
                    P4S_DBA_STEP(0x111, 0xF) P4S_DBA_STEP(0x112, 0xF) P4S_DBA_STEP(0x114, 0xF) P4S_DBA_STEP(0x118, 0xF)
                    P4S_DBA_STEP(0x142, 0xA) P4S_DBA_STEP(0x143, 0xC)
#undef P4S_DBA_STEP
                    // what no string of the step defines: from the string before the step
                    V = (V & M) | (*reinterpret_cast<const uint64_t *>(pv_in + j0) & ~M);
                }
                if ((uint32_t)j0 < len32) {
                    // the step's last string is the string before the next step
                    if (lane == cnt - 1) *reinterpret_cast<uint64_t *>(pv_out + j0) = V;
                    const uint32_t upto = (uint32_t)j0 + DBA_GROUP;
                    if (upto <= len32 || (j0 > 0 && upto <= room)) {
                        DbaU64 w8;
                        w8.v = V;
                        P4S_ST(reinterpret_cast<DbaU64 *>(dst + j0), w8, 53);
                    } else {
                        // exact: four, two and one byte
                        const uint32_t left = len32 - (uint32_t)j0;   // 1 .. 7
                        uint32_t at = 0;
                        if (left & 4u) { DbaU32 w4; w4.v = (uint32_t)V; P4S_ST(reinterpret_cast<DbaU32 *>(dst + j0), w4, 54); at = 4; }
                        if (left & 2u) { DbaU16 w2; w2.v = (uint16_t)(V >> (8 * at)); P4S_ST(reinterpret_cast<DbaU16 *>(dst + j0 + at), w2, 54); at += 2; }
                        if (left & 1u) P4S_ST(dst + j0 + at, (uint8_t)(V >> (8 * at)), 54);
                    }
                }
            };
            int32_t jtop = (int32_t)((wmax - 1u) & GMASK);
#pragma unroll
            for (int k = 0; k < DBA_BATCH; ++k)
                if (jtop - DBA_GROUP * k >= 0) column(jtop - DBA_GROUP * k, cur.g[k]);
            // longer strings: the remaining groups, DBA_BATCH loads at a time
            for (jtop -= DBA_GROUP * DBA_BATCH; jtop >= 0; jtop -= DBA_GROUP * DBA_BATCH) {
                uint64_t g[DBA_BATCH];
#pragma unroll
                for (int k = 0; k < DBA_BATCH; ++k) g[k] = gather(jtop - DBA_GROUP * k, pfx, len32, cur.a0);
#pragma unroll
                for (int k = 0; k < DBA_BATCH; ++k)
                    if (jtop - DBA_GROUP * k >= 0) column(jtop - DBA_GROUP * k, g[k]);
            }
            par ^= 1u;
            out_pos += cur.bytes;
        } else {
            // ---- a string longer than the LDS copy of the string before the step: every lane moves its own string
            // through global memory
            const uint2 m = fetch_pair(t0 + lane), a = fetch_pair(t0 + lane + 1);
            const uint32_t pfx = have ? m.x : 0u, slen = have ? a.y - m.y : 0u;
            const uint64_t my_len = (uint64_t)pfx + slen;
            const uint64_t inc = wave_iscan(my_len);
            const uint64_t step_bytes = bcast_u64(inc, 63);
            const uint8_t *pb = prev_global ? prev_global : prev[par];
            if (have) dba_assemble(obytes + out_pos + (inc - my_len), pfxs + t0, sufs + t0, lane, pfx, slen, pb, sdata);
            __threadfence_block();        // the next step reads this step's last string back from global memory
            const uint64_t last_len = bcast_u64(my_len, (int)((cnt - 1) & 63u));
            prev_global = obytes + out_pos + step_bytes - last_len;
            out_pos += step_bytes;
        }
    }
}

}  // namespace p4s
