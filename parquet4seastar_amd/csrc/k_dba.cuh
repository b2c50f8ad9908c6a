// DELTA_BYTE_ARRAY kernel (ref: delta_byte_array_decoder, src/encoding.cc:296-362; parity is
// with a single whole-page read_batch, the only call shape the reference decodes correctly).
//
// value_i = value_{i-1}[0 : prefix_i] ++ suffix_i.  Phases, one workgroup per page:
//   A. prefix lengths  = nested DELTA_BINARY_PACKED INT32 stream  -> scratch
//   B. suffix lengths  = second nested stream; two prefix sums per tile give every string its
//      suffix position and its output offset (offsets[] is final after this phase)
//   C. 256 strings at a time, one lane per string: copy the suffix, then resolve the shared
//      prefix by walking back over the earlier strings of the tile: string k supplies bytes
//      [prefix_k, need) from its own suffix and lowers `need` to prefix_k.  What is still missing
//      at the tile boundary is copied from the complete string just before the tile, so a
//      walk never leaves the tile.
#pragma once
#include "k_strings.cuh"

namespace p4s {

constexpr int DBA_TILE = 256;
static_assert(DBA_TILE == DELTA_THREADS, "one string per lane");

struct DbaTileShared {
    uint32_t prefix[DBA_TILE + 1];    // entry e (1-based) = string t0+e-1
    uint32_t sufoff[DBA_TILE + 2];    // suffix offset; [cnt+1] = end of the last suffix of the tile
    uint64_t last_start;              // output offset of the tile's last string
    uint64_t last_len;                // and its length
};

template <bool SIZES_ONLY>
__global__ __launch_bounds__(DELTA_THREADS) void k_delta_byte_array(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        int64_t *__restrict__ d_offsets, uint8_t *__restrict__ d_bytes, uint64_t bytes_cap,
        uint32_t *__restrict__ d_scratch, uint64_t scratch_slots, p4s_page_status *__restrict__ status) {
    __shared__ StringsShared sh;
    __shared__ DbaTileShared ts;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t tid = threadIdx.x;
    const uint64_t slot0 = pd.out_offset;
    const uint64_t slots = slot0 < scratch_slots ? scratch_slots - slot0 : 0;
    uint32_t *scratch = d_scratch + 2 * slot0;
    int64_t *offsets = SIZES_ONLY ? nullptr : d_offsets + slot0;
    if (tid == 0) sh.err = 0;

    // ---- A: every prefix length of the first stream (ref :331-345)
    DbaPrefixSink psink{scratch, slots};
    DeltaResult pres;
    delta_decode_stream<uint32_t>(sh.delta, pv, 0xFFFFFFFFull, true, psink, pres);
    if (pres.err) {
        if (tid == 0) write_status(&status[page], pres.err, 0, pres.aux0, pres.aux1, 0, 0);
        return;
    }
    __syncthreads();

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
    delta_decode_stream<uint32_t>(sh.delta, sv, 0xFFFFFFFFull, true, ssink, sres);
    if (sres.err) {
        if (tid == 0) write_status(&status[page], sres.err, 0, sres.aux0, sres.aux1, 0, 0);
        return;
    }
    if (!SIZES_ONLY && limit == 0 && tid == 0) offsets[0] = 0;
    const uint64_t data_pos = pres.end_pos + sres.end_pos;       // first suffix byte in the page
    const uint64_t data_len = pv.size - data_pos;
    if (ssink.all_suffix > data_len) {                            // ref :262-266, reached through :348-358
        if (tid == 0) write_status(&status[page], P4S_E_DLBA_VALUES_EOF, 0, 0, 0, 0, 0);
        return;
    }
    if (ssink.missing_prefix) {   // more suffixes than prefix lengths: the reference indexes past its array
        if (tid == 0) write_status(&status[page], P4S_E_DBA_BAD_PREFIX, 0, 0, 0, 0, 0);
        return;
    }
    const uint64_t out_bytes = ssink.out_carry;
    const uint64_t suffix_end = ssink.suf_carry;                  // end of the last emitted suffix
    bool can_write = !SIZES_ONLY;
    if (!SIZES_ONLY && (out_bytes > pd.param0 || pd.aux_offset + out_bytes > bytes_cap)) can_write = false;
    __syncthreads();

    // ---- C: validate prefix_i <= len(value_{i-1}) (ref :310-312) and rebuild the strings
    const uint8_t *sdata = pv.base + data_pos;
    uint8_t *obytes = SIZES_ONLY ? nullptr : d_bytes + pd.aux_offset;
    uint64_t prev_start = 0, prev_len = 0, prev_end = 0;   // the string before the tile
    uint32_t buf = 0;
    for (uint64_t t0 = 0; t0 < limit; t0 += DBA_TILE) {
        const uint64_t i = t0 + tid;
        const uint32_t cnt = (uint32_t)(limit - t0 < DBA_TILE ? limit - t0 : DBA_TILE);
        if (tid < cnt) {
            ts.prefix[tid + 1] = scratch[2 * i];
            ts.sufoff[tid + 1] = scratch[2 * i + 1];
            if (tid == cnt - 1)
                ts.sufoff[cnt + 1] = (i + 1 < limit) ? scratch[2 * (i + 1) + 1] : (uint32_t)suffix_end;
        }
        __syncthreads();
        uint32_t pfx = 0, slen = 0;
        if (tid < cnt) {
            pfx = ts.prefix[tid + 1];
            slen = ts.sufoff[tid + 2] - ts.sufoff[tid + 1];
        }
        const uint64_t my_len = (uint64_t)pfx + slen;
        uint64_t tile_bytes;
        const uint64_t my_start = prev_end + wg_exclusive_scan(sh.scan, buf, my_len, tile_bytes);
        if (tid < cnt) {
            const uint64_t before = tid == 0 ? prev_len
                                             : (uint64_t)ts.prefix[tid] + (ts.sufoff[tid + 1] - ts.sufoff[tid]);
            if ((uint64_t)pfx > before) atomicExch(&sh.err, (int32_t)P4S_E_DBA_BAD_PREFIX);
            if (tid == cnt - 1) { ts.last_start = my_start; ts.last_len = my_len; }
        }
        __syncthreads();
        if (sh.err) break;
        if (can_write && tid < cnt) {
            uint8_t *dst = obytes + my_start;
            const uint8_t *src = sdata + ts.sufoff[tid + 1];
            for (uint32_t b = 0; b < slen; ++b) dst[pfx + b] = src[b];
            uint32_t need = pfx;
            for (int32_t k = (int32_t)tid; need > 0 && k >= 1; --k) {
                const uint32_t pk = ts.prefix[k];
                if (pk < need) {
                    const uint8_t *sk = sdata + ts.sufoff[k];
                    for (uint32_t b = pk; b < need; ++b) dst[b] = sk[b - pk];
                    need = pk;
                }
            }
            if (need > 0) {
                const uint8_t *pb = obytes + prev_start;     // complete since the previous iteration
                for (uint32_t b = 0; b < need; ++b) dst[b] = pb[b];
            }
        }
        prev_start = ts.last_start;
        prev_len = ts.last_len;
        prev_end += tile_bytes;
        __syncthreads();      // the tile's bytes are written and ts can be reused
    }
    __syncthreads();
    if (tid == 0) {
        if (sh.err) write_status(&status[page], sh.err, 0, 0, 0, 0, 0);
        else if (!SIZES_ONLY && !can_write) write_status(&status[page], E_OUT_CAPACITY, limit, out_bytes, pd.param0, 0, out_bytes);
        else write_status(&status[page], P4S_E_NONE, limit, 0, 0, data_pos + ssink.all_suffix, out_bytes);
    }
}

}  // namespace p4s
