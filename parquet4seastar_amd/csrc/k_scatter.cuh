// Positional column from a decoded page: definition levels + dense non-null values ->
// one value slot per level (nulls zero-filled) + a validity bitmap.  This is what the consumer
// of the reference's triplets does value by value (record_reader.hh:449-466 hands out a value
// only where def == max_def, column_chunk_reader.hh:134-140 counts them); here it is one
// stream-compaction-in-reverse per page on data already in HBM.
//
// One 256-thread workgroup per page, tiles of 1024 slots.  Inside a wave the four slots of a lane
// are 64 apart, so one ballot per row is 64 consecutive validity bits (two bitmap words, no
// shuffles) and every load and store of a row is 64 consecutive elements.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int SCATTER_THREADS = 256;
constexpr int SCATTER_ROWS = 4;                                   // slots per lane per tile
constexpr int SCATTER_TILE = SCATTER_THREADS * SCATTER_ROWS;      // 1024 slots

// pd.out_offset = first slot of the page in d_def / d_values / d_out (the layout column_chunk_decoder uses),
// pd.num_values = levels of the page, pd.param0 = max definition level, pd.aux_offset = first 32-bit word of the
// page's validity bitmap (bit s of the page = slot s; the page owns ceil(num_values / 32) words),
// pd.param1 = number of dense values available (values_decoded of the value decode).
// status.values_decoded = levels processed, status.aux0 = non-null count, status.aux1 = 1 when the page holds
// more non-null levels than dense values (the reference reports that as "Number of values in batch ... less than
// indicated by def levels", column_chunk_reader.hh:141-144).
template <int ELEM>
__global__ __launch_bounds__(SCATTER_THREADS) void k_scatter_by_levels(
        const int32_t *__restrict__ d_def, const uint8_t *__restrict__ d_values, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, uint32_t *__restrict__ d_validity, p4s_page_status *__restrict__ status) {
    __shared__ uint32_t wcount[2][4];
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint64_t n = pd.num_values;
    const uint32_t max_def = pd.param0;
    const int32_t *def = d_def + pd.out_offset;
    const uint8_t *vals = d_values + pd.out_offset * ELEM;
    uint8_t *out = d_out + pd.out_offset * ELEM;
    uint32_t *bitmap = d_validity + pd.aux_offset;
    uint64_t carry = 0;          // non-null slots before the tile
    uint32_t buf = 0;
    for (uint64_t t0 = 0; t0 < n; t0 += SCATTER_TILE) {
        const uint64_t w0 = t0 + (uint64_t)wave * (64 * SCATTER_ROWS);       // first slot of this wave's 256
        uint64_t ballots[SCATTER_ROWS];
        uint32_t total = 0;
#pragma unroll
        for (int j = 0; j < SCATTER_ROWS; ++j) {
            const uint64_t s = w0 + (uint64_t)j * 64 + lane;
            const bool valid = s < n && max_def != 0 ? (uint32_t)def[s] == max_def : s < n;
            ballots[j] = __ballot(valid);
            total += (uint32_t)__popcll(ballots[j]);
        }
        if (lane == 0) wcount[buf][wave] = total;
        __syncthreads();
        uint64_t before = carry;
        uint32_t tile_total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = wcount[buf][i];
            if ((uint32_t)i < wave) before += c;
            tile_total += c;
        }
#pragma unroll
        for (int j = 0; j < SCATTER_ROWS; ++j) {
            const uint64_t s = w0 + (uint64_t)j * 64 + lane;
            const uint64_t b = ballots[j];
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
            const bool valid = (b >> lane) & 1ull;
            const uint64_t src = before + rank;
            if (s < n) {
                const bool have = valid && src < pd.param1;
                if constexpr (ELEM == 4) {
                    reinterpret_cast<uint32_t *>(out)[s] = have ? reinterpret_cast<const uint32_t *>(vals)[src] : 0u;
                } else if constexpr (ELEM == 8) {
                    reinterpret_cast<uint64_t *>(out)[s] = have ? reinterpret_cast<const uint64_t *>(vals)[src] : 0ull;
                } else {
                    for (int k = 0; k < ELEM; ++k) out[s * ELEM + k] = have ? vals[src * ELEM + k] : (uint8_t)0;
                }
            }
            // 64 validity bits of the row: two words, written by lanes 0 and 1
            const uint64_t row0 = w0 + (uint64_t)j * 64;
            if (lane < 2 && row0 + lane * 32 < n) bitmap[(row0 >> 5) + lane] = (uint32_t)(b >> (32 * lane));
            before += (uint32_t)__popcll(b);
        }
        carry += tile_total;
        buf ^= 1u;
    }
    if (tid == 0) write_status(&status[page], P4S_E_NONE, n, carry, carry > pd.param1 ? 1 : 0, 0, 0);
}

}  // namespace p4s
