// RLE / bit-packed hybrid ENCODER (ref: RleEncoder, include/parquet4seastar/rle_encoding.hh:150-505: Put() for every
// value, then Flush(); used for levels, include/parquet4seastar/encoding.hh:205-240, and for dictionary indices,
// src/encoding.cc:783-790).  One 256-thread workgroup per page; the output is the hybrid stream alone (no length
// prefix, no bit-width byte).
//
// What the reference's automaton does, said without the automaton.  Values are looked at in groups of eight, counted
// from wherever the last repeated run ended (:352-382: the buffer restarts behind FlushRepeatedRun).  A group whose
// eight values are equal opens a repeated run, which then takes every following value that is still equal (:358-364)
// and is written as [ULEB128 count << 1 | value in ceil(width / 8) bytes] (:414-427).  Any other group is eight
// bit-packed literals (width bytes); literal groups that follow each other share an indicator byte (groups << 1 | 1)
// up to 63 groups (:431-461).  At the end (:463-484) fewer than eight values are left: a repeated run if no literal
// run is open and they are all equal, else a literal group padded with zeros.
// So the only serial facts are where the groups start and how many literal groups the open indicator byte holds, and
// both only change at repeated runs: with one bit per value that says "differs from the value before it", a group
// start p opens a repeated run iff bits p+1 .. p+7 are clear, the run ends at the next set bit, and between two runs
// the groups are eight apart.  The chain is therefore "end of a repeated run -> end of the next one".  Per chunk of
// 2048 values: all threads stage the values (fetched a chunk ahead) and make the bit words (ballots); for every place
// a run of eight or more equal values can end at, all threads work out the next run when the groups are counted from
// there; wave 0 follows the chain through that table, one lookup per run, and lists the runs; then, 64 runs at a
// time and a lane each, the stretch of literal groups in front of a run, byte offsets, the run's header, the
// stretch's indicator bytes and its entry in the packing table; then all threads pack the literal groups, and the
// chunk's bytes leave LDS in 16-byte stores.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int RLEENC_THREADS = 256;
constexpr int RLEENC_CHUNK = 2048;                       // values per chunk (32 words of 64: a thread per word and alignment in the table pass)
constexpr int RLEENC_WORDS = RLEENC_CHUNK / 64;
constexpr int RLEENC_MAXSEG = RLEENC_CHUNK / 8 / 2 + 8;  // stretches of literal groups per chunk (a repeated run of >= 8 values lies between two)
// bytes a chunk can produce: 256 literal groups of up to 32 bytes with an indicator byte each at worst, a run that
// ends here, the padded last group; + 16 for the alignment of the first byte
constexpr int RLEENC_STAGE = (RLEENC_CHUNK / 8) * 33 + 64 + 16;
constexpr int32_t E_ENCODE_RANGE = 35;                   // P4S_E_ENCODE_RANGE

struct RleEncShared {
    alignas(16) uint32_t val[RLEENC_CHUNK + 16];         // the chunk's values and the 16 behind it (0 past the page's end)
    uint64_t chg[RLEENC_WORDS + 2];                      // bit i: value i differs from the value before it (set past the page's end)
    uint64_t eq8[RLEENC_WORDS];                          // bit i: values i .. i+7 are equal and inside the page
    alignas(16) uint8_t stage[RLEENC_STAGE];             // the chunk's output bytes
    uint16_t seg_pos[RLEENC_MAXSEG];                     // a stretch of literal groups: chunk offset of its first value,
    uint16_t seg_cum[RLEENC_MAXSEG + 1];                 //   groups of the chunk before it (seg_cum[nseg] = all),
    uint16_t seg_off[RLEENC_MAXSEG];                     //   stage offset of its first byte (an indicator byte when g0 == 0),
    uint16_t seg_g0[RLEENC_MAXSEG];                      //   groups the open indicator byte held before it
    uint32_t ev[RLEENC_CHUNK / 8 + 8];                   // the chunk's repeated runs as the walk lists them
    uint32_t clsmask[8];                                 // per alignment c: the words of eq8 that hold a bit at a position = c mod 8
    uint64_t chgmask;                                    // the words of chg that are not zero
    uint32_t nseg;
    uint32_t out_lo, out_hi;                             // stage offsets of the chunk's first byte and behind its last
    uint32_t bad_value;
    // the walker's state between chunks
    uint64_t pos;            // next value to look at (a group start, or inside the open repeated run)
    uint64_t run_start;      // repeated run: its first value's index
    uint32_t run_value;
    uint32_t in_run;
    uint32_t groups;         // literal groups under the open indicator byte (0: none open)
    uint64_t ind_at;         // output offset of that byte
    uint64_t o;              // output bytes so far (whether or not they fit the slot)
};

// Eight values of W bits (already checked to fit) from LDS to W bytes at `dst` (any alignment), LSB first as
// BitWriter::PutValue packs them (bit_stream_utils.hh:191-211): fields at compile-time bit positions, dword stores.
template <int W>
__device__ __forceinline__ void rle_pack8(const uint32_t *v, uint8_t *dst) {
    if constexpr (W > 0) {
        constexpr int NW = (W + 3) / 4;
        constexpr uint32_t M = W == 32 ? 0xFFFFFFFFu : ((1u << (W & 31)) - 1u);
        uint32_t x[8];
        __builtin_memcpy(x, v, 32);
        uint32_t wd[NW];
#pragma unroll
        for (int i = 0; i < NW; ++i) wd[i] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lo = j * W, i = lo >> 5, sft = lo & 31;
            const uint32_t f = x[j] & M;
            wd[i] |= f << sft;
            if (sft + W > 32) wd[i + 1] |= f >> (32 - sft);
        }
#pragma unroll
        for (int i = 0; i < W / 4; ++i) __builtin_memcpy(dst + 4 * i, &wd[i], 4);
        if constexpr ((W & 3) >= 2) { const uint16_t h = (uint16_t)wd[W / 4]; __builtin_memcpy(dst + 4 * (W / 4), &h, 2); }
        if constexpr (W & 1) dst[W - 1] = (uint8_t)(wd[W / 4] >> (8 * ((W & 3) - 1)));
    }
}

__device__ __forceinline__ void rle_pack8_w(uint32_t w, const uint32_t *v, uint8_t *dst) {
    switch (w) {
#define P4S_PACK_CASE(W) case W: rle_pack8<W>(v, dst); break;
        P4S_PACK_CASE(1) P4S_PACK_CASE(2) P4S_PACK_CASE(3) P4S_PACK_CASE(4) P4S_PACK_CASE(5) P4S_PACK_CASE(6) P4S_PACK_CASE(7)
        P4S_PACK_CASE(8) P4S_PACK_CASE(9) P4S_PACK_CASE(10) P4S_PACK_CASE(11) P4S_PACK_CASE(12) P4S_PACK_CASE(13) P4S_PACK_CASE(14)
        P4S_PACK_CASE(15) P4S_PACK_CASE(16) P4S_PACK_CASE(17) P4S_PACK_CASE(18) P4S_PACK_CASE(19) P4S_PACK_CASE(20) P4S_PACK_CASE(21)
        P4S_PACK_CASE(22) P4S_PACK_CASE(23) P4S_PACK_CASE(24) P4S_PACK_CASE(25) P4S_PACK_CASE(26) P4S_PACK_CASE(27) P4S_PACK_CASE(28)
        P4S_PACK_CASE(29) P4S_PACK_CASE(30) P4S_PACK_CASE(31) P4S_PACK_CASE(32)
        default: break;                     // width 0: a literal group has no bytes
#undef P4S_PACK_CASE
    }
}

// values: uint32, d_pages[i].in_offset / num_values in ELEMENTS, out_offset in bytes, param0 = capacity of the output
// slot, param1 = bit width (0 .. 32).
__global__ __launch_bounds__(RLEENC_THREADS, 2048 / RLEENC_THREADS) void k_rle_encode(
        const uint32_t *__restrict__ values, uint64_t n_values, const p4s_page_desc *__restrict__ pages,
        uint8_t *__restrict__ d_out, uint64_t out_bytes, p4s_page_status *__restrict__ status) {
    __shared__ RleEncShared sh;
    const uint32_t page = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const p4s_page_desc pd = pages[page];
    const uint32_t w = (uint32_t)pd.param1;
    uint64_t n = pd.num_values;
    const uint64_t in0 = pd.in_offset < n_values ? pd.in_offset : n_values;
    if (n > n_values - in0) n = n_values - in0;
    if (n > 0xFFFFFFF0ull) n = 0xFFFFFFF0ull;            // (a count is a 32-bit ULEB128 shifted by one: pages this long do not exist)
    if (w > 32) {
        if (tid == 0) write_status(&status[page], E_ENCODE_RANGE, 0, w, 0, 0, 0);
        return;
    }
    const uint32_t *in = values + in0;
    const uint64_t room = pd.out_offset < out_bytes ? out_bytes - pd.out_offset : 0;
    const uint64_t cap = pd.param0 < room ? pd.param0 : room;
    uint8_t *out = d_out + pd.out_offset;
    const uint32_t vbytes = (w + 7) >> 3;
    const uint32_t vmask = w >= 32 ? 0xFFFFFFFFu : ((1u << w) - 1u);
    if (tid == 0) {
        sh.pos = 0; sh.run_start = 0; sh.run_value = 0; sh.in_run = 0; sh.groups = 0; sh.ind_at = 0; sh.o = 0; sh.bad_value = 0;
    }
    uint32_t prev_last = 0;                              // the value before the chunk (every thread keeps it)
    __syncthreads();
#ifdef P4S_EXP_TIMING      // cycles per page: staging | bit words | listing the runs | entries | packing | flush
    uint64_t tacc[6] = {0, 0, 0, 0, 0, 0}, tlast = __builtin_readcyclecounter();
#define ENC_TICK(i) do { const uint64_t n_ = __builtin_readcyclecounter(); tacc[i] += n_ - tlast; tlast = n_; } while (0)
#else
#define ENC_TICK(i) do { } while (0)
#endif
    // The values are one chunk ahead of the work, in registers: pieces of four values tid and tid + THREADS ... of the
    // chunk, and (threads 0 .. 3) the 16 values behind it; zero past the page's end.
    constexpr int NPIECE = RLEENC_CHUNK / 4 / RLEENC_THREADS;
    uint4 nx[NPIECE + 1];
    auto fetch = [&](uint64_t cb) {
#pragma unroll
        for (int k = 0; k <= NPIECE; ++k) {
            nx[k] = make_uint4(0, 0, 0, 0);
            if (k == NPIECE && tid >= 4) continue;
            const uint64_t at = cb + 4ull * (tid + (uint32_t)RLEENC_THREADS * k);
            if (at + 4 <= n) {
                if (P4S_INB(in + at, 16, 130)) __builtin_memcpy(&nx[k], in + at, 16);
            } else if (at < n) {
                uint32_t t[4];
                for (int e = 0; e < 4; ++e) t[e] = at + e < n ? P4S_LD(in + at + e, 130) : 0u;
                nx[k] = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
    };
    fetch(0);
    for (uint64_t c0 = 0; c0 < n; c0 += RLEENC_CHUNK) {
        // the chunk that leaves fewer than eight values behind it also takes those: they are the page's last group
        const bool last_chunk = n - c0 <= (uint64_t)RLEENC_CHUNK + 7;
        // ---- the chunk's values (+ 16) to LDS, the next chunk's on their way
        {
            uint32_t over = 0;
#pragma unroll
            for (int k = 0; k < NPIECE; ++k) {
                *reinterpret_cast<uint4 *>(&sh.val[4u * (tid + (uint32_t)RLEENC_THREADS * k)]) = nx[k];
                over |= nx[k].x | nx[k].y | nx[k].z | nx[k].w;
            }
            if (tid < 4) *reinterpret_cast<uint4 *>(&sh.val[RLEENC_CHUNK + 4u * tid]) = nx[NPIECE];
            if (over & ~vmask) sh.bad_value = 1;
            if (!last_chunk) fetch(c0 + RLEENC_CHUNK);
        }
        __syncthreads();
        ENC_TICK(0);
        // ---- "differs from the value before it", a 64-bit word per ballot
#pragma unroll
        for (int it = 0; it < (RLEENC_WORDS + 2 + RLEENC_THREADS / 64 - 1) / (RLEENC_THREADS / 64); ++it) {
            const uint32_t j = wave + (uint32_t)it * (RLEENC_THREADS / 64);
            if (j >= RLEENC_WORDS + 2) break;
            const uint32_t i = j * 64 + lane;
            bool d = true;                               // past the page's end: set
            if (i < RLEENC_CHUNK + 16 && c0 + i < n) {
                const uint32_t before = i ? sh.val[i - 1] : prev_last;
                d = (c0 + i == 0) || sh.val[i] != before;
            }
            const uint64_t m = __ballot(d);
            if (lane == 0) sh.chg[j] = m;
        }
        prev_last = sh.val[RLEENC_CHUNK - 1];
        __syncthreads();
        if (tid < RLEENC_WORDS) {
            const uint64_t a = sh.chg[tid], b = sh.chg[tid + 1];
            uint64_t any = 0;
#pragma unroll
            for (int k = 1; k <= 7; ++k) any |= (a >> k) | (b << (64 - k));
            sh.eq8[tid] = ~any;
        }
        __syncthreads();
        // ---- for every value that differs from the one before it (every place a repeated run can end at, and so every
        // place from which the next one is looked for): the next repeated run when the groups are counted from there.
        // With the words that hold anything marked, the search is two words long at most.
        static_assert(RLEENC_THREADS == 256 && RLEENC_WORDS == 32, "eight alignments x 32 words, a thread each");
        {
            const bool has = (sh.eq8[tid & 31u] & (0x0101010101010101ull << (tid >> 5))) != 0;
            const uint64_t bm = __ballot(has);                 // wave v: alignments 2 v (low half) and 2 v + 1
            if (lane == 0) { sh.clsmask[2 * wave] = (uint32_t)bm; sh.clsmask[2 * wave + 1] = (uint32_t)(bm >> 32); }
            if (wave == 0) {
                const uint64_t cm = __ballot(lane < RLEENC_WORDS + 2 && sh.chg[lane < RLEENC_WORDS + 2 ? lane : 0u] != 0);
                if (lane == 0) sh.chgmask = cm;
            }
        }
        __syncthreads();
        const uint64_t left_ = n - c0;
        const uint32_t glim_ = left_ >= 8 ? (left_ - 7 < RLEENC_CHUNK ? (uint32_t)(left_ - 7) : (uint32_t)RLEENC_CHUNK) : 0u;   // whole groups start below it
        // start of the next repeated run at or after x on x's alignment | its end << 16 (0xFFFF: none below glim)
        auto find_run = [&](uint32_t x) -> uint32_t {
            const uint32_t c = x & 7u;
            const uint64_t cls = 0x0101010101010101ull << c;
            uint32_t w0 = x >> 6;
            uint64_t m = sh.eq8[w0] & cls & (~0ull << (x & 63u));
            if (!m) {
                const uint32_t rest = w0 + 1 < (uint32_t)RLEENC_WORDS ? sh.clsmask[c] & (~0u << (w0 + 1)) : 0u;
                if (rest) { w0 = (uint32_t)__builtin_ctz(rest); m = sh.eq8[w0] & cls; }
            }
            if (!m) return 0xFFFFFFFFu;
            const uint32_t p = w0 * 64 + (uint32_t)__builtin_ctzll(m);
            if (p >= glim_) return 0xFFFFFFFFu;
            const uint32_t y = p + 1;                            // (its next seven values are equal)
            uint32_t w1 = y >> 6;
            uint64_t m2 = sh.chg[w1] & (~0ull << (y & 63u));
            if (!m2) {                                           // (the last word is all ones: past the values)
                w1 = (uint32_t)__builtin_ctzll(sh.chgmask & (~0ull << (w1 + 1)));
                m2 = sh.chg[w1];
            }
            return p | ((w1 * 64 + (uint32_t)__builtin_ctzll(m2)) << 16);
        };
        uint32_t *tab = reinterpret_cast<uint32_t *>(sh.stage);          // (the stage is written after the walk)
        static_assert(RLEENC_STAGE >= RLEENC_CHUNK * 4, "the table of next runs fits the stage");
        {
            // (only where a run of eight or more equal values ends: the value differs and the eight before it are
            // equal; the ends among a chunk's first eight values belong to runs that came in from the chunk before,
            // and the search behind those is the walk's own)
            const uint32_t x0 = tid * 8u;                        // eight positions a thread
            uint32_t bits = (uint32_t)(sh.chg[x0 >> 6] >> (x0 & 63u)) & 0xFFu;
            bits &= x0 >= 8u ? (uint32_t)(sh.eq8[(x0 - 8u) >> 6] >> ((x0 - 8u) & 63u)) : 0u;
            while (bits) {
                const uint32_t x = x0 + (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1u;
                if (x < glim_) tab[x] = find_run(x);
            }
        }
        __syncthreads();
        ENC_TICK(1);
        // ---- wave 0: from repeated run to repeated run.  The walk itself only lists the runs (where each starts and
        // ends: two bit scans per run); what follows from the list (the stretches of literal groups between the runs,
        // byte offsets, run headers, indicator bytes, the table for the packing pass) is done for 64 runs at a time, a
        // lane each
        if (wave == 0) {
            uint64_t o = first_u64(sh.o);
            const uint64_t o_begin = o;
            const uint32_t lead = (uint32_t)((uintptr_t)(out + o_begin) & 15u);       // the chunk's first byte goes to stage[lead]
            uint32_t r = first_u32((uint32_t)(first_u64(sh.pos) - c0));               // chunk offset of the next value to look at
            bool in_run = first_u32(sh.in_run) != 0;
            uint64_t run_start = first_u64(sh.run_start);
            uint32_t run_value = first_u32(sh.run_value);
            uint32_t g = first_u32(sh.groups);
            uint64_t ind_at = first_u64(sh.ind_at);
            uint32_t nseg = 0, cum = 0;
            const uint64_t left = n - c0;                                              // values from the chunk's start on
            const uint32_t cend = last_chunk ? (uint32_t)left : (uint32_t)RLEENC_CHUNK;       // values this chunk deals with
            const uint32_t glim = left >= 8 ? (left - 7 < RLEENC_CHUNK ? (uint32_t)(left - 7) : (uint32_t)RLEENC_CHUNK) : 0u;   // whole groups start below it
            // a byte of the output: into the stage when it belongs to this chunk, else (an indicator byte reserved by
            // an earlier chunk) straight to its place
            auto put_byte = [&](uint64_t at, uint32_t b) {
                if (at >= o_begin) sh.stage[lead + (uint32_t)(at - o_begin)] = (uint8_t)b;
                else if (at < cap) P4S_ST(out + at, (uint8_t)b, 131);
            };
            auto close_literals = [&]() {
                if (g > 0) { if (lane == 0) put_byte(ind_at, (g << 1) | 1u); g = 0; }
            };
            // [ULEB128 count << 1 | value]: a lane per byte (a run's bytes always belong to this chunk)
            auto emit_run = [&](uint32_t count, uint32_t value) {
                const uint32_t v = count << 1;
                const uint32_t vl = v < (1u << 7) ? 1u : v < (1u << 14) ? 2u : v < (1u << 21) ? 3u : v < (1u << 28) ? 4u : 5u;
                uint32_t b = lane < vl ? ((v >> (7 * lane)) & 0x7Fu) | (lane + 1 < vl ? 0x80u : 0u) : (value >> (8 * ((lane - vl) & 3u))) & 0xFFu;
                if (lane < vl + vbytes) sh.stage[lead + (uint32_t)(o - o_begin) + lane] = (uint8_t)b;
                o += vl + vbytes;
            };
            // the bit words, a lane each (a word is one readlane away instead of an LDS round trip)
            const uint64_t my_chg = sh.chg[lane < RLEENC_WORDS + 2 ? lane : 0u];
            // first value at or after `from` that differs from the one before it (past the page's end every bit is set)
            auto next_change = [&](uint32_t from) {
                uint32_t t = ~0u;
                for (uint32_t j = from >> 6; j < RLEENC_WORDS + 1; ++j) {
                    uint64_t m = bcast_u64(my_chg, (int)j);
                    if (j == (from >> 6)) m &= ~0ull << (from & 63u);
                    if (m) { t = j * 64 + (uint32_t)__builtin_ctzll(m); break; }
                }
                return t;
            };
            // G literal groups from chunk offset `from` on (the page's last, padded group)
            auto literal_groups = [&](uint32_t from, uint32_t G) {
                if (lane == 0) { sh.seg_pos[nseg] = (uint16_t)from; sh.seg_cum[nseg] = (uint16_t)cum; sh.seg_off[nseg] = (uint16_t)(lead + (uint32_t)(o - o_begin)); sh.seg_g0[nseg] = (uint16_t)g; }
                ++nseg;
                cum += G;
                while (G > 0) {
                    if (g == 0) { ind_at = o; ++o; }
                    const uint32_t take = G < 63u - g ? G : 63u - g;
                    o += (uint64_t)take * w;
                    g += take;
                    G -= take;
                    if (g == 63) close_literals();
                }
            };
            // ---- the list: ev[e] = start of run e | its end << 16 (0xFFFF: none).  Run e is preceded by the literal
            // groups from where run e - 1 ended (event 0: from r0) up to its start; an entry without a run is the
            // stretch that ends the chunk's whole groups
            constexpr uint32_t EV_NONE = 0xFFFFu, EV_CARRIED = 0xFFFEu;       // (as a start: the run came in from the chunk before)
            uint32_t nev = 0;
            const uint64_t carried_start = run_start;
            const uint32_t carried_value = run_value;
            if (in_run) {
                const uint32_t t = next_change(r);
                if (t > cend || (t == cend && !last_chunk)) {
                    r = cend;                                                        // it goes on in the next chunk
                } else {
                    if (lane == 0) sh.ev[0] = EV_CARRIED | (t << 16);
                    nev = 1;
                    in_run = false;
                    r = t;
                }
            }
            const uint32_t r0 = r;
            if (!in_run) {
                bool first = true;
                while (r < glim) {
                    // the next repeated run: from the table (r is where a run ended: a value that differs), or, for
                    // the chunk's first search, worked out here
                    const uint32_t nr = first_u32(first ? find_run(r) : tab[r]);
                    first = false;
                    const uint32_t p = nr & 0xFFFFu, t = nr >> 16;
                    if (p == EV_NONE) {
                        if (lane == 0) sh.ev[nev] = EV_NONE | (EV_NONE << 16);    // literal groups up to glim
                        ++nev;
                        r += ((glim - r + 7u) >> 3) * 8u;
                        break;
                    }
                    if (t > cend || (t == cend && !last_chunk)) {
                        if (lane == 0) sh.ev[nev] = p | (EV_NONE << 16);              // the run goes on in the next chunk
                        ++nev;
                        in_run = true;
                        run_start = c0 + p;
                        run_value = first_u32(sh.val[p]);
                        r = cend;
                        break;
                    }
                    if (lane == 0) sh.ev[nev] = nr;
                    ++nev;
                    r = t;
                }
            }
            ENC_TICK(2);
            // ---- 64 entries at a time, a lane each
            uint32_t prev_t = r0;                                    // where the entry before the batch ended
            for (uint32_t base = 0; base < nev; base += 64) {
                const uint32_t e = base + lane;
                const bool valid = e < nev;
                const uint32_t ev = valid ? sh.ev[e] : (EV_NONE | (EV_NONE << 16));
                const uint32_t ps = ev & 0xFFFFu, te = ev >> 16;
                const bool carried = valid && ps == EV_CARRIED;
                const bool has_run = valid && ps != EV_NONE;             // (a run starts, or the carried one ends)
                const bool ended = has_run && te != EV_NONE;
                uint32_t from = wave_shr1(te);                           // the end of the run before (every entry but the last has one)
                if (lane == 0) from = prev_t;
                const uint32_t upto = carried ? from : (has_run ? ps : glim);
                const uint32_t G = valid && upto > from ? (upto - from + 7u) >> 3 : 0u;
                const uint32_t g0 = e == 0 ? g : 0u;                     // groups under the indicator byte that is open (a run closes it)
                const uint32_t total = g0 + G;
                const uint32_t nind = (total + 62u) / 63u;               // indicator bytes the stretch touches (the open one included)
                const uint32_t sbytes = G * w + (G ? nind - (g0 ? 1u : 0u) : 0u);
                const uint32_t count = carried ? (uint32_t)(c0 + te - carried_start) : te - ps;
                const uint32_t hv = count << 1;
                const uint32_t vl = hv < (1u << 7) ? 1u : hv < (1u << 14) ? 2u : hv < (1u << 21) ? 3u : hv < (1u << 28) ? 4u : 5u;
                const uint32_t rbytes = ended ? vl + vbytes : 0u;
                uint32_t sc[2] = {valid ? sbytes + rbytes : 0u, G};
                const uint32_t own_bytes = sc[0];
                wave_iscan_n(sc);
                const uint32_t off = (uint32_t)(o - o_begin) + sc[0] - own_bytes;      // chunk-relative offset of the entry's first byte
                const uint64_t segs = __ballot(G > 0);
                if (G > 0) {
                    const uint32_t k = nseg + __builtin_amdgcn_mbcnt_hi((uint32_t)(segs >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)segs, 0u));
                    sh.seg_pos[k] = (uint16_t)from;
                    sh.seg_cum[k] = (uint16_t)(cum + sc[1] - G);
                    sh.seg_off[k] = (uint16_t)(lead + off);
                    sh.seg_g0[k] = (uint16_t)g0;
                }
                // indicator bytes: number i of the stretch sits in front of group 63 i - g0 (number 0 with g0 > 0 is the one
                // that was open); all are full but the last, which the run closes - or nobody yet, at the chunk's end
                uint32_t g_after = 0;
                uint64_t ind_after = 0;
                if (valid && total > 0 && (G > 0 || has_run)) {
                    const uint32_t g_last = total - 63u * (nind - 1u);
                    for (uint32_t i = 0; i < nind; ++i) {
                        const uint64_t at = (i == 0 && g0 > 0) ? ind_at
                                                               : o_begin + off + (63u * i - g0) * w + i - (g0 ? 1u : 0u);
                        if (i + 1 < nind || g_last == 63u || has_run) put_byte(at, i + 1 < nind ? 127u : (g_last << 1) | 1u);
                        else { g_after = g_last; ind_after = at; }
                    }
                } else if (valid && total > 0) {
                    g_after = g0; ind_after = ind_at;                    // (not reached: an entry has groups or a run)
                }
                // the run's header
                if (ended) {
                    const uint32_t value = carried ? carried_value : sh.val[ps];
                    uint8_t *q = sh.stage + lead + off + sbytes;
                    for (uint32_t k = 0; k < vl; ++k) q[k] = (uint8_t)(((hv >> (7 * k)) & 0x7Fu) | (k + 1 < vl ? 0x80u : 0u));
                    for (uint32_t k = 0; k < vbytes; ++k) q[vl + k] = (uint8_t)(value >> (8 * k));
                }
                const uint32_t nb = nev - base < 64u ? nev - base : 64u;          // entries of this batch
                o += bcast_u32(sc[0], 63);
                cum += bcast_u32(sc[1], 63);
                nseg += (uint32_t)__popcll(segs);
                prev_t = bcast_u32(te, (int)nb - 1);
                g = bcast_u32(g_after, (int)nb - 1);                     // (0 behind every entry with a run)
                if (g) ind_at = bcast_u64(ind_after, (int)nb - 1);
            }
            if (last_chunk) {
                // Flush (:463-484) (a repeated run has ended above: the bit behind the page's last value is set)
                if (!in_run) {
                    const uint32_t k = cend - r;                                      // 0 .. 7 values left (r .. r+k-1 < CHUNK + 8)
                    if (k > 0) {
                        const uint32_t word = r >> 6, bit = r & 63u;
                        uint64_t m = bcast_u64(my_chg, (int)word) >> bit;
                        if (bit) m |= bcast_u64(my_chg, (int)(word + 1)) << (64 - bit);
                        const bool equal = ((m >> 1) & ((1ull << (k - 1)) - 1ull)) == 0;
                        if (g == 0 && equal) emit_run(k, first_u32(sh.val[r]));
                        else literal_groups(r, 1);                                    // padded with zeros: the values past the end are 0 here
                    }
                    close_literals();
                }
            }
            if (lane == 0) {
                sh.seg_cum[nseg] = (uint16_t)cum;
                sh.nseg = nseg;
                sh.out_lo = lead;
                sh.out_hi = lead + (uint32_t)(o - o_begin);
                sh.pos = c0 + r; sh.in_run = in_run ? 1u : 0u; sh.run_start = run_start; sh.run_value = run_value;
                sh.groups = g; sh.ind_at = ind_at; sh.o = o;
            }
        }
        __syncthreads();
        ENC_TICK(3);
        // ---- all threads: the literal groups, eight values to `w` bytes each
        {
            const uint32_t nseg = sh.nseg, total = sh.seg_cum[nseg];
            for (uint32_t q = tid; q < total; q += RLEENC_THREADS) {
                uint32_t lo = 0, hi = nseg;                                           // seg_cum[lo] <= q < seg_cum[hi]
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if (sh.seg_cum[mid] <= q) lo = mid; else hi = mid;
                }
                const uint32_t k = q - sh.seg_cum[lo], g0 = sh.seg_g0[lo];
                const uint32_t from = sh.seg_pos[lo] + 8 * k;
                const uint32_t at = sh.seg_off[lo] + k * w + (g0 + k) / 63u + (g0 == 0 ? 1u : 0u);
                rle_pack8_w(w, sh.val + from, sh.stage + at);
            }
        }
        __syncthreads();
        ENC_TICK(4);
        // ---- the chunk's bytes to the page's slot (as far as it holds them)
        // (the next chunk's values have arrived long ago; saying so here, before the stores, keeps the wait for them at
        // the top of the next round from being one for these stores as well: loads and stores share a counter)
#pragma unroll
        for (int k = 0; k <= NPIECE; ++k) asm volatile("" : "+v"(nx[k].x), "+v"(nx[k].y), "+v"(nx[k].z), "+v"(nx[k].w));
        {
            const uint32_t lo = sh.out_lo, hi = sh.out_hi;
            const uint64_t o_end = sh.o;
            const uint64_t o_begin = o_end - (hi - lo);
            // stage offset x is output byte o_begin - lo + x, at a 16-byte aligned address for x a multiple of 16
            const uint64_t fit = cap > o_begin ? cap - o_begin : 0;                   // bytes of this chunk the slot holds
            const uint32_t top = fit < hi - lo ? lo + (uint32_t)fit : hi;
            uint8_t *g0 = out + o_begin - lo;
            const uint32_t a_lo = (lo + 15u) & ~15u, a_hi = top & ~15u;
            if (a_lo < a_hi) {
                for (uint32_t x = a_lo + tid * 16; x < a_hi; x += RLEENC_THREADS * 16) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(sh.stage + x);
                    if (P4S_INB(g0 + x, 16, 132)) *reinterpret_cast<uint4 *>(g0 + x) = v;
                }
                for (uint32_t x = lo + tid; x < a_lo; x += RLEENC_THREADS) P4S_ST(g0 + x, sh.stage[x], 133);
                for (uint32_t x = a_hi + tid; x < top; x += RLEENC_THREADS) P4S_ST(g0 + x, sh.stage[x], 133);
            } else {
                for (uint32_t x = lo + tid; x < top; x += RLEENC_THREADS) P4S_ST(g0 + x, sh.stage[x], 133);
            }
        }
        __syncthreads();
        ENC_TICK(5);
        if (last_chunk) break;
    }
    if (tid == 0) {
        const uint64_t o = sh.o;
        if (sh.bad_value) write_status(&status[page], E_ENCODE_RANGE, 0, w, 0, 0, 0);
        else if (o > cap) write_status(&status[page], E_OUT_CAPACITY, 0, o, cap, 0, o);
#ifdef P4S_EXP_TIMING
        else write_status(&status[page], P4S_E_NONE, n, (tacc[0] & 0xFFFFFFFFull) | (tacc[1] << 32), (tacc[2] & 0xFFFFFFFFull) | (tacc[3] << 32), (tacc[4] & 0xFFFFFFFFull) | (tacc[5] << 32), o);
#else
        else write_status(&status[page], P4S_E_NONE, n, 0, 0, n * 4, o);
#endif
    }
#undef ENC_TICK
}

}  // namespace p4s
