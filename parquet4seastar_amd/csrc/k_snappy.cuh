// Snappy page decompression (ref: snappy_compressor::decompress, src/compression.cc:49-65, which calls
// libsnappy's GetUncompressedLength and RawUncompress): the step in front of the value decoders for pages of
// compressed column chunks (column_chunk_reader.cc:104-110).
//
// Raw format: varint32 decompressed length, then elements.  Tag byte low bits 00: literal (length - 1 in the
// upper six bits, 60..63 = that many minus 59 length bytes follow); 01: copy, length 4..11, 11-bit offset;
// 10: copy, length 1..64, 16-bit offset; 11: copy, length 1..64, 32-bit offset.
//
// One wavefront per page.  The element chain is serial (an element's position is known only when the one
// before it has been parsed), so the tags are parsed by the whole wave in uniform control flow from an LDS
// window of the input, and the 64 lanes move the bytes, one byte per lane and step.  The last 32 KiB of the
// output live in an LDS ring: elements write into the ring only, copies whose source is still in the ring
// (nearly all: compressors look back 64 KiB at most and mostly much less) read it from there at LDS latency,
// and the ring is written to global memory half a ring at a time with coalesced 16-byte stores.  A copy
// from further back reads the page's own output from global memory, which was written at least half a ring
// earlier by this wave.
#pragma once
#include "p4s_common.cuh"

namespace p4s {

constexpr int SNAPPY_THREADS = 64;
constexpr int SNAPPY_WIN = 4096;
#ifndef P4S_SNAPPY_RING
#define P4S_SNAPPY_RING 32768
#endif
constexpr uint32_t SNAPPY_RING = P4S_SNAPPY_RING;  // power of two
constexpr uint32_t SNAPPY_HALF = SNAPPY_RING / 2;
constexpr uint32_t SNAPPY_PIECE = 4096;            // literals enter the ring in pieces of at most this size
constexpr int SNAPPY_SPAN = 192;                   // input bytes the window must hold past the parse position

struct SnappyShared {
    alignas(16) uint8_t win[SNAPPY_WIN + 16];
    alignas(16) uint8_t ring[SNAPPY_RING];
    uint8_t mark[64];                  // per output byte of a step: the lane (+ 1) of the element that starts there
};

// Output bytes [from, to) of the page, all still in the ring, go to global memory; `from` is a multiple of 16.
__device__ __forceinline__ void snappy_flush(const SnappyShared &sh, uint8_t *out, uint64_t from, uint64_t to, uint32_t lane) {
    const uint64_t n16 = (to - from) >> 4;
    for (uint64_t c = lane; c < n16; c += SNAPPY_THREADS) {
        const uint64_t pos = from + (c << 4);
        const uint4 v = *reinterpret_cast<const uint4 *>(sh.ring + (pos & (SNAPPY_RING - 1)));
        if (P4S_INB(out + pos, 16, 103)) __builtin_memcpy(out + pos, &v, 16);
    }
    for (uint64_t pos = from + (n16 << 4) + lane; pos < to; pos += SNAPPY_THREADS)
        P4S_ST(out + pos, sh.ring[pos & (SNAPPY_RING - 1)], 104);
}

__global__ __launch_bounds__(SNAPPY_THREADS) void k_snappy_decompress(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *d_out, uint64_t out_bytes, p4s_page_status *__restrict__ status) {
    __shared__ SnappyShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t lane = threadIdx.x;
    const uint64_t size = first_u64(pv.size);       // uniform values are kept in scalar registers throughout

    // ---- GetUncompressedLength: varint32 of at most 5 bytes (the 5th may carry 4 bits)
    uint32_t expected = 0;
    uint64_t ip = 0;
    bool bad_len = false;
    for (uint32_t shift = 0;; shift += 7) {
        if (shift >= 32 || ip >= size) { bad_len = true; break; }
        const uint32_t c = P4S_LD(pv.base + ip, 100);
        ++ip;
        const uint32_t val = c & 0x7Fu;
        if (((val << shift) >> shift) != val) { bad_len = true; break; }
        expected |= val << shift;
        if (c < 128) break;
    }
    if (bad_len) {
        if (lane == 0) write_status(&status[page], P4S_E_SNAPPY_LENGTH, 0, 0, 0, 0, 0);
        return;
    }
    // the reference's output buffer has the page header's uncompressed size (param0); the caller's has out_bytes
    uint64_t cap = pd.param0;
    const uint64_t room = pd.out_offset < out_bytes ? out_bytes - pd.out_offset : 0;
    if (cap < expected) {                                                   // compression.cc:56-58
        if (lane == 0) write_status(&status[page], P4S_E_SNAPPY_BUFFER, 0, expected, cap, 0, 0);
        return;
    }
    if (room < expected) {
        if (lane == 0) write_status(&status[page], E_OUT_CAPACITY, 0, expected, room, 0, expected);
        return;
    }
    uint8_t *out = d_out + pd.out_offset;
    expected = first_u32(expected);
    ip = first_u64(ip);                 // the same in every lane: positions and lengths live in scalar registers

    uint64_t op = 0;                    // output bytes produced (the ring holds the last SNAPPY_RING of them)
    uint64_t flushed = 0;               // output bytes already in global memory (a multiple of SNAPPY_HALF)
    bool fail = false;
    int64_t wlo = 0, whi = 0;           // the window holds page bytes [wlo, whi)
    while (ip < size) {
        // the window must hold the 64-byte span parsed below, the trailing bytes of its last tag and a short literal
        // after it (and in any case the tag at ip with its four trailing bytes)
        if ((int64_t)ip < wlo || (int64_t)ip + SNAPPY_SPAN > whi) {
            __syncthreads();
            wlo = (int64_t)first_u64((uint64_t)window_load<SNAPPY_WIN, SNAPPY_THREADS>(sh.win, pv, ip));
            whi = wlo + SNAPPY_WIN;
            __syncthreads();
        }
        const uint32_t rel = (uint32_t)((int64_t)ip - wlo);
#ifndef P4S_EXP_SNAPPY_NOBATCH
        {
            // ---- Up to 64 output bytes per step, a lane per byte.  Lane L parses the tag that would start at ip + L; the
            // real elements are found by following the chain through the lanes (readlane: scalar work only, nothing is
            // moved on the way), as many as make up 64 output bytes.  Only short literals and copies from inside the
            // ring take part; at anything else, and at anything the reference rejects, the chain stops and the
            // element-at-a-time code below deals with that element.  Then every lane finds the element its output byte
            // belongs to and where the byte comes from: a literal byte of the input window, a byte the ring already holds,
            // or, for a copy that reaches into this step's own output (a value that repeats part of the value before
            // it; a copy whose offset is shorter than its length), the output byte of another lane, followed through the
            // lanes by pointer jumping until it ends at a byte that is known.
            const uint32_t r = rel + lane;
            const uint32_t b0 = sh.win[r], b1 = sh.win[r + 1], b2 = sh.win[r + 2], b3 = sh.win[r + 3], b4 = sh.win[r + 4];
            const uint32_t kd = b0 & 3u;
            uint32_t n_, hdr_, off_;
            if (kd == 0) { n_ = (b0 >> 2) + 1; hdr_ = 1; off_ = 0; }
            else if (kd == 1) { n_ = ((b0 >> 2) & 7u) + 4; hdr_ = 2; off_ = ((b0 >> 5) << 8) | b1; }
            else if (kd == 2) { n_ = (b0 >> 2) + 1; hdr_ = 3; off_ = b1 | (b2 << 8); }
            else { n_ = (b0 >> 2) + 1; hdr_ = 5; off_ = b1 | (b2 << 8) | (b3 << 16) | (b4 << 24); }
            const uint32_t packed = kd | (hdr_ << 2) | (n_ << 8);
            // inside the span everything is 32-bit: the output is at most 2^32 - 1 bytes long
            const uint32_t avail = size - ip > 0x7FFFFFFFull ? 0x7FFFFFFFu : (uint32_t)(size - ip);
            uint32_t op32 = (uint32_t)op, fl32 = (uint32_t)flushed;
            // the chain itself: nothing but "where does the next tag start" is serial
            const uint32_t in_len_ = kd == 0 ? hdr_ + n_ : hdr_;
            uint32_t cur = 0;
            uint64_t starts = 0;                     // lanes at which an element starts
            while (cur < 64) {
                starts |= 1ull << cur;
                cur += bcast_u32(in_len_, (int)cur);
            }
            // which of them take part: all up to the first that is not simple, does not fit the step, the input or
            // the output, or reaches further back than the ring holds
            bool is_start = (starts >> lane) & 1ull;
            uint32_t my_n = is_start ? n_ : 0u;
            uint32_t my_pos = wave_iscan(my_n) - my_n;
            {
                const uint32_t at = op32 + my_pos;
                const uint32_t reach = at < SNAPPY_RING - 128 ? at : SNAPPY_RING - 128;      // offsets 1 .. reach are moved here
                const bool good = (kd == 0 ? n_ <= 60 : off_ - 1u < reach) && lane + in_len_ <= avail &&
                                  my_pos + n_ <= 64 && expected - at >= n_;
                const uint64_t bad = __ballot(is_start && !good);
                const uint32_t stop = bad ? (uint32_t)__builtin_ctzll(bad) : 64u;           // the first element that stays out
                if (lane >= stop) { is_start = false; my_n = 0; }
                if (bad) cur = stop;                 // (without one, cur is where the tag behind the span's last element starts)
            }
            const uint32_t tot = bcast_u32(wave_iscan(my_n), 63);
            if (tot > 0) {
                // for every output byte the lane of its element
                sh.mark[lane] = 0;
                if (is_start) sh.mark[my_pos] = (uint8_t)(lane + 1);
                __syncthreads();                     // (one wavefront: what this orders is the compiler, which may otherwise answer a lane's read with the lane's own write)
                const uint32_t el = wave_imaxscan((uint32_t)sh.mark[lane]) - 1u;             // (byte 0 belongs to the element at lane 0)
                const bool live = lane < tot;
                const uint32_t src_lane = live ? el : lane;
                const uint32_t e_pk = __shfl(packed, (int)src_lane), e_off = __shfl(off_, (int)src_lane), e_pos = __shfl(my_pos, (int)src_lane);
                const uint32_t k = lane - e_pos;                                             // byte of the element
                const bool lit = (e_pk & 3u) == 0;
                int32_t from = -1;                   // lane whose output byte this one repeats; -1: the byte is known
                uint32_t byte = 0;
                if (live) {
                    if (lit) byte = sh.win[rel + el + 1 + k];
                    else if (lane >= e_off) from = (int32_t)(lane - e_off);
                    else byte = sh.ring[(op32 + lane - e_off) & (SNAPPY_RING - 1)];
                }
                while (__ballot(from >= 0)) {
                    const uint32_t mine = from >= 0 ? 0x80000000u | (uint32_t)from : byte;
                    const uint32_t got = __shfl(mine, from >= 0 ? from : (int)lane);
                    if (from >= 0) {
                        if (got & 0x80000000u) from = (int32_t)(got & 63u);
                        else { byte = got; from = -1; }
                    }
                }
                if (live) sh.ring[(op32 + lane) & (SNAPPY_RING - 1)] = (uint8_t)byte;
                op32 += tot;
                if (op32 - fl32 >= SNAPPY_HALF) {
                    __syncthreads();
                    snappy_flush(sh, out, fl32, fl32 + SNAPPY_HALF, lane);
                    fl32 += SNAPPY_HALF;
                }
                ip += cur;
                op = op32;
                flushed = fl32;
                continue;
            }
        }
#endif
        // (every lane reads the same bytes: say so, the element chain then runs on scalar registers)
        const uint32_t c = first_u32(sh.win[rel]);
        // bytes past the end of the page read as whatever follows (or zero): every use below is bounds-checked
        const uint32_t t1 = first_u32(sh.win[rel + 1]), t2 = first_u32(sh.win[rel + 2]), t3 = first_u32(sh.win[rel + 3]),
                       t4 = first_u32(sh.win[rel + 4]);
        const uint32_t kind = c & 3u;
        if (kind == 0) {
            uint64_t n = (c >> 2) + 1;
            uint64_t hdr = 1;
            if (n > 60) {
                const uint32_t nb = (uint32_t)n - 60;
                if (ip + 1 + nb > size) { fail = true; break; }
                uint32_t v = t1;
                if (nb > 1) v |= t2 << 8;
                if (nb > 2) v |= t3 << 16;
                if (nb > 3) v |= t4 << 24;
                n = (uint64_t)v + 1;
                hdr = 1 + nb;
            }
            if (size - ip - hdr < n) { fail = true; break; }              // the literal runs past the input
            if ((uint64_t)expected - op < n) { fail = true; break; }
            if ((int64_t)(ip + hdr + n) <= whi) {
                // the usual case, a short literal that lies inside the input window: LDS to LDS
                // (n < SNAPPY_WIN here, so one piece)
                const uint32_t srel = rel + (uint32_t)hdr;
#if !defined(P4S_EXP_SNAPPY_PARSEONLY) && !defined(P4S_EXP_SNAPPY_NOLIT)      // timing experiments: the element chain without the byte movement
                for (uint32_t i = lane; i < (uint32_t)n; i += SNAPPY_THREADS)
                    sh.ring[(op + i) & (SNAPPY_RING - 1)] = sh.win[srel + i];
#endif
                op += n;
                ip += hdr + n;
                if (op - flushed >= SNAPPY_HALF) {
                    __syncthreads();
                    snappy_flush(sh, out, flushed, flushed + SNAPPY_HALF, lane);
                    flushed += SNAPPY_HALF;
                }
                continue;
            }
            // input -> ring, in pieces so that no byte that is not in global memory yet is overwritten
            const uint8_t *src = pv.base + ip + hdr;
            for (uint64_t done = 0; done < n;) {
                const uint32_t piece = (uint32_t)(n - done < SNAPPY_PIECE ? n - done : SNAPPY_PIECE);
                for (uint32_t i = lane; i < piece; i += SNAPPY_THREADS)
                    sh.ring[(op + i) & (SNAPPY_RING - 1)] = P4S_LD(src + done + i, 105);
                op += piece;
                done += piece;
                if (op - flushed >= SNAPPY_HALF) {
                    __syncthreads();
                    snappy_flush(sh, out, flushed, flushed + SNAPPY_HALF, lane);
                    flushed += SNAPPY_HALF;
                }
            }
            ip += hdr + n;
        } else {
            uint32_t n, hdr;
            uint64_t offset;
            if (kind == 1) {
                n = ((c >> 2) & 7u) + 4; offset = ((uint64_t)(c >> 5) << 8) | t1; hdr = 2;
            } else if (kind == 2) {
                n = (c >> 2) + 1; offset = t1 | (t2 << 8); hdr = 3;
            } else {
                n = (c >> 2) + 1; offset = (uint64_t)t1 | ((uint64_t)t2 << 8) | ((uint64_t)t3 << 16) | ((uint64_t)t4 << 24); hdr = 5;
            }
            if (ip + hdr > size) { fail = true; break; }
            if (offset == 0 || offset > op) { fail = true; break; }
            if ((uint64_t)expected - op < n) { fail = true; break; }
            const uint64_t src0 = op - offset;
            uint32_t k = lane;
            if (offset < n) k = lane % (uint32_t)offset;                         // offset < length repeats the pattern
            if (offset <= SNAPPY_RING - 64) {
                // the source is in the ring and stays there while this element is written
#if !defined(P4S_EXP_SNAPPY_PARSEONLY) && !defined(P4S_EXP_SNAPPY_NOCOPY)
                if (lane < n) sh.ring[(op + lane) & (SNAPPY_RING - 1)] = sh.ring[(src0 + k) & (SNAPPY_RING - 1)];
#endif
            } else {
                // further back than the ring: those bytes went to global memory at least half a ring ago
                // (offset > RING - 64 and op - flushed < HALF + 64, so src0 + n <= flushed)
                __threadfence_block();
                if (lane < n) sh.ring[(op + lane) & (SNAPPY_RING - 1)] = P4S_LD(out + src0 + k, 101);
            }
            ip += hdr;
            op += n;
            if (op - flushed >= SNAPPY_HALF) {
                __syncthreads();
                snappy_flush(sh, out, flushed, flushed + SNAPPY_HALF, lane);
                flushed += SNAPPY_HALF;
            }
        }
    }
    if (!fail && op != expected) fail = true;
    __syncthreads();
    if (!fail) snappy_flush(sh, out, flushed, op, lane);
    if (lane == 0) {
        if (fail) write_status(&status[page], P4S_E_SNAPPY_FAILED, 0, 0, 0, 0, 0);     // compression.cc:61-63
        else write_status(&status[page], P4S_E_NONE, expected, 0, 0, size, expected);
    }
}

}  // namespace p4s
