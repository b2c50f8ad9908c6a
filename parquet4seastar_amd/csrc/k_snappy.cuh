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
constexpr uint32_t SNAPPY_RING = 32768;            // power of two
constexpr uint32_t SNAPPY_HALF = SNAPPY_RING / 2;
constexpr uint32_t SNAPPY_PIECE = 4096;            // literals enter the ring in pieces of at most this size
constexpr int SNAPPY_SPAN = 192;                   // input bytes the window must hold past the parse position

struct SnappyShared {
    alignas(16) uint8_t win[SNAPPY_WIN + 16];
    alignas(16) uint8_t ring[SNAPPY_RING];
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
            // ---- Many elements per step.  Lane L parses the tag that would start at ip + L; the real elements are
            // found by following the chain through the lanes (readlane, no LDS round trip per element).  Only short
            // literals and copies from inside the ring are moved here; at anything else, and at anything the
            // reference rejects, the chain stops and the element-at-a-time code below deals with that element.
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
            uint32_t cur = 0;
            while (cur < 64) {
                const uint32_t e0 = bcast_u32(packed, (int)cur), eoff = bcast_u32(off_, (int)cur);
                const uint32_t ek = e0 & 3u, eh = (e0 >> 2) & 7u, en = e0 >> 8;
                const uint32_t in_len = ek == 0 ? eh + en : eh;
                const uint32_t reach = op32 < SNAPPY_RING - 64 ? op32 : SNAPPY_RING - 64;    // offsets 1 .. reach are moved here
                if (ek == 0 ? en > 60 : eoff - 1u >= reach) break;
                if (cur + in_len > avail) break;
                if (expected - op32 < en) break;
                // The element after it goes in the same step when it is as simple and reads nothing the first one
                // writes (a literal, or a copy from at least both lengths back): the two LDS round trips overlap.
                const uint32_t cur1 = cur + in_len, op1 = op32 + en;
                bool pair = false;
                uint32_t eoff1 = 0, ek1 = 0, en1 = 0, in_len1 = 0;
#ifndef P4S_EXP_SNAPPY_NOPAIR
                if (cur1 < 64) {
                    const uint32_t e1 = bcast_u32(packed, (int)cur1);
                    eoff1 = bcast_u32(off_, (int)cur1);
                    ek1 = e1 & 3u;
                    en1 = e1 >> 8;
                    in_len1 = ek1 == 0 ? ((e1 >> 2) & 7u) + en1 : ((e1 >> 2) & 7u);
                    const uint32_t reach1 = op1 < SNAPPY_RING - 64 ? op1 : SNAPPY_RING - 64;
                    pair = (ek1 == 0 ? en1 <= 60 : (eoff1 - 1u < reach1 && eoff1 >= en + en1)) &&
                           cur1 + in_len1 <= avail && expected - op1 >= en1;
                }
#endif
                {
                    uint32_t k = lane;
                    if (ek != 0 && eoff < en) k = lane % eoff;                               // offset < length repeats the pattern
                    const uint8_t *s0 = ek == 0 ? sh.win + (rel + cur + 1 + lane) : sh.ring + ((op32 - eoff + k) & (SNAPPY_RING - 1));
                    const uint8_t *s1 = ek1 == 0 ? sh.win + (rel + cur1 + 1 + lane) : sh.ring + ((op1 - eoff1 + lane) & (SNAPPY_RING - 1));
                    uint8_t v0 = 0, v1 = 0;
                    if (lane < en) v0 = *s0;
                    if (pair && lane < en1) v1 = *s1;
                    if (lane < en) sh.ring[(op32 + lane) & (SNAPPY_RING - 1)] = v0;
                    if (pair && lane < en1) sh.ring[(op1 + lane) & (SNAPPY_RING - 1)] = v1;
                }
                cur = cur1;
                op32 = op1;
                if (pair) { cur += in_len1; op32 += en1; }
                if (op32 - fl32 >= SNAPPY_HALF) {
                    __syncthreads();
                    snappy_flush(sh, out, fl32, fl32 + SNAPPY_HALF, lane);
                    fl32 += SNAPPY_HALF;
                }
            }
            ip += cur;
            op = op32;
            flushed = fl32;
            if (cur > 0) continue;
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
