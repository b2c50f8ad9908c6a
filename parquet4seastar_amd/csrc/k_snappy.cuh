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
// window of the input, and the 64 lanes move the bytes: a literal is a coalesced copy from the input, a copy
// (at most 64 bytes) is one byte per lane from the page's own earlier output in global memory, offset < length
// repeating the pattern.  Output written since the last fence may not be visible to the wave's loads yet:
// `safe` is the output position up to which it is; a copy whose source reaches past it fences first.
#pragma once
#include "k_strings.cuh"

namespace p4s {

constexpr int SNAPPY_THREADS = 64;
constexpr int SNAPPY_WIN = 4096;

struct SnappyShared {
    alignas(16) uint8_t win[SNAPPY_WIN + 16];
};

__global__ __launch_bounds__(SNAPPY_THREADS) void k_snappy_decompress(
        const uint8_t *__restrict__ d_in, uint64_t in_bytes, const p4s_page_desc *__restrict__ pages,
        uint8_t *d_out, uint64_t out_bytes, p4s_page_status *__restrict__ status) {
    __shared__ SnappyShared sh;
    const uint32_t page = blockIdx.x;
    const p4s_page_desc pd = pages[page];
    const PageView pv = page_view(d_in, in_bytes, pd);
    const uint32_t lane = threadIdx.x;
    const uint64_t size = pv.size;

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

    uint64_t op = 0, safe = 0;
    bool fail = false;
    int64_t wlo = 0, whi = 0;           // the window holds page bytes [wlo, whi)
    while (ip < size) {
        // the tag and its (up to four) trailing bytes must lie inside the window
        if ((int64_t)ip < wlo || (int64_t)ip + 5 > whi) {
            __syncthreads();
            wlo = window_load<SNAPPY_WIN, SNAPPY_THREADS>(sh.win, pv, ip);
            whi = wlo + SNAPPY_WIN;
            __syncthreads();
        }
        const uint32_t rel = (uint32_t)((int64_t)ip - wlo);
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
            wg_copy(out + op, pv.base + ip + hdr, n);
            ip += hdr + n;
            op += n;
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
            const uint64_t src_end = src0 + (offset < n ? offset : (uint64_t)n);
            if (src_end > safe) {
                __threadfence();        // this wave's earlier stores are visible to its loads from here on
                safe = op;
            }
            if (lane < n) {
                const uint32_t k = offset >= n ? lane : lane % (uint32_t)offset;
                P4S_ST(out + op + lane, P4S_LD(out + src0 + k, 101), 102);
            }
            ip += hdr;
            op += n;
        }
    }
    if (!fail && op != expected) fail = true;
    if (lane == 0) {
        if (fail) write_status(&status[page], P4S_E_SNAPPY_FAILED, 0, 0, 0, 0, 0);     // compression.cc:61-63
        else write_status(&status[page], P4S_E_NONE, expected, 0, 0, size, expected);
    }
}

}  // namespace p4s
