/*
 * Synthetic Parquet page encoders (host C).  These build the INPUTS of the
 * decode path for tests and for the benchmark: the reference library can only
 * encode PLAIN, RLE_DICTIONARY and DELTA_BINARY_PACKED (src/encoding.cc:1013-1046
 * of the reference), so every other encoding needs a generator of its own, and
 * the benchmark needs hundreds of megabytes of distinct pages quickly.
 *
 * Wire formats follow the Parquet format specification (doc/parquet/Encodings.md
 * in the reference tree).  Padding bits are always written as zero.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint8_t *out;
    int64_t cap;
    int64_t pos;      /* bytes written */
    uint64_t acc;     /* pending bits */
    int nacc;         /* number of pending bits (< 8 after every put) */
    int overflow;
} bitsink;

static void sink_init(bitsink *s, uint8_t *out, int64_t cap) {
    s->out = out; s->cap = cap; s->pos = 0; s->acc = 0; s->nacc = 0; s->overflow = 0;
}

static void sink_byte(bitsink *s, uint8_t b) {
    if (s->pos < s->cap) s->out[s->pos] = b; else s->overflow = 1;
    s->pos++;
}

/* append the low w bits of v (w <= 64), LSB first */
static void sink_bits(bitsink *s, uint64_t v, int w) {
    while (w > 0) {
        int take = 56 - s->nacc;            /* keep acc below 64 bits */
        if (take > w) take = w;
        uint64_t part = (take == 64) ? v : (v & (((uint64_t)1 << take) - 1));
        s->acc |= part << s->nacc;
        s->nacc += take;
        v = (take >= 64) ? 0 : (v >> take);
        w -= take;
        while (s->nacc >= 8) {
            sink_byte(s, (uint8_t)(s->acc & 0xFF));
            s->acc >>= 8;
            s->nacc -= 8;
        }
    }
}

static void sink_align(bitsink *s) {
    if (s->nacc > 0) {
        sink_byte(s, (uint8_t)(s->acc & 0xFF));
        s->acc = 0; s->nacc = 0;
    }
}

static void sink_uleb(bitsink *s, uint64_t v) {
    sink_align(s);
    while (v >= 0x80) { sink_byte(s, (uint8_t)(v | 0x80)); v >>= 7; }
    sink_byte(s, (uint8_t)v);
}

static void sink_zigzag(bitsink *s, int64_t v) {
    sink_uleb(s, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63));
}

static int64_t sink_finish(bitsink *s) {
    sink_align(s);
    return s->overflow ? -1 : s->pos;
}

/* ---------------- plain bit packing ---------------- */
int64_t gen_bitpack(const uint32_t *v, int64_t n, int w, uint8_t *out, int64_t cap) {
    bitsink s; sink_init(&s, out, cap);
    for (int64_t i = 0; i < n; ++i) sink_bits(&s, v[i], w);
    return sink_finish(&s);
}

/* ---------------- RLE / bit-packed hybrid ---------------- */

static void rle_flush_literal(bitsink *s, const uint32_t *v, int64_t start, int64_t groups,
                              int64_t n, int w) {
    if (groups == 0) return;
    sink_uleb(s, ((uint64_t)groups << 1) | 1);
    for (int64_t i = start; i < start + groups * 8; ++i) sink_bits(s, i < n ? v[i] : 0, w);
    sink_align(s);
}

/* Greedy encoder: a stretch of >= min_repeat equal values starting on a group
 * boundary becomes a repeated run, everything else goes into literal groups of
 * 8 (the last group zero-padded).  max_groups bounds a literal run (63 keeps
 * one-byte headers). */
int64_t gen_rle_encode(const uint32_t *v, int64_t n, int w, int max_groups, int min_repeat,
                       uint8_t *out, int64_t cap) {
    bitsink s; sink_init(&s, out, cap);
    int vbytes = (w + 7) / 8;
    int64_t i = 0, lit_start = 0, lit_groups = 0;
    if (min_repeat < 1) min_repeat = 8;
    while (i < n) {
        int64_t r = 1;
        while (i + r < n && v[i + r] == v[i]) ++r;
        if (r >= min_repeat) {
            rle_flush_literal(&s, v, lit_start, lit_groups, n, w);
            lit_groups = 0;
            sink_uleb(&s, (uint64_t)r << 1);
            for (int b = 0; b < vbytes; ++b) sink_byte(&s, (uint8_t)(v[i] >> (8 * b)));
            i += r;
            lit_start = i;
        } else {
            if (lit_groups == 0) lit_start = i;
            ++lit_groups;
            i += 8;
            if (lit_groups == max_groups) {
                rle_flush_literal(&s, v, lit_start, lit_groups, n, w);
                lit_groups = 0;
                lit_start = i;
            }
        }
    }
    rle_flush_literal(&s, v, lit_start, lit_groups, n, w);
    return sink_finish(&s);
}

/* Exact run structure: run r is repeated (kinds[r]==0: counts[r] copies of
 * rep_values[r], written with value_bytes[r] bytes) or literal (kinds[r]==1:
 * counts[r] GROUPS taken from lit_values in order).  Lets tests build
 * zero-length runs, multi-byte headers and unmasked repeat values. */
int64_t gen_rle_from_runs(const int32_t *kinds, const int64_t *counts, const uint32_t *rep_values,
                          const uint32_t *lit_values, int64_t n_runs, int w,
                          uint8_t *out, int64_t cap) {
    bitsink s; sink_init(&s, out, cap);
    int vbytes = (w + 7) / 8;
    int64_t lit = 0;
    for (int64_t r = 0; r < n_runs; ++r) {
        if (kinds[r]) {
            sink_uleb(&s, ((uint64_t)counts[r] << 1) | 1);
            for (int64_t i = 0; i < counts[r] * 8; ++i) sink_bits(&s, lit_values[lit++], w);
            sink_align(&s);
        } else {
            sink_uleb(&s, (uint64_t)counts[r] << 1);
            for (int b = 0; b < vbytes; ++b) sink_byte(&s, (uint8_t)(rep_values[r] >> (8 * b)));
        }
    }
    return sink_finish(&s);
}

/* ---------------- DELTA_BINARY_PACKED ---------------- */

static int bits_needed(uint64_t v) {
    int b = 0;
    while (v) { ++b; v >>= 1; }
    return b;
}

/* elem_size 4 (int32 values, deltas computed in 32-bit wrap-around and
 * sign-extended, like a writer for INT32 columns) or 8.  block_size must be a
 * multiple of n_miniblocks. */
int64_t gen_delta_encode(const void *values, int elem_size, int64_t n,
                         uint32_t block_size, uint32_t n_miniblocks,
                         uint8_t *out, int64_t cap) {
    bitsink s; sink_init(&s, out, cap);
    const int32_t *v32 = (const int32_t *)values;
    const int64_t *v64 = (const int64_t *)values;
    uint32_t vpm = block_size / n_miniblocks;
    sink_uleb(&s, block_size);
    sink_uleb(&s, n_miniblocks);
    sink_uleb(&s, (uint64_t)n);
    int64_t first = n > 0 ? (elem_size == 4 ? (int64_t)v32[0] : v64[0]) : 0;
    sink_zigzag(&s, first);
    int64_t *deltas = (int64_t *)malloc(sizeof(int64_t) * (size_t)block_size);
    int64_t i = 1;
    while (i < n) {
        int64_t nb = n - i < (int64_t)block_size ? n - i : (int64_t)block_size;
        int64_t min_delta = INT64_MAX;
        for (int64_t k = 0; k < nb; ++k) {
            int64_t d;
            if (elem_size == 4) d = (int64_t)(int32_t)((uint32_t)v32[i + k] - (uint32_t)v32[i + k - 1]);
            else d = (int64_t)((uint64_t)v64[i + k] - (uint64_t)v64[i + k - 1]);
            deltas[k] = d;
            if (d < min_delta) min_delta = d;
        }
        sink_zigzag(&s, min_delta);
        uint8_t widths[4096];
        for (uint32_t m = 0; m < n_miniblocks; ++m) {
            int64_t lo = (int64_t)m * vpm, hi = lo + vpm;
            if (hi > nb) hi = nb;
            uint64_t mx = 0;
            for (int64_t k = lo; k < hi; ++k) {
                uint64_t u = (uint64_t)deltas[k] - (uint64_t)min_delta;
                if (u > mx) mx = u;
            }
            widths[m] = (lo < nb) ? (uint8_t)bits_needed(mx) : 0;
        }
        sink_align(&s);
        for (uint32_t m = 0; m < n_miniblocks; ++m) sink_byte(&s, widths[m]);
        for (uint32_t m = 0; m < n_miniblocks; ++m) {
            int64_t lo = (int64_t)m * vpm;
            if (lo >= nb) break;                      /* unused miniblocks carry no body */
            for (int64_t k = lo; k < lo + vpm; ++k) {
                uint64_t u = k < nb ? (uint64_t)deltas[k] - (uint64_t)min_delta : 0;
                sink_bits(&s, u, widths[m]);
            }
        }
        i += nb;
    }
    free(deltas);
    return sink_finish(&s);
}

/* upper bound on gen_delta_encode output */
int64_t gen_delta_max_size(int elem_size, int64_t n, uint32_t block_size, uint32_t n_miniblocks) {
    int64_t blocks = (n + block_size - 1) / block_size + 1;
    return 64 + blocks * (10 + (int64_t)n_miniblocks + (int64_t)block_size * elem_size);
}

/* ---------------- BYTE_STREAM_SPLIT ---------------- */
void gen_bss_encode(const uint8_t *values, int elem_size, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < elem_size; ++k)
            out[(int64_t)k * n + i] = values[i * elem_size + k];
}
