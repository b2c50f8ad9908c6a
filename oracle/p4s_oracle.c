/*
 * CPU oracle: scalar restatement of the reference page decoders.
 * TEST INFRASTRUCTURE ONLY (see p4s_oracle.h).  Plain C99, no dependencies.
 *
 * The reference keeps a (byte_offset, bit_offset, buffered 64-bit word) reader;
 * this restatement keeps a single bit position `pos` = byte_offset*8+bit_offset,
 * which has the same observable behaviour: values, success/failure of every
 * call, and bytes_left().
 */
#include "p4s_oracle.h"

#include <stdlib.h>
#include <string.h>

static void set_err(orc_error *e, int code, uint64_t a0, uint64_t a1) {
    if (e) { e->code = code; e->aux0 = a0; e->aux1 = a1; }
}

/* ------------------------------------------------------------------ */
/* bit reader: bit_stream_utils.hh:114-189, 244-460                    */
/* ------------------------------------------------------------------ */

void orc_br_init(orc_bitreader *r, const uint8_t *buf, int64_t len) {
    r->buf = buf; r->len = len; r->pos = 0;
}

/* bits [pos, pos+w) of the little-endian bit string, 0 <= w <= 64.  The
 * caller guarantees the bits lie inside the buffer. */
static uint64_t bits_at(const uint8_t *buf, int64_t len, uint64_t pos, int w) {
    if (w == 0) return 0;
    uint64_t byte = pos >> 3;
    int sh = (int)(pos & 7);
    uint64_t lo = 0;
    for (int i = 0; i < 8; ++i) {
        if ((int64_t)(byte + (uint64_t)i) < len) lo |= (uint64_t)buf[byte + (uint64_t)i] << (8 * i);
    }
    uint64_t v = lo >> sh;
    if (sh + w > 64 && (int64_t)(byte + 8) < len) {
        v |= (uint64_t)buf[byte + 8] << (64 - sh);
    }
    if (w < 64) v &= ((uint64_t)1 << w) - 1;
    return v;
}

static void store_elem(void *out, int elem_size, int64_t i, uint64_t v) {
    switch (elem_size) {
    case 1: ((uint8_t *)out)[i] = (uint8_t)v; break;
    case 2: ((uint16_t *)out)[i] = (uint16_t)v; break;
    case 4: ((uint32_t *)out)[i] = (uint32_t)v; break;
    default: ((uint64_t *)out)[i] = v; break;
    }
}

/* BitReader::GetBatch, bit_stream_utils.hh:290-366: the batch is clamped to
 * remaining_bits / num_bits (:301-305); values are plain LSB-first fields. */
int64_t orc_br_get_batch(orc_bitreader *r, int num_bits, void *out, int elem_size, int64_t n) {
    uint64_t remaining = (uint64_t)r->len * 8 - r->pos;
    uint64_t needed = (uint64_t)num_bits * (uint64_t)n;
    if (remaining < needed) {
        n = (int64_t)(remaining / (uint64_t)num_bits);
    }
    for (int64_t i = 0; i < n; ++i) {
        uint64_t v = bits_at(r->buf, r->len, r->pos, num_bits);
        r->pos += (uint64_t)num_bits;
        store_elem(out, elem_size, i, v);
    }
    return n;
}

/* BitReader::GetAligned, bit_stream_utils.hh:368-393: fails when the target
 * type is too small or when fewer than num_bytes remain after rounding the
 * cursor up to a byte; only the low num_bytes of *v are overwritten. */
int orc_br_get_aligned(orc_bitreader *r, int num_bytes, int elem_size, uint64_t *v) {
    if (num_bytes > elem_size) return 0;
    uint64_t b = (r->pos + 7) >> 3;
    if ((int64_t)(b + (uint64_t)num_bytes) > r->len) return 0;
    uint64_t nv = 0;
    for (int i = 0; i < num_bytes; ++i) nv |= (uint64_t)r->buf[b + (uint64_t)i] << (8 * i);
    if (num_bytes >= 8) {
        *v = nv;
    } else {
        uint64_t mask = ((uint64_t)1 << (8 * num_bytes)) - 1;
        *v = (*v & ~mask) | nv;
    }
    r->pos = (b + (uint64_t)num_bytes) * 8;
    return 1;
}

/* bit_stream_utils.hh:405-422: at most 5 bytes, 32-bit accumulator. */
int orc_br_get_vlq32(orc_bitreader *r, uint32_t *v) {
    uint32_t tmp = 0;
    for (int i = 0; i < 5; ++i) {
        uint64_t byte = 0;
        if (!orc_br_get_aligned(r, 1, 1, &byte)) return 0;
        tmp |= (uint32_t)(byte & 0x7F) << (7 * i);
        if ((byte & 0x80) == 0) { *v = tmp; return 1; }
    }
    return 0;
}

/* bit_stream_utils.hh:424-441: at most 10 bytes, 64-bit accumulator. */
int orc_br_get_vlq64(orc_bitreader *r, uint64_t *v) {
    uint64_t tmp = 0;
    for (int i = 0; i < 10; ++i) {
        uint64_t byte = 0;
        if (!orc_br_get_aligned(r, 1, 1, &byte)) return 0;
        tmp |= (uint64_t)(byte & 0x7F) << (7 * i);
        if ((byte & 0x80) == 0) { *v = tmp; return 1; }
    }
    return 0;
}

/* bit_stream_utils.hh:448-460 */
int orc_br_get_zigzag32(orc_bitreader *r, int32_t *v) {
    uint32_t u;
    if (!orc_br_get_vlq32(r, &u)) return 0;
    *v = (int32_t)((u >> 1) ^ (uint32_t)(-(int32_t)(u & 1)));
    return 1;
}

int orc_br_get_zigzag64(orc_bitreader *r, int64_t *v) {
    uint64_t u;
    if (!orc_br_get_vlq64(r, &u)) return 0;
    *v = (int64_t)((u >> 1) ^ (uint64_t)(-(int64_t)(u & 1)));
    return 1;
}

/* bit_stream_utils.hh:174-177 */
int64_t orc_br_bytes_left(const orc_bitreader *r) {
    return r->len - (int64_t)((r->pos + 7) >> 3);
}

/* ------------------------------------------------------------------ */
/* RLE / bit-packed hybrid: rle_encoding.hh:282-348                    */
/* ------------------------------------------------------------------ */

void orc_rle_init(orc_rle *d, const uint8_t *buf, int64_t len, int bit_width) {
    orc_br_init(&d->br, buf, len);
    d->bit_width = bit_width;
    d->current_value = 0;
    d->repeat_count = 0;
    d->literal_count = 0;
}

/* RleDecoder::NextCounts, rle_encoding.hh:321-348 */
static int rle_next_counts(orc_rle *d, int elem_size) {
    uint32_t indicator = 0;
    if (!orc_br_get_vlq32(&d->br, &indicator)) return 0;
    uint32_t count = indicator >> 1;
    if (indicator & 1) {
        if (count > (uint32_t)INT32_MAX / 8) return 0;
        d->literal_count = (int64_t)count * 8;
    } else {
        if (count > (uint32_t)INT32_MAX) return 0;
        d->repeat_count = count;
        int nbytes = (d->bit_width + 7) / 8;
        if (!orc_br_get_aligned(&d->br, nbytes, elem_size, &d->current_value)) return 0;
    }
    return 1;
}

/* RleDecoder::GetBatch, rle_encoding.hh:282-315.  A literal run that cannot
 * supply min(remaining, literal_count) values ends the call with the count
 * reached so far (:301-304); the repeated value is the raw ceil(w/8)-byte
 * little-endian word, not masked to w bits (:342-343). */
int64_t orc_rle_get_batch(orc_rle *d, void *out, int elem_size, int64_t n) {
    int64_t done = 0;
    uint8_t *o = (uint8_t *)out;
    while (done < n) {
        int64_t remaining = n - done;
        if (d->repeat_count > 0) {
            int64_t k = remaining < d->repeat_count ? remaining : d->repeat_count;
            for (int64_t i = 0; i < k; ++i) store_elem(o + done * elem_size, elem_size, i, d->current_value);
            d->repeat_count -= k;
            done += k;
        } else if (d->literal_count > 0) {
            int64_t k = remaining < d->literal_count ? remaining : d->literal_count;
            int64_t got = orc_br_get_batch(&d->br, d->bit_width, o + done * elem_size, elem_size, k);
            if (got != k) return done;
            d->literal_count -= k;
            done += k;
        } else {
            if (!rle_next_counts(d, elem_size)) return done;
        }
    }
    return done;
}

int64_t orc_rle_decode(const uint8_t *buf, int64_t len, int bit_width,
                       void *out, int elem_size, int64_t n) {
    orc_rle d;
    orc_rle_init(&d, buf, len, bit_width);
    return orc_rle_get_batch(&d, out, elem_size, n);
}

void orc_rle_decode_batches(const uint8_t *buf, int64_t len, int bit_width,
                            void *out, int elem_size,
                            const int64_t *batch_sizes, int n_batches, int64_t *counts) {
    orc_rle d;
    orc_rle_init(&d, buf, len, bit_width);
    uint8_t *o = (uint8_t *)out;
    for (int b = 0; b < n_batches; ++b) {
        counts[b] = orc_rle_get_batch(&d, o, elem_size, batch_sizes[b]);
        o += counts[b] * elem_size;
    }
}

int64_t orc_bitpacked_decode(const uint8_t *buf, int64_t len, int bit_width,
                             void *out, int elem_size, int64_t n) {
    orc_bitreader r;
    orc_br_init(&r, buf, len);
    return orc_br_get_batch(&r, bit_width, out, elem_size, n);
}

/* ------------------------------------------------------------------ */
/* level_decoder: encoding.hh:42-82, encoding.cc:26-79                 */
/* ------------------------------------------------------------------ */

static uint32_t level_bit_width(uint32_t max_level) {   /* encoding.hh:47-49 */
    uint32_t w = 0;
    while (max_level) { ++w; max_level >>= 1; }
    return w;
}

static void fill_zero(void *out, int elem_size, int64_t n) {
    memset(out, 0, (size_t)(n * elem_size));
}

int64_t orc_levels_v1(const uint8_t *buf, int64_t len, int encoding, uint32_t max_level,
                      uint32_t num_values, int64_t n, void *out, int elem_size,
                      int64_t *consumed, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    uint32_t w = level_bit_width(max_level);
    if (n > (int64_t)num_values) n = num_values;            /* encoding.hh:63 */
    if (w == 0) {                                           /* encoding.cc:32-34, encoding.hh:64-68 */
        *consumed = 0;
        fill_zero(out, elem_size, n);
        return n;
    }
    if (encoding == 3) {                                    /* RLE, encoding.cc:35-50 */
        if (len < 4) { set_err(err, ORC_E_LEVELS_EOF, 4, (uint64_t)len); return -1; }
        int32_t rlen;
        memcpy(&rlen, buf, 4);
        if (rlen < 0) { set_err(err, ORC_E_LEVELS_NEGATIVE, (uint64_t)(int64_t)rlen, 0); return -1; }
        if ((int64_t)rlen > len) { set_err(err, ORC_E_LEVELS_EOF, (uint64_t)rlen, (uint64_t)len); return -1; }
        /* The reference compares rlen with size() rather than size()-4 and may
         * run up to 4 bytes past the page; the restatement clamps instead. */
        int64_t avail = len - 4;
        int64_t use = rlen < avail ? rlen : avail;
        *consumed = 4 + use;
        return orc_rle_decode(buf + 4, use, (int)w, out, elem_size, n);
    } else if (encoding == 4) {                             /* BIT_PACKED, encoding.cc:51-62 */
        uint64_t bit_len = (uint64_t)num_values * w;
        uint64_t byte_len = (bit_len + 7) >> 3;
        if (byte_len > (uint64_t)INT32_MAX) { set_err(err, ORC_E_LEVELS_BITPACKED_LEN, byte_len, 0); return -1; }
        if (byte_len > (uint64_t)len) { set_err(err, ORC_E_LEVELS_EOF, byte_len, (uint64_t)len); return -1; }
        *consumed = (int64_t)byte_len;
        return orc_bitpacked_decode(buf, (int64_t)byte_len, (int)w, out, elem_size, n);
    }
    set_err(err, ORC_E_LEVELS_ENCODING, (uint64_t)encoding, 0);  /* encoding.cc:64 */
    return -1;
}

int64_t orc_levels_v2(const uint8_t *buf, int64_t len, uint32_t max_level,
                      uint32_t num_values, int64_t n, void *out, int elem_size,
                      orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    if (len > (int64_t)INT32_MAX) { set_err(err, ORC_E_LEVELS_V2_LEN, (uint64_t)len, 0); return -1; }
    uint32_t w = level_bit_width(max_level);
    if (n > (int64_t)num_values) n = num_values;
    if (w == 0) { fill_zero(out, elem_size, n); return n; }
    return orc_rle_decode(buf, len, (int)w, out, elem_size, n);
}

/* ------------------------------------------------------------------ */
/* DELTA_BINARY_PACKED: encoding.cc:141-250                            */
/* ------------------------------------------------------------------ */

void orc_delta_free(orc_delta *d) { free(d->widths); d->widths = NULL; d->widths_cap = 0; }

/* encoding.cc:178-205 */
int orc_delta_reset(orc_delta *d, const uint8_t *buf, int64_t len, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    memset(d, 0, sizeof(*d));
    orc_br_init(&d->br, buf, len);
    if (!orc_br_get_vlq64(&d->br, &d->values_per_block)) { set_err(err, ORC_E_DELTA_HEADER_EOF, 0, 0); return -1; }
    if (!orc_br_get_vlq64(&d->br, &d->num_mini_blocks)) { set_err(err, ORC_E_DELTA_HEADER_EOF, 0, 0); return -1; }
    if (d->num_mini_blocks == 0) { set_err(err, ORC_E_DELTA_ZERO_MINIBLOCKS, 0, 0); return -1; }
    if (!orc_br_get_vlq64(&d->br, &d->values_remaining)) { set_err(err, ORC_E_DELTA_HEADER_EOF, 0, 0); return -1; }
    int64_t first;
    if (!orc_br_get_zigzag64(&d->br, &first)) { set_err(err, ORC_E_DELTA_HEADER_EOF, 0, 0); return -1; }
    d->last_value = (uint64_t)first;
    /* The reference allocates num_mini_blocks width bytes up front; a page can
     * never supply more width bytes than it has bytes, so cap the allocation. */
    uint64_t cap = d->num_mini_blocks;
    if (cap > (uint64_t)len + 1) cap = (uint64_t)len + 1;
    d->widths = (uint8_t *)malloc((size_t)cap);
    d->widths_cap = cap;
    d->values_per_mini_block = d->values_per_block / d->num_mini_blocks;
    d->values_current_mini_block = 0;
    d->mini_block_idx = d->num_mini_blocks;
    if (d->values_per_mini_block == 0 && d->values_remaining > 1) {
        /* block_size < n_miniblocks: the reference underflows its per-miniblock
         * counter; the build rejects the page instead (recorded deviation). */
        set_err(err, ORC_E_DELTA_BAD_SHAPE, d->values_per_block, d->num_mini_blocks);
        return -1;
    }
    return 0;
}

/* encoding.cc:157-170 */
static int delta_init_block(orc_delta *d, orc_error *err) {
    int64_t min_delta;
    if (!orc_br_get_zigzag64(&d->br, &min_delta)) { set_err(err, ORC_E_DELTA_BLOCK_HEADER_EOF, 0, 0); return -1; }
    d->min_delta = (uint64_t)min_delta;
    for (uint64_t i = 0; i < d->num_mini_blocks; ++i) {
        uint64_t b = 0;
        if (!orc_br_get_aligned(&d->br, 1, 1, &b)) { set_err(err, ORC_E_DELTA_BLOCK_HEADER_EOF, 0, 0); return -1; }
        d->widths[i] = (uint8_t)b;   /* i < widths_cap because the read above succeeded */
    }
    d->mini_block_idx = 0;
    return 0;
}

/* encoding.cc:241-249: the rest of the current miniblock must be present. */
static int delta_eat_final_padding(orc_delta *d, orc_error *err) {
    uint64_t w = d->delta_bit_width;
    if (w != 0 && d->values_current_mini_block > 0) {
        uint64_t remaining = (uint64_t)d->br.len * 8 - d->br.pos;
        uint64_t can = remaining / w;
        if (can < d->values_current_mini_block) {
            set_err(err, ORC_E_DELTA_DATA_EOF, 0, 0);
            return -1;
        }
        d->br.pos += d->values_current_mini_block * w;
    }
    d->values_current_mini_block = 0;
    return 0;
}

/* encoding.cc:207-239: running sum in uint64 wrap-around, truncated to the
 * output type on store (:213). */
int64_t orc_delta_read_batch(orc_delta *d, void *out, int elem_size, int64_t n, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    if (d->values_remaining == 0) return 0;
    int64_t i = 0;
    while (i < n) {
        store_elem(out, elem_size, i, d->last_value);
        ++i;
        --d->values_remaining;
        if (d->values_remaining == 0) {
            if (delta_eat_final_padding(d, err) < 0) return -1;
            break;
        }
        if (d->values_current_mini_block == 0) {
            if (d->mini_block_idx == d->num_mini_blocks) {
                if (delta_init_block(d, err) < 0) return -1;
            }
            d->delta_bit_width = d->widths[d->mini_block_idx];
            if (d->delta_bit_width > 64) {
                /* undefined in the reference (shift >= 64); rejected here */
                set_err(err, ORC_E_DELTA_BAD_WIDTH, d->delta_bit_width, 0);
                return -1;
            }
            d->values_current_mini_block = d->values_per_mini_block;
            ++d->mini_block_idx;
        }
        uint64_t delta = 0;
        if (orc_br_get_batch(&d->br, d->delta_bit_width, &delta, 8, 1) != 1) {
            set_err(err, ORC_E_DELTA_DATA_EOF, 0, 0);
            return -1;
        }
        delta += d->min_delta;
        d->last_value += delta;
        --d->values_current_mini_block;
    }
    return i;
}

int64_t orc_delta_decode(const uint8_t *buf, int64_t len, void *out, int elem_size, int64_t n,
                         int64_t *bytes_left, orc_error *err) {
    orc_delta d;
    if (orc_delta_reset(&d, buf, len, err) < 0) { orc_delta_free(&d); return -1; }
    int64_t r = orc_delta_read_batch(&d, out, elem_size, n, err);
    if (bytes_left) *bytes_left = orc_br_bytes_left(&d.br);
    orc_delta_free(&d);
    return r;
}

/* ---- DELTA_BINARY_PACKED encoder ---- */
static int64_t enc_vlq(uint8_t *out, int64_t pos, int64_t cap, uint64_t v) {      /* BitWriter::PutVlqInt, bit_stream_utils.hh:395-403 */
    while (v & ~(uint64_t)0x7F) {
        if (pos >= cap) return -1;
        out[pos++] = (uint8_t)((v & 0x7F) | 0x80);
        v >>= 7;
    }
    if (pos >= cap) return -1;
    out[pos++] = (uint8_t)(v & 0x7F);
    return pos;
}
static int64_t enc_zigzag(uint8_t *out, int64_t pos, int64_t cap, int64_t v) {    /* PutZigZagVlqInt, :443-446 */
    return enc_vlq(out, pos, cap, ((uint64_t)v << 1) ^ (uint64_t)(v >> 63));
}

int64_t orc_delta_encode(const void *values, int elem_size, int64_t n, uint8_t *out, int64_t cap) {
    enum { BLOCK = 256, MINIBLOCKS = 8, PER_MB = 32 };                             /* encoding.cc:871-873 */
    const int32_t *v32 = (const int32_t *)values;
    const int64_t *v64 = (const int64_t *)values;
#define VAL(i) (elem_size == 4 ? (int64_t)v32[i] : v64[i])
    const uint64_t mask = elem_size == 4 ? 0xFFFFFFFFull : ~0ull;                   /* unsigned_type arithmetic, :893-906 */
    /* flush(): header = block size, miniblocks per block, total, first value (:977-982; 0 when nothing was put) */
    int64_t pos = 0;
    if ((pos = enc_vlq(out, pos, cap, BLOCK)) < 0) return -1;
    if ((pos = enc_vlq(out, pos, cap, MINIBLOCKS)) < 0) return -1;
    if ((pos = enc_vlq(out, pos, cap, (uint64_t)n)) < 0) return -1;
    if ((pos = enc_zigzag(out, pos, cap, n > 0 ? VAL(0) : 0)) < 0) return -1;
    /* put_batch(): value 0 is the first value, the others fill blocks of 256 (:955-967) */
    for (int64_t b0 = 1; b0 < n; b0 += BLOCK) {
        const int64_t cnt = n - b0 < BLOCK ? n - b0 : BLOCK;
        uint64_t deltas[BLOCK];
        for (int64_t i = 0; i < BLOCK; ++i) deltas[i] = 0;                          /* (the reference: uninitialised) */
        for (int64_t i = 0; i < cnt; ++i)                                           /* :893-898 */
            deltas[i] = ((uint64_t)VAL(b0 + i) - (uint64_t)VAL(b0 + i - 1)) & mask;
        /* the smallest delta as a value of signed_type (:900-904) */
#define SIGNED(u) (elem_size == 4 ? (int64_t)(int32_t)(uint32_t)(u) : (int64_t)(u))
        int64_t min_delta = SIGNED(deltas[0]);
        for (int64_t i = 0; i < cnt; ++i) if (SIGNED(deltas[i]) < min_delta) min_delta = SIGNED(deltas[i]);
        uint64_t maxd[MINIBLOCKS] = {0};
        for (int64_t i = 0; i < cnt; ++i) {                                         /* :905-911 */
            deltas[i] = (deltas[i] - (uint64_t)min_delta) & mask;
            if (deltas[i] > maxd[i / PER_MB]) maxd[i / PER_MB] = deltas[i];
        }
        if ((pos = enc_zigzag(out, pos, cap, min_delta)) < 0) return -1;            /* :922 */
        int widths[MINIBLOCKS];
        for (int mb = 0; mb < MINIBLOCKS; ++mb) {                                   /* :912-914, :923-925 */
            int w = 0;
            for (uint64_t m = maxd[mb]; m; m >>= 1) ++w;
            widths[mb] = w;
            if (pos >= cap) return -1;
            out[pos++] = (uint8_t)w;
        }
        for (int mb = 0; mb < MINIBLOCKS && (int64_t)mb * PER_MB < cnt; ++mb) {     /* :926-935: whole miniblocks */
            const int w = widths[mb];
            const int64_t bytes = (int64_t)PER_MB * w / 8;
            if (pos + bytes > cap) return -1;
            memset(out + pos, 0, (size_t)bytes);
            for (int i = 0; i < PER_MB; ++i) {                                      /* BitWriter::PutValue: LSB first */
                const uint64_t x = deltas[mb * PER_MB + i];
                for (int bit = 0; bit < w; ++bit)
                    if ((x >> bit) & 1u) {
                        const int64_t at = (int64_t)i * w + bit;
                        out[pos + (at >> 3)] |= (uint8_t)(1u << (at & 7));
                    }
            }
            pos += bytes;
        }
#undef SIGNED
    }
#undef VAL
    return pos;
}

/* ---------------------------------------------------------------------------
 * RleEncoder (include/parquet4seastar/rle_encoding.hh:150-505): Put() for every value, then Flush().
 * The state machine is restated member for member; the output buffer is taken to be large enough (the reference
 * stops accepting values when fewer than MinBufferSize bytes are left, :486-491; callers size the buffer with
 * MaxBufferSize, src/encoding.cc:783-788).  Returns the bytes written, or -1 when `cap` does not hold them.
 * bit_width 0 .. 32 (values are uint32 here; the reference takes uint64 and widths up to 64).
 * ------------------------------------------------------------------------- */
typedef struct {
    uint8_t *out; int64_t cap, pos;        /* BitWriter: bytes written so far (literal groups are whole bytes) */
    int w;
    uint64_t current_value; int repeat_count;
    uint64_t buffered[8]; int num_buffered;
    int literal_count; int64_t indicator;  /* literal_indicator_byte_ (-1: none reserved) */
    int overflow;
} orc_rle_enc;

static void rle_enc_byte(orc_rle_enc *e, uint8_t b) {
    if (e->pos >= e->cap) { e->overflow = 1; return; }
    e->out[e->pos++] = b;
}
static void rle_enc_flush_literal(orc_rle_enc *e, int update_indicator) {            /* FlushLiteralRun, :384-412 */
    if (e->indicator < 0) {                                                          /* GetNextBytePtr */
        e->indicator = e->pos;
        rle_enc_byte(e, 0);
    }
    if (e->num_buffered > 0) {            /* PutValue x num_buffered: a group of 8 values is bit_width whole bytes */
        uint8_t tmp[64];
        memset(tmp, 0, sizeof(tmp));
        for (int i = 0; i < e->num_buffered; ++i)
            for (int bit = 0; bit < e->w; ++bit)
                if ((e->buffered[i] >> bit) & 1u) {
                    const int at = i * e->w + bit;
                    tmp[at >> 3] |= (uint8_t)(1u << (at & 7));
                }
        const int bytes = (e->num_buffered * e->w + 7) / 8;
        for (int i = 0; i < bytes; ++i) rle_enc_byte(e, tmp[i]);
    }
    e->num_buffered = 0;
    if (update_indicator) {
        const int num_groups = e->literal_count / 8;
        if (e->indicator < e->cap) e->out[e->indicator] = (uint8_t)((num_groups << 1) | 1);
        e->indicator = -1;
        e->literal_count = 0;
    }
}
static void rle_enc_flush_repeated(orc_rle_enc *e) {                                 /* FlushRepeatedRun, :414-427 */
    uint32_t v = (uint32_t)e->repeat_count << 1;                                     /* PutVlqInt */
    while (v & ~0x7Fu) { rle_enc_byte(e, (uint8_t)((v & 0x7F) | 0x80)); v >>= 7; }
    rle_enc_byte(e, (uint8_t)(v & 0x7F));
    for (int i = 0; i < (e->w + 7) / 8; ++i) rle_enc_byte(e, (uint8_t)(e->current_value >> (8 * i)));   /* PutAligned */
    e->num_buffered = 0;
    e->repeat_count = 0;
}
static void rle_enc_flush_buffered(orc_rle_enc *e, int done) {                       /* FlushBufferedValues, :431-461 */
    if (e->repeat_count >= 8) {
        e->num_buffered = 0;
        if (e->literal_count != 0) rle_enc_flush_literal(e, 1);
        return;
    }
    e->literal_count += e->num_buffered;
    const int num_groups = e->literal_count / 8;
    if (num_groups + 1 >= (1 << 6)) rle_enc_flush_literal(e, 1);
    else rle_enc_flush_literal(e, done);
    e->repeat_count = 0;
}
int64_t orc_rle_encode(const uint32_t *values, int64_t n, int w, uint8_t *out, int64_t cap) {
    orc_rle_enc e;
    memset(&e, 0, sizeof(e));
    e.out = out; e.cap = cap; e.w = w; e.indicator = -1;
    for (int64_t i = 0; i < n; ++i) {                                                /* Put, :352-382 */
        const uint64_t value = values[i];
        if (e.current_value == value) {
            ++e.repeat_count;
            if (e.repeat_count > 8) continue;
        } else {
            if (e.repeat_count >= 8) rle_enc_flush_repeated(&e);
            e.repeat_count = 1;
            e.current_value = value;
        }
        e.buffered[e.num_buffered] = value;
        if (++e.num_buffered == 8) rle_enc_flush_buffered(&e, 0);
    }
    /* Flush, :463-484 */
    if (e.literal_count > 0 || e.repeat_count > 0 || e.num_buffered > 0) {
        const int all_repeat = e.literal_count == 0 && (e.repeat_count == e.num_buffered || e.num_buffered == 0);
        if (e.repeat_count > 0 && all_repeat) {
            rle_enc_flush_repeated(&e);
        } else {
            for (; e.num_buffered != 0 && e.num_buffered < 8; ++e.num_buffered) e.buffered[e.num_buffered] = 0;
            e.literal_count += e.num_buffered;
            rle_enc_flush_literal(&e, 1);
            e.repeat_count = 0;
        }
    }
    return e.overflow ? -1 : e.pos;
}

/* Decode ALL lengths of a nested INT32 delta stream in 1000-value batches
 * (encoding.cc:277-287 / :332-342).  Returns a malloc'd array. */
static int32_t *delta_all_lengths(const uint8_t *buf, int64_t len, int64_t *count,
                                  int64_t *bytes_left, orc_error *err) {
    orc_delta d;
    if (orc_delta_reset(&d, buf, len, err) < 0) { orc_delta_free(&d); return NULL; }
    int64_t cap = 1000, got = 0;
    int32_t *v = (int32_t *)malloc((size_t)cap * 4);
    for (;;) {
        if (got + 1000 > cap) { cap = cap * 2 + 1000; v = (int32_t *)realloc(v, (size_t)cap * 4); }
        int64_t r = orc_delta_read_batch(&d, v + got, 4, 1000, err);
        if (r < 0) { free(v); orc_delta_free(&d); return NULL; }
        if (r == 0) break;
        got += r;
    }
    *count = got;
    *bytes_left = orc_br_bytes_left(&d.br);
    orc_delta_free(&d);
    return v;
}

/* ------------------------------------------------------------------ */
/* DELTA_LENGTH_BYTE_ARRAY: encoding.cc:252-294                        */
/* ------------------------------------------------------------------ */

int64_t orc_dlba_decode(const uint8_t *buf, int64_t len, int64_t n,
                        int64_t *offsets, uint8_t *out_bytes, int64_t bytes_cap,
                        orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    int64_t count = 0, left = 0;
    int32_t *lengths = delta_all_lengths(buf, len, &count, &left, err);
    if (!lengths) return -1;
    const uint8_t *vals = buf + (len - left);      /* :289-291 */
    uint64_t vals_left = (uint64_t)left;
    if (n > count) n = count;                      /* :260 */
    int64_t off = 0;
    offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint32_t l = (uint32_t)lengths[i];         /* :262 */
        if (l > vals_left) { set_err(err, ORC_E_DLBA_VALUES_EOF, 0, 0); free(lengths); return -1; }
        if (off + (int64_t)l > bytes_cap) { free(lengths); return -2; }
        memcpy(out_bytes + off, vals, l);
        vals += l; vals_left -= l; off += l;
        offsets[i + 1] = off;
    }
    free(lengths);
    return n;
}

/* ------------------------------------------------------------------ */
/* DELTA_BYTE_ARRAY: encoding.cc:296-362 (single whole-page batch)     */
/* ------------------------------------------------------------------ */

static int64_t dba_core(const uint8_t *buf, int64_t len, int64_t n,
                        int64_t *offsets, uint8_t *out_bytes, int64_t bytes_cap,
                        int64_t *total_bytes, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    int64_t n_prefix = 0, left = 0;
    int32_t *prefix = delta_all_lengths(buf, len, &n_prefix, &left, err);   /* :331-345 */
    if (!prefix) return -1;
    const uint8_t *sbuf = buf + (len - left);
    int64_t slen = left;
    int64_t n_suffix = 0, sleft = 0;
    int32_t *suffix_len = delta_all_lengths(sbuf, slen, &n_suffix, &sleft, err);  /* :347 -> :273-293 */
    if (!suffix_len) { free(prefix); return -1; }
    const uint8_t *vals = sbuf + (slen - sleft);
    uint64_t vals_left = (uint64_t)sleft;
    /* reset() slices EVERY suffix before read_batch runs (:348-358), so a short
     * data section fails even when fewer values are requested. */
    {
        uint64_t vl = vals_left;
        for (int64_t i = 0; i < n_suffix; ++i) {
            uint32_t l = (uint32_t)suffix_len[i];
            if (l > vl) { set_err(err, ORC_E_DLBA_VALUES_EOF, 0, 0); free(prefix); free(suffix_len); return -1; }
            vl -= l;
        }
    }
    if (n > n_suffix) n = n_suffix;                /* :306 */
    int64_t off = 0;
    uint64_t last_len = 0;
    int64_t last_off = 0;
    if (offsets) offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        /* The reference indexes the prefix array without a bound check; more
         * suffixes than prefixes is rejected here (recorded deviation). */
        if (i >= n_prefix) { set_err(err, ORC_E_DBA_BAD_PREFIX, 0, 0); free(prefix); free(suffix_len); return -1; }
        uint32_t p = (uint32_t)prefix[i];          /* :308 */
        uint32_t s = (uint32_t)suffix_len[i];
        if (p > last_len) { set_err(err, ORC_E_DBA_BAD_PREFIX, 0, 0); free(prefix); free(suffix_len); return -1; }
        if (out_bytes) {
            if (off + (int64_t)p + (int64_t)s > bytes_cap) { free(prefix); free(suffix_len); return -2; }
            memmove(out_bytes + off, out_bytes + last_off, p);
            memcpy(out_bytes + off + p, vals, s);
        }
        vals += s;
        last_off = off;
        last_len = (uint64_t)p + s;
        off += (int64_t)p + s;
        if (offsets) offsets[i + 1] = off;
    }
    if (total_bytes) *total_bytes = off;
    free(prefix); free(suffix_len);
    return n;
}

int64_t orc_dba_decode(const uint8_t *buf, int64_t len, int64_t n,
                       int64_t *offsets, uint8_t *out_bytes, int64_t bytes_cap,
                       orc_error *err) {
    return dba_core(buf, len, n, offsets, out_bytes, bytes_cap, NULL, err);
}

int64_t orc_dba_sizes(const uint8_t *buf, int64_t len, int64_t n, int64_t *total_bytes,
                      orc_error *err) {
    return dba_core(buf, len, n, NULL, NULL, 0, total_bytes, err);
}

/* ------------------------------------------------------------------ */
/* BYTE_STREAM_SPLIT: encoding.cc:364-394                              */
/* ------------------------------------------------------------------ */

int64_t orc_bss_decode(const uint8_t *buf, int64_t len, int elem_size, void *out, int64_t n,
                       orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    if (len % elem_size != 0) { set_err(err, ORC_E_BSS_SIZE, 0, 0); return -1; }
    int64_t total = len / elem_size;
    if (n > total) n = total;
    uint8_t *o = (uint8_t *)out;
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < elem_size; ++k)
            o[i * elem_size + k] = buf[(int64_t)k * total + i];
    return n;
}

/* ------------------------------------------------------------------ */
/* PLAIN: encoding.cc:396-466, RLE boolean :510-516                    */
/* ------------------------------------------------------------------ */

int64_t orc_plain_fixed(const uint8_t *buf, int64_t len, int elem_size, void *out, int64_t n) {
    int64_t avail = len / elem_size;                    /* :417 */
    if (n > avail) n = avail;
    if (n > 0) memcpy(out, buf, (size_t)(n * elem_size));
    return n;
}

int64_t orc_plain_boolean(const uint8_t *buf, int64_t len, uint8_t *out, int64_t n) {
    return orc_bitpacked_decode(buf, len, 1, out, 1, n); /* :426-428 */
}

int64_t orc_plain_byte_array(const uint8_t *buf, int64_t len, int64_t n,
                             int64_t *offsets, uint8_t *out_bytes, int64_t bytes_cap,
                             orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    int64_t off = 0;
    offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) {                   /* :430-450 */
        if (len == 0) return i;
        if (len < 4) { set_err(err, ORC_E_PLAIN_BA_LEN_EOF, 4, (uint64_t)len); return -1; }
        uint32_t l;
        memcpy(&l, buf, 4);
        buf += 4; len -= 4;
        if ((int64_t)l > len) { set_err(err, ORC_E_PLAIN_BA_EOF, l, (uint64_t)len); return -1; }
        if (off + (int64_t)l > bytes_cap) return -2;
        memcpy(out_bytes + off, buf, l);
        buf += l; len -= l; off += l;
        offsets[i + 1] = off;
    }
    return n;
}

int64_t orc_plain_flba(const uint8_t *buf, int64_t len, uint32_t fixed_len, int64_t n,
                       uint8_t *out_bytes, int64_t bytes_cap, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    int64_t off = 0;
    for (int64_t i = 0; i < n; ++i) {                   /* :452-466 */
        if (len == 0) return i;
        if ((int64_t)fixed_len > len) { set_err(err, ORC_E_PLAIN_FLBA_EOF, fixed_len, (uint64_t)len); return -1; }
        if (off + (int64_t)fixed_len > bytes_cap) return -2;
        memcpy(out_bytes + off, buf, fixed_len);
        buf += fixed_len; len -= fixed_len; off += fixed_len;
    }
    return n;
}

int64_t orc_rle_boolean(const uint8_t *buf, int64_t len, uint8_t *out, int64_t n) {
    return orc_rle_decode(buf, len, 1, out, 1, n);      /* :510-516: no length prefix skipped */
}

/* ------------------------------------------------------------------ */
/* dictionary: encoding.cc:468-508                                     */
/* ------------------------------------------------------------------ */

int64_t orc_dict_decode(const uint8_t *buf, int64_t len, const void *dict, uint64_t dict_size,
                        int elem_size, void *out, int64_t n, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    if (len == 0) { set_err(err, ORC_E_DICT_EMPTY_PAGE, 0, 0); return -1; }  /* reference reads data[0] */
    int bw = buf[0];
    if (bw > 32) { set_err(err, ORC_E_DICT_BAD_WIDTH, (uint64_t)bw, 0); return -1; }  /* :473-477 */
    orc_rle d;
    orc_rle_init(&d, buf + 1, len - 1, bw);
    uint32_t idx[1000];                                  /* :483 */
    int64_t completed = 0;
    const uint8_t *dp = (const uint8_t *)dict;
    uint8_t *o = (uint8_t *)out;
    while (completed < n) {
        int64_t want = n - completed < 1000 ? n - completed : 1000;
        int64_t got = orc_rle_get_batch(&d, idx, 4, want);
        for (int64_t i = 0; i < got; ++i) {
            if (idx[i] > dict_size) {                    /* :489 (idx == dict_size passes) */
                set_err(err, ORC_E_DICT_INDEX, dict_size, idx[i]);
                return -1;
            }
        }
        for (int64_t i = 0; i < got; ++i)
            memcpy(o + (completed + i) * elem_size, dp + (uint64_t)idx[i] * (uint64_t)elem_size, (size_t)elem_size);
        completed += got;
        if (got < want) return completed;
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* scripted BitReader session (differential tests against oracle/_ref)  */
/* ops[i] = {opcode, arg}: 0 GetValue<u64>(bits) 1 GetAligned<u64>(bytes)
 * 2 vlq32 3 vlq64 4 zigzag32 5 zigzag64 6 bytes_left; results[i]={ok,value} */
/* ------------------------------------------------------------------ */
void orc_br_script(const uint8_t *buf, int64_t len, const int32_t *ops, int nops, int64_t *results) {
    orc_bitreader r;
    orc_br_init(&r, buf, len);
    for (int i = 0; i < nops; ++i) {
        int op = ops[2 * i], arg = ops[2 * i + 1];
        int64_t ok = 0, val = 0;
        switch (op) {
        case 0: { uint64_t v = 0; ok = orc_br_get_batch(&r, arg, &v, 8, 1) == 1; val = (int64_t)v; break; }
        case 1: { uint64_t v = 0; ok = orc_br_get_aligned(&r, arg, 8, &v); val = (int64_t)v; break; }
        case 2: { uint32_t v = 0; ok = orc_br_get_vlq32(&r, &v); val = v; break; }
        case 3: { uint64_t v = 0; ok = orc_br_get_vlq64(&r, &v); val = (int64_t)v; break; }
        case 4: { int32_t v = 0; ok = orc_br_get_zigzag32(&r, &v); val = v; break; }
        case 5: { int64_t v = 0; ok = orc_br_get_zigzag64(&r, &v); val = v; break; }
        default: ok = 1; val = orc_br_bytes_left(&r); break;
        }
        results[2 * i] = ok;
        results[2 * i + 1] = ok ? val : 0;
    }
}

/* ---------------------------------------------------------------------------
 * Snappy (src/compression.cc:49-65 -> libsnappy).  Test infrastructure only.
 * ------------------------------------------------------------------------- */
int64_t orc_snappy_decompress(const uint8_t *buf, int64_t len, uint8_t *out, int64_t out_cap, orc_error *err) {
    set_err(err, ORC_OK, 0, 0);
    /* GetUncompressedLength: varint32, at most 5 bytes, the 5th may carry 4 bits */
    uint32_t expected = 0;
    int64_t ip = 0;
    for (uint32_t shift = 0;; shift += 7) {
        if (shift >= 32 || ip >= len) { set_err(err, ORC_E_SNAPPY_LENGTH, 0, 0); return -1; }
        const uint32_t c = buf[ip++];
        const uint32_t val = c & 0x7Fu;
        if (((val << shift) >> shift) != val) { set_err(err, ORC_E_SNAPPY_LENGTH, 0, 0); return -1; }
        expected |= val << shift;
        if (c < 128) break;
    }
    if (out_cap < (int64_t)expected) { set_err(err, ORC_E_SNAPPY_BUFFER, 0, 0); return -1; }   /* compression.cc:56 */
    uint64_t op = 0;
    while (ip < len) {
        const uint32_t c = buf[ip];
        uint64_t n;            /* bytes the element produces */
        if ((c & 3u) == 0) {
            int64_t hdr = 1;
            n = (c >> 2) + 1;
            if (n > 60) {
                const int nb = (int)(n - 60);
                if (ip + 1 + nb > len) goto fail;
                uint32_t v = 0;
                for (int i = 0; i < nb; ++i) v |= (uint32_t)buf[ip + 1 + i] << (8 * i);
                n = (uint64_t)v + 1;
                hdr = 1 + nb;
            }
            if ((uint64_t)(len - ip - hdr) < n) goto fail;       /* the literal runs past the input */
            if ((uint64_t)expected - op < n) goto fail;
            memcpy(out + op, buf + ip + hdr, (size_t)n);
            ip += hdr + (int64_t)n;
            op += n;
        } else {
            uint64_t offset;
            int64_t hdr;
            if ((c & 3u) == 1) {
                if (ip + 2 > len) goto fail;
                n = ((c >> 2) & 7u) + 4;
                offset = ((uint64_t)(c >> 5) << 8) | buf[ip + 1];
                hdr = 2;
            } else if ((c & 3u) == 2) {
                if (ip + 3 > len) goto fail;
                n = (c >> 2) + 1;
                offset = (uint64_t)buf[ip + 1] | ((uint64_t)buf[ip + 2] << 8);
                hdr = 3;
            } else {
                if (ip + 5 > len) goto fail;
                n = (c >> 2) + 1;
                offset = (uint64_t)buf[ip + 1] | ((uint64_t)buf[ip + 2] << 8) | ((uint64_t)buf[ip + 3] << 16) |
                         ((uint64_t)buf[ip + 4] << 24);
                hdr = 5;
            }
            if (offset == 0 || offset > op) goto fail;
            if ((uint64_t)expected - op < n) goto fail;
            for (uint64_t i = 0; i < n; ++i) out[op + i] = out[op - offset + i];   /* byte order matters: runs repeat */
            ip += hdr;
            op += n;
        }
    }
    if (op != expected) goto fail;
    return (int64_t)op;
fail:
    set_err(err, ORC_E_SNAPPY_FAILED, 0, 0);
    return -1;
}
