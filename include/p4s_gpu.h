/*
 * p4s_gpu.h - C ABI of the MI355X (gfx950) Parquet page-value decoders.
 *
 * This is the drop-in boundary for the page-decoding path of parquet4seastar:
 * every entry point decodes a BATCH of independent pages with one kernel
 * family, reproducing what one `reset(page) + read_batch(n, out)` pair of the
 * reference decoder cited next to it returns for each page.  "ref:" citations
 * are file:line in the reference tree.  Around that path, in the same batched
 * form: Snappy page decompression, the page-header walk of column chunks,
 * level-driven scatter to positional columns and the DELTA_BINARY_PACKED
 * encoder.
 *
 * Conventions
 *  - All `d_*` pointers are DEVICE pointers on the context's GPU; they stay
 *    owned by the caller.  `d_in` must be 16-byte aligned and `in_bytes` is the
 *    size of the whole input allocation (kernels never read past it).
 *  - Work is enqueued on the context's stream and returns without waiting for it;
 *    call p4s_ctx_sync() (or synchronise the stream) before reading results.
 *  - Context workspaces.  Three families keep side tables that the kernels of one
 *    call hand to each other in device memory the CONTEXT owns, one workspace per
 *    family: p4s_decode_rle / p4s_decode_levels / p4s_decode_dictionary (batches of
 *    up to 512 pages and 1 GiB, which are split over many workgroups: about 2.2
 *    bytes per input byte, never more than P4S_WORKSPACE_MAX_BYTES - a batch that
 *    would need more is decoded without the split), p4s_decode_delta_binary_packed
 *    (batches of up to 64 pages: under 1 KiB per page) and p4s_delta_byte_array_sizes /
 *    p4s_decode_delta_byte_array (128 bytes per page + 4 per 256 offsets slots).
 *    A call whose batch needs more room than any earlier call of its family on
 *    this context waits for the stream, frees and reallocates (hipFree / hipMalloc:
 *    such a call can block and cannot be captured in a HIP graph); a call that fits
 *    what is there never blocks.  Warm a context up with its largest batch shape, or
 *    use p4s_decode_delta_byte_array_static, which allocates nothing.  A workspace
 *    remembers the stream it was last used on: after p4s_ctx_set_stream the next
 *    call of the family first waits for that stream, so tables that queued launches
 *    still read are never overwritten.  The workspaces live until p4s_ctx_destroy.
 *  - No exceptions cross this boundary.  Functions return P4S_OK or a negative
 *    P4S_ERR_* for API misuse / HIP failures; per-page decode errors (what the
 *    reference reports by throwing parquet_exception) are written to
 *    `d_status[i].code` as positive P4S_E_* codes with the message arguments
 *    in aux0/aux1.  p4s_format_error() renders the reference's message text.
 *  - Pages are independent: a multi-GPU caller shards the page list and gives
 *    each GPU its own context.  There is no cross-GPU step.
 */
#ifndef P4S_GPU_H
#define P4S_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- API return codes ---- */
#define P4S_OK 0
#define P4S_ERR_INVALID_ARG (-1)
#define P4S_ERR_HIP (-2)
#define P4S_ERR_NO_DEVICE (-3)
#define P4S_ERR_OOM (-4)

/* Upper bound of one context workspace (see "Context workspaces" above). */
#define P4S_WORKSPACE_MAX_BYTES (2048ull << 20)

/* ---- per-page decode error codes (p4s_page_status.code) ----
 * Text of each: p4s_format_error().  Codes marked "deviation" have no
 * counterpart in the reference, whose behaviour is undefined for that input. */
enum p4s_page_error {
    P4S_E_NONE = 0,
    P4S_E_DELTA_HEADER_EOF = 1,        /* ref: src/encoding.cc:181-196 */
    P4S_E_DELTA_ZERO_MINIBLOCKS = 2,   /* ref: src/encoding.cc:187-189 */
    P4S_E_DELTA_BLOCK_HEADER_EOF = 3,  /* ref: src/encoding.cc:159-167 */
    P4S_E_DELTA_DATA_EOF = 4,          /* ref: src/encoding.cc:231-233,244-246 */
    P4S_E_DELTA_BAD_WIDTH = 5,         /* deviation: miniblock width > 64 (aux0 = width) */
    P4S_E_DELTA_BAD_SHAPE = 6,         /* deviation: block_size < n_miniblocks */
    P4S_E_DLBA_VALUES_EOF = 7,         /* ref: src/encoding.cc:263-266 */
    P4S_E_DBA_BAD_PREFIX = 8,          /* ref: src/encoding.cc:310-312 */
    P4S_E_BSS_SIZE = 9,                /* ref: src/encoding.cc:386-389 */
    P4S_E_DICT_BAD_WIDTH = 10,         /* ref: src/encoding.cc:474-477 (aux0 = width) */
    P4S_E_DICT_INDEX = 11,             /* ref: src/encoding.cc:489-492 (aux0 = dict size, aux1 = index) */
    P4S_E_PLAIN_BA_LEN_EOF = 12,       /* ref: src/encoding.cc:435-438 */
    P4S_E_PLAIN_BA_EOF = 13,           /* ref: src/encoding.cc:442-445 */
    P4S_E_PLAIN_FLBA_EOF = 14,         /* ref: src/encoding.cc:457-461 */
    P4S_E_LEVELS_EOF = 15,             /* ref: src/encoding.cc:36-48,57-60 */
    P4S_E_LEVELS_NEGATIVE = 16,        /* ref: src/encoding.cc:42-44 */
    P4S_E_LEVELS_BITPACKED_LEN = 17,   /* ref: src/encoding.cc:54-56 */
    P4S_E_LEVELS_ENCODING = 18,        /* ref: src/encoding.cc:64 */
    P4S_E_LEVELS_V2_LEN = 19,          /* ref: src/encoding.cc:71-74 */
    P4S_E_NO_DICT = 20,                /* ref: src/encoding.cc:542-544 (raised by the host layer) */
    P4S_E_BAD_ENCODING = 21,           /* ref: src/encoding.cc:547-583 (raised by the host layer) */
    P4S_E_DICT_EMPTY_PAGE = 22,        /* deviation: ref reads data[0] of an empty page, src/encoding.cc:470-473 */
    P4S_E_OUT_CAPACITY = 23,           /* build-specific: the caller's output slot is too small (aux0 = bytes needed) */
    P4S_E_SNAPPY_LENGTH = 24,          /* ref: src/compression.cc:53-55 */
    P4S_E_SNAPPY_BUFFER = 25,          /* ref: src/compression.cc:56-58 */
    P4S_E_SNAPPY_FAILED = 26,          /* ref: src/compression.cc:61-63 */
    /* p4s_walk_page_headers: what deserialize_thrift_msg and page_reader::next_page report
     * (include/parquet4seastar/thrift_serdes.hh:117-154, src/column_chunk_reader.cc:27-50) */
    P4S_E_THRIFT_EOF = 27,             /* the chunk ends inside a header (aux0 = bytes the chunk had from the header on) */
    P4S_E_THRIFT_VARINT = 28,          /* a variable-length integer of more than 10 bytes */
    P4S_E_THRIFT_DEPTH = 29,           /* structs / containers nested more than 32 deep */
    P4S_E_THRIFT_WIRE_TYPE = 30,       /* aux0 = the invalid compact-protocol type */
    P4S_E_THRIFT_NOT_INT = 31,         /* an integer field of PageHeader carries another type */
    P4S_E_THRIFT_INVALID_DATA = 32,    /* a required field is missing (TProtocolException: Invalid data) */
    P4S_E_PAGE_NEGATIVE_SIZE = 33,     /* ref: src/column_chunk_reader.cc:33-36 (aux0 = compressed_page_size) */
    P4S_E_PAGE_EOF = 34,               /* ref: src/column_chunk_reader.cc:41-46 (aux0 = expected, aux1 = got) */
    P4S_E_ENCODE_RANGE = 35            /* deviation: a value does not fit the bit width, or the width is above 32 (aux0 = bit width); the reference asserts, include/parquet4seastar/rle_encoding.hh:353 */
};

/* One page of a batch.  `in_offset`/`in_size` locate the ENCODED section inside
 * d_in (for a v1 data page: the bytes after the levels).  `out_offset` is in
 * ELEMENTS of the output array (for string decoders: first slot in d_offsets).
 * `aux_offset` is the page's first byte in the d_bytes buffer (string decoders).
 * `num_values` is the n of read_batch(n, out).  param0/param1/flags are
 * encoding specific and documented per entry point. */
typedef struct p4s_page_desc {
    uint64_t in_offset;
    uint64_t in_size;
    uint64_t out_offset;
    uint64_t aux_offset;
    uint32_t num_values;
    uint32_t param0;
    uint32_t param1;
    uint32_t flags;
} p4s_page_desc;

typedef struct p4s_page_status {
    int32_t code;             /* P4S_E_NONE or a p4s_page_error */
    uint32_t reserved;
    uint64_t values_decoded;  /* what read_batch would have returned */
    uint64_t aux0;
    uint64_t aux1;
    uint64_t bytes_consumed;  /* levels v1: bytes reset_v1 returns; DELTA: size - bytes_left() */
    uint64_t out_bytes;       /* string decoders: bytes produced for this page */
} p4s_page_status;

typedef struct p4s_ctx p4s_ctx;

/* ---- context / plumbing ---- */
int p4s_ctx_create(int device, p4s_ctx **out);
void p4s_ctx_destroy(p4s_ctx *ctx);
/* Use an existing hipStream_t (passed as void*); NULL restores the context's own stream. */
int p4s_ctx_set_stream(p4s_ctx *ctx, void *hip_stream);
int p4s_ctx_sync(p4s_ctx *ctx);
const char *p4s_last_error(p4s_ctx *ctx);
/* Renders the reference's exception text for a page status into buf. Returns length. */
int p4s_format_error(const p4s_page_status *st, char *buf, size_t buf_len);

int p4s_dev_alloc(p4s_ctx *ctx, uint64_t bytes, void **d_ptr);
int p4s_dev_free(p4s_ctx *ctx, void *d_ptr);
int p4s_host_alloc(p4s_ctx *ctx, uint64_t bytes, void **h_ptr);   /* pinned */
int p4s_host_free(p4s_ctx *ctx, void *h_ptr);
int p4s_copy_h2d(p4s_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes);  /* async on the stream */
int p4s_copy_d2h(p4s_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes);  /* async on the stream */
int p4s_memset(p4s_ctx *ctx, void *d_dst, int value, uint64_t bytes);
/* HIP-event stopwatch on the context's stream (kernel time of what is enqueued in between). */
int p4s_timer_start(p4s_ctx *ctx);
int p4s_timer_stop(p4s_ctx *ctx, float *elapsed_ms);   /* synchronises */

/* ---- decoders ---- */

/* BYTE_STREAM_SPLIT, elem_size 4 (FLOAT) or 8 (DOUBLE).
 * ref: byte_stream_split_decoder<T>, src/encoding.cc:364-394. */
int p4s_decode_byte_stream_split(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                 const p4s_page_desc *d_pages, uint32_t n_pages,
                                 void *d_out, p4s_page_status *d_status, int elem_size);

/* PLAIN fixed width (INT32/INT64/INT96/FLOAT/DOUBLE): floor(in_size/elem_size) values, capped by n.
 * ref: plain_decoder_trivial<T>, src/encoding.cc:415-424. */
int p4s_decode_plain_fixed(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                           const p4s_page_desc *d_pages, uint32_t n_pages,
                           void *d_out, p4s_page_status *d_status, int elem_size);

/* PLAIN BOOLEAN: LSB-first bits to one uint8 each.
 * ref: plain_decoder_boolean, src/encoding.cc:401-403,426-428. */
int p4s_decode_plain_boolean(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                             const p4s_page_desc *d_pages, uint32_t n_pages,
                             uint8_t *d_out, p4s_page_status *d_status);

/* DELTA_BINARY_PACKED, elem_size 4 (INT32) or 8 (INT64); running sum in
 * unsigned wrap-around, truncated to the element type.
 * ref: delta_binary_packed_decoder<T>, src/encoding.cc:141-250. */
int p4s_decode_delta_binary_packed(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                   const p4s_page_desc *d_pages, uint32_t n_pages,
                                   void *d_out, p4s_page_status *d_status, int elem_size);

/* RLE / bit-packed hybrid, one RleDecoder::GetBatch<T>(out, n) per page;
 * param0 = bit width (0..32), out_elem_size = sizeof(T) in {1,2,4}.
 * With out_elem_size 1 and width 1 this is rle_decoder_boolean.
 * ref: RleDecoder::GetBatch/NextCounts, include/parquet4seastar/rle_encoding.hh:282-348;
 *      rle_decoder_boolean, src/encoding.cc:510-516. */
int p4s_decode_rle(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                   const p4s_page_desc *d_pages, uint32_t n_pages,
                   void *d_out, p4s_page_status *d_status, int out_elem_size);

/* Definition / repetition levels of a data page.  param0 = max level,
 * param1 = num_values of the page header, flags: P4S_LEVELS_* below.
 * status.bytes_consumed = what reset_v1 returns; status.aux0 = number of decoded
 * levels equal to max level (the value count of column_chunk_reader.hh:134-139);
 * status.aux1 = number of decoded levels above max level (column_chunk_reader.hh:124-133
 * rejects a batch holding one).
 * ref: level_decoder, include/parquet4seastar/encoding.hh:42-82, src/encoding.cc:26-79. */
#define P4S_LEVELS_V1_RLE 0u         /* [int32 length][hybrid] */
#define P4S_LEVELS_V1_BIT_PACKED 1u  /* ceil(n*w/8) bytes, LSB-first (as the reference reads it) */
#define P4S_LEVELS_V2 2u             /* bare hybrid */
int p4s_decode_levels(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                      const p4s_page_desc *d_pages, uint32_t n_pages,
                      void *d_out, p4s_page_status *d_status, int out_elem_size);

/* Copies status[i].aux0 of a levels decode into pages[i].num_values of a value
 * page batch (the values_to_read of column_chunk_reader.hh:134-140), on device. */
int p4s_pages_set_num_values_from_levels(p4s_ctx *ctx, p4s_page_desc *d_value_pages,
                                         const p4s_page_status *d_level_status, uint32_t n_pages);

/* RLE_DICTIONARY / PLAIN_DICTIONARY data pages over a fixed-width dictionary
 * (elem_size 1, 4, 8 or 12).  Page byte 0 is the index width.  d_dict must hold
 * dict_size + 1 elements: the reference accepts index == dict_size
 * (src/encoding.cc:489) and reads past the end; the extra element is what such
 * an index returns here.
 * ref: dict_decoder<T>, src/encoding.cc:468-508. */
int p4s_decode_dictionary(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_page_desc *d_pages, uint32_t n_pages,
                          const void *d_dict, uint64_t dict_size, int elem_size,
                          void *d_out, p4s_page_status *d_status);

/* String decoders produce Arrow-style (offsets, bytes) instead of the reference's one
 * temporary_buffer per value: page i writes values_decoded+1 int64 offsets (relative to the
 * page's own bytes, first = 0) at d_offsets[out_offset ...] and its bytes at
 * d_bytes[aux_offset ...].  param0 of each page = capacity in bytes of its d_bytes slot and
 * bytes_cap = size of the whole d_bytes allocation: a page that needs more is not written and
 * reports P4S_E_OUT_CAPACITY with status.out_bytes = the size it needs.
 * status.out_bytes = bytes produced; DELTA_LENGTH_BYTE_ARRAY output never exceeds in_size,
 * PLAIN never exceeds in_size - 4 * values. */

/* ref: delta_length_byte_array_decoder, src/encoding.cc:252-294. */
int p4s_decode_delta_length_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                       const p4s_page_desc *d_pages, uint32_t n_pages,
                                       int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                       p4s_page_status *d_status);

/* ref: delta_byte_array_decoder, src/encoding.cc:296-362.  Parity is with ONE whole-page
 * read_batch: the reference's decoder re-reads the first strings on every further call
 * (src/encoding.cc:306-309 index with i instead of _current_idx + i).
 * The decoded size of a page is not bounded by its encoded size (shared prefixes expand), so
 * p4s_delta_byte_array_sizes first reports values_decoded and out_bytes per page; the caller
 * lays out d_bytes and calls p4s_decode_delta_byte_array.
 * d_scratch: 8 bytes per offsets slot (same slot numbering as d_offsets: page i uses slots
 * out_offset .. out_offset + values), contents undefined afterwards.
 * Both calls run several kernels (the two length streams of every page side by side; offsets and checks per page;
 * for the decode then the bytes, many wavefronts per page) and keep a small side table between them (128 bytes per
 * page + 4 per 256 slots) in memory the context owns: the first call of a given size may allocate, later ones do not. */
int p4s_delta_byte_array_sizes(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                               const p4s_page_desc *d_pages, uint32_t n_pages,
                               void *d_scratch, uint64_t scratch_bytes, p4s_page_status *d_status);
int p4s_decode_delta_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                const p4s_page_desc *d_pages, uint32_t n_pages,
                                int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                void *d_scratch, uint64_t scratch_bytes, p4s_page_status *d_status);

/* p4s_decode_delta_byte_array for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): the same kernels and the same results, with the scratch in a
 * static device array of the library instead of a caller-provided workspace.  It has no counterpart in the
 * reference.  It calls no hipMalloc / hipFree and no allocating library routine.
 *  - n_offset_slots = number of int64 slots of d_offsets the batch spans (the largest out_offset + num_values + 1 of
 *    its pages); more than P4S_STATIC_SCRATCH_SLOTS, or more than P4S_STATIC_SCRATCH_PAGES pages, is rejected with
 *    P4S_ERR_INVALID_ARG (p4s_last_error says why); a page whose slots lie past n_offset_slots reports
 *    P4S_E_OUT_CAPACITY.
 *  - the array costs P4S_STATIC_SCRATCH_BYTES on every GPU the library is loaded on, until the process ends.
 *  - every call on one GPU uses that GPU's one array (its contents after a call are undefined, nothing needs
 *    clearing between calls): calls that overlap on one GPU, from any context or stream, race.  Serialise them. */
#define P4S_STATIC_SCRATCH_SLOTS (2u << 20)                       /* 2Mi strings (+ one slot per page) per call */
#define P4S_STATIC_SCRATCH_PAGES (1u << 15)                       /* pages per call */
#define P4S_STATIC_SCRATCH_PAGE_BYTES 128u                        /* per-page records the kernels hand to each other */
/* 8 bytes per slot, 128 per page, 4 per 256 slots: 20 MiB + 36 KiB per GPU */
#define P4S_STATIC_SCRATCH_BYTES (8ull * P4S_STATIC_SCRATCH_SLOTS + (unsigned long long)P4S_STATIC_SCRATCH_PAGE_BYTES * P4S_STATIC_SCRATCH_PAGES + 4ull * (P4S_STATIC_SCRATCH_SLOTS / 256 + 1024))
int p4s_decode_delta_byte_array_static(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                       const p4s_page_desc *d_pages, uint32_t n_pages,
                                       int64_t *d_offsets, uint64_t n_offset_slots, uint8_t *d_bytes, uint64_t bytes_cap,
                                       p4s_page_status *d_status);

/* ref: plain_decoder_byte_array, src/encoding.cc:405-408,430-450. */
int p4s_decode_plain_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                const p4s_page_desc *d_pages, uint32_t n_pages,
                                int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                p4s_page_status *d_status);

/* ref: plain_decoder_fixed_len_byte_array, src/encoding.cc:410-413,452-466; param0 = type length.
 * Output: values_decoded * param0 bytes at d_bytes[aux_offset ...]. */
int p4s_decode_plain_fixed_len_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                          const p4s_page_desc *d_pages, uint32_t n_pages,
                                          uint8_t *d_bytes, uint64_t bytes_cap,
                                          p4s_page_status *d_status);

/* Snappy-compressed pages -> their bytes, one page per descriptor: in_offset / in_size = the compressed bytes,
 * out_offset = first byte of the page in d_out (BYTES, any alignment), param0 = the uncompressed size the page
 * header announces (the size of the reference's output buffer).  out_bytes = size of the d_out allocation.
 * status.values_decoded = status.out_bytes = bytes produced.  The three failures of the reference are reported
 * with its texts (P4S_E_SNAPPY_LENGTH, P4S_E_SNAPPY_BUFFER, P4S_E_SNAPPY_FAILED); a page that fails may have
 * written part of its slot.
 * ref: snappy_compressor::decompress, src/compression.cc:49-65 (libsnappy's GetUncompressedLength and
 *      RawUncompress), called per page at src/column_chunk_reader.cc:104-110. */
int p4s_decompress_snappy(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_page_desc *d_pages, uint32_t n_pages,
                          uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status);

/* Positional column from a decoded page: one value slot per level (nulls zero-filled) plus a validity bitmap.
 * For page i: out_offset = first slot of the page in d_def, d_values and d_out, num_values = levels of the page,
 * param0 = max definition level, param1 = dense values available at d_values[out_offset ...] (values_decoded of the
 * value decode), aux_offset = first 32-bit word of the page's bitmap in d_validity (bit s = slot s of the page; the
 * page owns ceil(num_values / 32) words).  d_out must not alias d_values.  elem_size 1, 4, 8 or 12.
 * status.values_decoded = levels, aux0 = non-null count, aux1 = 1 when there are more non-null levels than values.
 * ref: the per-value loop of the triplet consumer, include/parquet4seastar/record_reader.hh:449-466, and the
 *      count of column_chunk_reader.hh:134-144; the reference has no bulk form of it. */
int p4s_scatter_by_levels(p4s_ctx *ctx, const int32_t *d_def, const void *d_values,
                          const p4s_page_desc *d_pages, uint32_t n_pages, int elem_size,
                          void *d_out, uint32_t *d_validity, p4s_page_status *d_status);

/* One page of a column chunk as page_reader::next_page sees it: where its header and its contents lie and what the
 * Thrift PageHeader says (doc/parquet/parquet.thrift:507-659).  Fields of a page-type struct that the page does not
 * carry are 0.  For a v1 data page of an uncompressed chunk the two level-section length prefixes that
 * level_decoder::reset_v1 reads (src/encoding.cc:35-50) are included, so that a caller can cut the page into its
 * sections without touching the chunk: v1_rep_prefix = the int32 at contents[0] when the chunk has repetition levels
 * (P4S_CHUNK_HAS_REP) and their encoding is RLE, v1_def_prefix = the int32 that follows the repetition section
 * likewise; -1 where not applicable or not readable. */
typedef struct p4s_page_info {
    uint64_t header_offset;                   /* chunk-relative */
    uint64_t body_offset;                     /* chunk-relative: first byte of the page contents */
    int32_t type;                             /* PageType: 0 DATA_PAGE, 1 INDEX_PAGE, 2 DICTIONARY_PAGE, 3 DATA_PAGE_V2 */
    int32_t uncompressed_page_size;
    int32_t compressed_page_size;
    int32_t num_values;                       /* data_page_header / dictionary_page_header / data_page_header_v2 */
    int32_t encoding;
    int32_t definition_level_encoding;        /* v1 */
    int32_t repetition_level_encoding;        /* v1 */
    int32_t definition_levels_byte_length;    /* v2 */
    int32_t repetition_levels_byte_length;    /* v2 */
    int32_t num_nulls;                        /* v2 */
    int32_t num_rows;                         /* v2 */
    uint32_t flags;                           /* P4S_PAGE_* */
    int32_t v1_rep_prefix;
    int32_t v1_def_prefix;
    uint32_t reserved[2];
} p4s_page_info;                              /* 80 bytes */
enum {
    P4S_PAGE_HAS_DATA_HEADER = 1,             /* PageHeader.data_page_header is set */
    P4S_PAGE_HAS_DICT_HEADER = 2,             /* ... dictionary_page_header */
    P4S_PAGE_HAS_DATA_HEADER_V2 = 4,          /* ... data_page_header_v2 */
    P4S_PAGE_IS_COMPRESSED = 8,               /* v2 is_compressed (default true when absent) */
    P4S_PAGE_IS_SORTED = 16,                  /* dictionary is_sorted */
    P4S_PAGE_HAS_CRC = 32
};
typedef struct p4s_chunk_desc {
    uint64_t offset;                          /* of the column chunk in d_in */
    uint64_t size;
    uint64_t first_page;                      /* index in d_pages of the chunk's first record */
    uint32_t max_pages;                       /* records the chunk may write */
    uint32_t flags;                           /* P4S_CHUNK_* */
} p4s_chunk_desc;                             /* 32 bytes */
enum { P4S_CHUNK_HAS_REP = 1, P4S_CHUNK_HAS_DEF = 2, P4S_CHUNK_UNCOMPRESSED = 4 };

/* Walks the page headers of n_chunks column chunks that lie in device memory and writes one p4s_page_info per page:
 * the page index that page_reader::next_page builds one header at a time on the host.  d_status[c]: code 0 and
 * values_decoded = pages of the chunk, or the error that ends the walk (pages before it are valid: the reference
 * throws when it reaches that page), with values_decoded = pages written, bytes_consumed = chunk offset of the
 * header that failed.  More pages than max_pages: P4S_E_OUT_CAPACITY (aux0 = max_pages).
 * ref: page_reader::next_page, src/column_chunk_reader.cc:27-50; deserialize_thrift_msg,
 *      include/parquet4seastar/thrift_serdes.hh:117-154; PageHeader, doc/parquet/parquet.thrift:507-659. */
int p4s_walk_page_headers(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_chunk_desc *d_chunks, uint32_t n_chunks,
                          p4s_page_info *d_pages, p4s_page_status *d_status);

/* DELTA_BINARY_PACKED ENCODER (the write path's counterpart of p4s_decode_delta_binary_packed; the first of the
 * value encoders): page i = the n = num_values values d_values[in_offset .. in_offset + n) (in_offset counts
 * ELEMENTS of elem_size 4 = INT32 or 8 = INT64), written as the page one put_batch(values, n) + flush() of the
 * reference's encoder produces (blocks of 256 values in 8 miniblocks) to d_out + out_offset (bytes), param0 = the
 * capacity of that slot.  P4S_DELTA_ENCODE_BOUND(n, elem_size) bytes always suffice.  The reference writes the
 * uninitialised delta slots behind a page's last value as the padding of that value's miniblock; here the padding
 * bits are 0.  status: values_decoded = values encoded, out_bytes = encoded size, bytes_consumed = n * elem_size;
 * P4S_E_OUT_CAPACITY (aux0 = bytes needed) when the slot is too small.
 * ref: delta_binary_packed_encoder<T>::put_batch / flush_block / flush, src/encoding.cc:862-996;
 *      BitWriter::PutVlqInt / PutZigZagVlqInt / PutValue, include/parquet4seastar/bit_stream_utils.hh:395-403,443-446. */
#define P4S_DELTA_ENCODE_BOUND(n, elem_size) \
    (32ull + (((uint64_t)(n) + 255ull) / 256ull) * (18ull + 256ull * (uint64_t)(elem_size)))
int p4s_encode_delta_binary_packed(p4s_ctx *ctx, const void *d_values, uint64_t n_values,
                                   const p4s_page_desc *d_pages, uint32_t n_pages,
                                   uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status, int elem_size);

/* RLE / BIT-PACKED HYBRID ENCODER (levels and dictionary indices on the write path; the counterpart of p4s_decode_rle):
 * page i = the n = num_values uint32 values d_values[in_offset .. in_offset + n) (in_offset counts elements), written
 * as the stream the reference's RleEncoder produces for Put(value) x n + Flush() at bit width param1 (0 .. 32) to
 * d_out + out_offset (bytes), param0 = the capacity of that slot: the hybrid stream alone, without the length prefix
 * or bit-width byte its callers put in front.  P4S_RLE_ENCODE_BOUND(n, width) bytes always suffice (the reference's
 * MaxBufferSize + MinBufferSize, as src/encoding.cc:783-784 sizes its sink).  status: values_decoded = values
 * encoded, out_bytes = encoded size, bytes_consumed = 4 * n; P4S_E_OUT_CAPACITY (aux0 = bytes needed) when the slot
 * is too small; P4S_E_ENCODE_RANGE when a value has bits above the width (the reference asserts).
 * ref: RleEncoder::Put / FlushBufferedValues / FlushLiteralRun / FlushRepeatedRun / Flush,
 *      include/parquet4seastar/rle_encoding.hh:352-484. */
#define P4S_RLE_ENCODE_BOUND(n, width) \
    ((((uint64_t)(n) + 7ull) / 8ull) * (1ull + (uint64_t)(width)) + 64ull * (uint64_t)(width) + 16ull)
int p4s_encode_rle(p4s_ctx *ctx, const uint32_t *d_values, uint64_t n_values,
                   const p4s_page_desc *d_pages, uint32_t n_pages,
                   uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status);

/* Library version string. */
const char *p4s_version(void);

#ifdef __cplusplus
}
#endif
#endif /* P4S_GPU_H */
