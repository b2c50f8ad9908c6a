// C ABI of the gfx950 page decoders (declarations and contracts: include/p4s_gpu.h).
// Host side only enqueues kernels on the context's stream; nothing here blocks
// except the calls documented as synchronising.
#include "p4s_abi_internal.hh"

#include <cstdio>
#include <cstring>
#include <new>

#include "k_simple.cuh"
#include "k_scatter.cuh"
#include "k_headers.cuh"

static_assert(p4s::E_OUT_CAPACITY == P4S_E_OUT_CAPACITY, "error code mismatch");

using p4s_abi::check_common;
using p4s_abi::fail;

namespace p4s_abi {

int ensure_workspace(p4s_ctx *ctx, int id, uint64_t bytes) {
    p4s_workspace &ws = ctx->ws[id];
    if (ws.used && ws.last_stream != ctx->stream) {
        // launches queued on the stream that used the workspace last may still read it
        P4S_HIP(ctx, hipStreamSynchronize(ws.last_stream));
    }
    if (bytes > ws.bytes) {
        if (ws.ptr) {
            if (ws.used) P4S_HIP(ctx, hipStreamSynchronize(ws.last_stream));
            P4S_HIP(ctx, hipFree(ws.ptr));
            ws.ptr = nullptr;
            ws.bytes = 0;
        }
        const uint64_t want = std::max<uint64_t>(bytes, 1u << 20);
        hipError_t e = hipMalloc(&ws.ptr, want);
        if (e != hipSuccess) return fail(ctx, P4S_ERR_OOM, "hipMalloc(workspace)", e);
        ws.bytes = want;
    }
    ws.last_stream = ctx->stream;
    ws.used = true;
    return P4S_OK;
}

}  // namespace p4s_abi

extern "C" {

const char *p4s_version(void) { return "parquet4seastar_amd 0.1 (gfx950)"; }

int p4s_ctx_create(int device, p4s_ctx **out) {
    if (!out) return P4S_ERR_INVALID_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return P4S_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return P4S_ERR_INVALID_ARG;
    p4s_ctx *ctx = new (std::nothrow) p4s_ctx();
    if (!ctx) return P4S_ERR_OOM;
    ctx->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreate(&ctx->ev0) != hipSuccess || hipEventCreate(&ctx->ev1) != hipSuccess) {
        delete ctx;
        return P4S_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    // this device's copy of the static scratch array (the device is current: hipSetDevice above)
    void *sp = nullptr;
    if (p4s_abi::static_scratch_address(&sp) != P4S_OK || !sp) {
        p4s_ctx_destroy(ctx);
        return P4S_ERR_HIP;
    }
    ctx->static_scratch = sp;
    *out = ctx;
    return P4S_OK;
}

void p4s_ctx_destroy(p4s_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (p4s_workspace &ws : ctx->ws)
        if (ws.ptr) (void)hipFree(ws.ptr);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int p4s_ctx_set_stream(p4s_ctx *ctx, void *hip_stream) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    ctx->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return P4S_OK;
}

int p4s_ctx_sync(p4s_ctx *ctx) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    P4S_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return P4S_OK;
}

const char *p4s_last_error(p4s_ctx *ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }

int p4s_dev_alloc(p4s_ctx *ctx, uint64_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(ctx, P4S_ERR_OOM, "hipMalloc", e);
    return P4S_OK;
}

int p4s_dev_free(p4s_ctx *ctx, void *d_ptr) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    P4S_HIP(ctx, hipFree(d_ptr));
    return P4S_OK;
}

int p4s_host_alloc(p4s_ctx *ctx, uint64_t bytes, void **h_ptr) {
    if (!ctx || !h_ptr) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(h_ptr, bytes ? bytes : 16, hipHostMallocDefault);
    if (e != hipSuccess) return fail(ctx, P4S_ERR_OOM, "hipHostMalloc", e);
    return P4S_OK;
}

int p4s_host_free(p4s_ctx *ctx, void *h_ptr) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipHostFree(h_ptr));
    return P4S_OK;
}

int p4s_copy_h2d(p4s_ctx *ctx, void *d_dst, const void *h_src, uint64_t bytes) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (bytes == 0) return P4S_OK;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    P4S_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return P4S_OK;
}

int p4s_copy_d2h(p4s_ctx *ctx, void *h_dst, const void *d_src, uint64_t bytes) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (bytes == 0) return P4S_OK;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    P4S_HIP(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return P4S_OK;
}

int p4s_memset(p4s_ctx *ctx, void *d_dst, int value, uint64_t bytes) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (bytes == 0) return P4S_OK;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    P4S_HIP(ctx, hipMemsetAsync(d_dst, value, bytes, ctx->stream));
    return P4S_OK;
}

int p4s_timer_start(p4s_ctx *ctx) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
    return P4S_OK;
}

int p4s_timer_stop(p4s_ctx *ctx, float *elapsed_ms) {
    if (!ctx || !elapsed_ms) return P4S_ERR_INVALID_ARG;
    P4S_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
    P4S_HIP(ctx, hipEventSynchronize(ctx->ev1));
    P4S_HIP(ctx, hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return P4S_OK;
}

static const char *const kCorrupt = "Invalid or corrupted parquet file: ";

int p4s_format_error(const p4s_page_status *st, char *buf, size_t buf_len) {
    if (!st || !buf || buf_len == 0) return 0;
    const unsigned long long a0 = st->aux0, a1 = st->aux1;
    int n = 0;
    switch (st->code) {
    case P4S_E_NONE: n = snprintf(buf, buf_len, "ok"); break;
    case P4S_E_DELTA_HEADER_EOF: n = snprintf(buf, buf_len, "Unexpected end of DELTA_BINARY_PACKED header"); break;
    case P4S_E_DELTA_ZERO_MINIBLOCKS: n = snprintf(buf, buf_len, "In DELTA_BINARY_PACKED number miniblocks per block is 0"); break;
    case P4S_E_DELTA_BLOCK_HEADER_EOF: n = snprintf(buf, buf_len, "Unexpected end of DELTA_BINARY_PACKED block header"); break;
    case P4S_E_DELTA_DATA_EOF: n = snprintf(buf, buf_len, "Unexpected end of data in DELTA_BINARY_PACKED"); break;
    case P4S_E_DELTA_BAD_WIDTH: n = snprintf(buf, buf_len, "DELTA_BINARY_PACKED miniblock bit width %llu exceeds 64", a0); break;
    case P4S_E_DELTA_BAD_SHAPE: n = snprintf(buf, buf_len, "DELTA_BINARY_PACKED block size %llu smaller than miniblock count %llu", a0, a1); break;
    case P4S_E_DLBA_VALUES_EOF: n = snprintf(buf, buf_len, "Unexpected end of values in DELTA_LENGTH_BYTE_ARRAY"); break;
    case P4S_E_DBA_BAD_PREFIX: n = snprintf(buf, buf_len, "Invalid prefix length in DELTA_BYTE_ARRAY"); break;
    case P4S_E_BSS_SIZE: n = snprintf(buf, buf_len, "Data size in BYTE_STREAM_SPLIT is not divisible by size of data type"); break;
    case P4S_E_DICT_BAD_WIDTH: n = snprintf(buf, buf_len, "%sIllegal dictionary index bit width (should be 0 <= bit width <= 32, got %llu)", kCorrupt, a0); break;
    case P4S_E_DICT_INDEX: n = snprintf(buf, buf_len, "%sDict index exceeds dict size (dict size = %llu, index = %llu)", kCorrupt, a0, a1); break;
    case P4S_E_PLAIN_BA_LEN_EOF: n = snprintf(buf, buf_len, "%sEnd of page while reading BYTE_ARRAY length (needed %lluB, got %lluB)", kCorrupt, a0, a1); break;
    case P4S_E_PLAIN_BA_EOF: n = snprintf(buf, buf_len, "%sEnd of page while reading BYTE_ARRAY (needed %lluB, got %lluB)", kCorrupt, a0, a1); break;
    case P4S_E_PLAIN_FLBA_EOF: n = snprintf(buf, buf_len, "%sEnd of page while reading FIXED_LEN_BYTE_ARRAY (needed %lluB, got %lluB)", kCorrupt, a0, a1); break;
    case P4S_E_LEVELS_EOF: n = snprintf(buf, buf_len, "%sEnd of page while reading levels (needed %lluB, got %lluB)", kCorrupt, a0, a1); break;
    case P4S_E_LEVELS_NEGATIVE: n = snprintf(buf, buf_len, "%sNegative RLE levels length (%lld)", kCorrupt, (long long)st->aux0); break;
    case P4S_E_LEVELS_BITPACKED_LEN: n = snprintf(buf, buf_len, "%sBIT_PACKED length exceeds int (%lluB)", kCorrupt, a0); break;
    case P4S_E_LEVELS_ENCODING: n = snprintf(buf, buf_len, "Unknown level encoding (%llu)", a0); break;
    case P4S_E_LEVELS_V2_LEN: n = snprintf(buf, buf_len, "%sLevels length exceeds int (%lluB)", kCorrupt, a0); break;
    case P4S_E_NO_DICT: n = snprintf(buf, buf_len, "%sNo dictionary page found before a dictionary-encoded page", kCorrupt); break;
    case P4S_E_DICT_EMPTY_PAGE: n = snprintf(buf, buf_len, "%sEmpty dictionary-encoded data page", kCorrupt); break;
    case P4S_E_OUT_CAPACITY: n = snprintf(buf, buf_len, "output slot too small (%llu bytes needed)", a0); break;
    case P4S_E_SNAPPY_LENGTH: n = snprintf(buf, buf_len, "%sCorrupt snappy data", kCorrupt); break;
    case P4S_E_SNAPPY_BUFFER: n = snprintf(buf, buf_len, "%sUncompression buffer size too small", kCorrupt); break;
    case P4S_E_SNAPPY_FAILED: n = snprintf(buf, buf_len, "Could not decompress snappy."); break;
    case P4S_E_THRIFT_EOF: n = snprintf(buf, buf_len, "Could not deserialize thrift: unexpected end of stream at %lluB", a0); break;
    case P4S_E_THRIFT_VARINT: n = snprintf(buf, buf_len, "Could not deserialize thrift: variable-length int over 10 bytes"); break;
    case P4S_E_THRIFT_DEPTH: n = snprintf(buf, buf_len, "Could not deserialize thrift: nesting too deep"); break;
    case P4S_E_THRIFT_WIRE_TYPE: n = snprintf(buf, buf_len, "Could not deserialize thrift: invalid wire type %llu", a0); break;
    case P4S_E_THRIFT_NOT_INT: n = snprintf(buf, buf_len, "Could not deserialize thrift: expected an integer field"); break;
    case P4S_E_THRIFT_INVALID_DATA: n = snprintf(buf, buf_len, "Could not deserialize thrift: TProtocolException: Invalid data"); break;
    case P4S_E_PAGE_NEGATIVE_SIZE: n = snprintf(buf, buf_len, "%sNegative compressed_page_size in header: %lld", kCorrupt, (long long)a0); break;
    case P4S_E_PAGE_EOF: n = snprintf(buf, buf_len, "%sUnexpected end of column chunk while reading compressed page contents (expected %lluB, got %lluB)", kCorrupt, a0, a1); break;
    case P4S_E_ENCODE_RANGE: n = snprintf(buf, buf_len, "value does not fit the bit width %llu", a0); break;
    default: n = snprintf(buf, buf_len, "page decode error %d", st->code); break;
    }
    return n < 0 ? 0 : n;
}

// ---------------------------------------------------------------------------
// decoders
// ---------------------------------------------------------------------------

// WGs per page along x for the kernels whose pages are split into tiles: the
// descriptors live on the device, so the host picks a fixed fan-out and the
// kernels grid-stride inside the page.
static unsigned fanout(uint32_t n_pages, unsigned want_total) {
    unsigned per = (want_total + n_pages - 1) / n_pages;
    return std::min(std::max(per, 1u), 64u);
}

// Launch grid of the streaming kernels that give every piece of a page a workgroup of its own: enough for a
// page of the batch's average size (the input allocation holds the pages).  A larger page takes several pieces per
// workgroup.
static dim3 piece_grid(uint32_t n_pages, uint64_t in_bytes, uint64_t piece_in_bytes) {
    const uint64_t avg = (in_bytes / n_pages + piece_in_bytes - 1) / piece_in_bytes + 1;
    const unsigned fan = (unsigned)std::min<uint64_t>(avg, 65536);
    // x = piece, (y, z) = page: the kernels find their page without a division
    const unsigned gy = std::min(n_pages, 32768u), gz = (n_pages + gy - 1) / gy;
    return dim3(fan, gy, gz);
}

int p4s_decode_byte_stream_split(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                 const p4s_page_desc *d_pages, uint32_t n_pages,
                                 void *d_out, p4s_page_status *d_status, int elem_size) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (elem_size != 4 && elem_size != 8) return fail(ctx, P4S_ERR_INVALID_ARG, "elem_size must be 4 or 8");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid = piece_grid(n_pages, in_bytes, (uint64_t)p4s::BSS_PIECE * elem_size);
    if (elem_size == 4)
        hipLaunchKernelGGL(p4s::k_byte_stream_split<4>, grid, dim3(p4s::BSS_THREADS), 0, ctx->stream,
                           d_in, in_bytes, d_pages, static_cast<uint8_t *>(d_out), d_status, n_pages);
    else
        hipLaunchKernelGGL(p4s::k_byte_stream_split<8>, grid, dim3(p4s::BSS_THREADS), 0, ctx->stream,
                           d_in, in_bytes, d_pages, static_cast<uint8_t *>(d_out), d_status, n_pages);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_decode_plain_fixed(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                           const p4s_page_desc *d_pages, uint32_t n_pages,
                           void *d_out, p4s_page_status *d_status, int elem_size) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (elem_size <= 0 || elem_size > 64) return fail(ctx, P4S_ERR_INVALID_ARG, "bad elem_size");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const dim3 grid = piece_grid(n_pages, in_bytes, p4s::COPY_PIECE);
    hipLaunchKernelGGL(p4s::k_plain_fixed, grid, dim3(p4s::COPY_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, static_cast<uint8_t *>(d_out), d_status, (uint32_t)elem_size, n_pages);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_decode_plain_boolean(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                             const p4s_page_desc *d_pages, uint32_t n_pages,
                             uint8_t *d_out, p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const unsigned fan = fanout(n_pages, 8192);
    dim3 grid(fan * n_pages);
    hipLaunchKernelGGL(p4s::k_plain_boolean, grid, dim3(p4s::COPY_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_out, d_status, fan);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_scatter_by_levels(p4s_ctx *ctx, const int32_t *d_def, const void *d_values,
                          const p4s_page_desc *d_pages, uint32_t n_pages, int elem_size,
                          void *d_out, uint32_t *d_validity, p4s_page_status *d_status) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_pages == 0) return P4S_OK;
    if (!d_pages || !d_status || !d_out || !d_validity || !d_values) return fail(ctx, P4S_ERR_INVALID_ARG, "null buffer");
    if (d_out == d_values) return fail(ctx, P4S_ERR_INVALID_ARG, "d_out must not alias d_values");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    dim3 grid(n_pages), block(p4s::SCATTER_THREADS);
    const uint8_t *v = static_cast<const uint8_t *>(d_values);
    uint8_t *o = static_cast<uint8_t *>(d_out);
    switch (elem_size) {
    case 1: hipLaunchKernelGGL(p4s::k_scatter_by_levels<1>, grid, block, 0, ctx->stream, d_def, v, d_pages, o, d_validity, d_status); break;
    case 4: hipLaunchKernelGGL(p4s::k_scatter_by_levels<4>, grid, block, 0, ctx->stream, d_def, v, d_pages, o, d_validity, d_status); break;
    case 8: hipLaunchKernelGGL(p4s::k_scatter_by_levels<8>, grid, block, 0, ctx->stream, d_def, v, d_pages, o, d_validity, d_status); break;
    case 12: hipLaunchKernelGGL(p4s::k_scatter_by_levels<12>, grid, block, 0, ctx->stream, d_def, v, d_pages, o, d_validity, d_status); break;
    default: return fail(ctx, P4S_ERR_INVALID_ARG, "elem_size must be 1, 4, 8 or 12");
    }
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_walk_page_headers(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_chunk_desc *d_chunks, uint32_t n_chunks,
                          p4s_page_info *d_pages, p4s_page_status *d_status) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_chunks == 0) return P4S_OK;
    if (!d_in || !d_chunks || !d_pages || !d_status) return fail(ctx, P4S_ERR_INVALID_ARG, "null argument");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_walk_page_headers, dim3(n_chunks), dim3(64), 0, ctx->stream,
                       d_in, in_bytes, d_chunks, d_pages, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

#ifdef P4S_DEBUG_BOUNDS
// Debug build only: the buffers the checked accesses may touch (up to six [lo, hi) ranges), and the outcome.
int p4s_debug_set_bounds(const uint64_t *lo, const uint64_t *hi) {
    p4s::DbgBounds b;
    memset(&b, 0, sizeof(b));
    for (int i = 0; i < 6; ++i) { b.lo[i] = lo[i]; b.hi[i] = hi[i]; }
    return hipMemcpyToSymbol(HIP_SYMBOL(p4s::g_dbg), &b, sizeof(b)) == hipSuccess ? 0 : -1;
}
// out[0] = accesses outside the ranges; out[1 + 4k ..] = address, site | size << 32, workgroup, thread of the
// first 16 of them (65 words)
int p4s_debug_get_hits(uint64_t *out) {
    p4s::DbgBounds b;
    if (hipMemcpyFromSymbol(&b, HIP_SYMBOL(p4s::g_dbg), sizeof(b)) != hipSuccess) return -1;
    out[0] = b.hits;
    for (int k = 0; k < 16; ++k)
        for (int j = 0; j < 4; ++j) out[1 + 4 * k + j] = b.rec[k][j];
    return 0;
}
#endif

}  // extern "C"
