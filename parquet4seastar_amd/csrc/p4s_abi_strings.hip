// C ABI of the BYTE_ARRAY decoders: DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY (with its sizes pass and its variant
// that allocates nothing), PLAIN BYTE_ARRAY and FIXED_LEN_BYTE_ARRAY (contracts: include/p4s_gpu.h).
#include "p4s_abi_internal.hh"

using p4s_abi::check_common;
using p4s_abi::ensure_workspace;
using p4s_abi::fail;

#include "k_strings.cuh"
#include "k_dba.cuh"

// Scratch of p4s_decode_delta_byte_array_static, the entry point for callers that cannot allocate: part of the
// code object, so every GPU the library is loaded on holds its own copy until the process ends.  In host code this
// name is only a handle: the device address is looked up once per device (p4s_ctx_create calls
// static_scratch_address below with that device current) and kept in the context.
static __device__ __align__(256) unsigned char scratch[P4S_STATIC_SCRATCH_BYTES];

namespace p4s_abi {

int static_scratch_address(void **out) {
    void *sp = nullptr;
    if (hipGetSymbolAddress(&sp, HIP_SYMBOL(scratch)) != hipSuccess || !sp) return P4S_ERR_HIP;
    *out = sp;
    return P4S_OK;
}

}  // namespace p4s_abi

extern "C" {

int p4s_decode_delta_length_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                       const p4s_page_desc *d_pages, uint32_t n_pages,
                                       int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                       p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_offsets || (!d_bytes && bytes_cap)) return fail(ctx, P4S_ERR_INVALID_ARG, "null output buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_delta_length_byte_array, dim3(n_pages), dim3(p4s::DELTA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_offsets, d_bytes, bytes_cap, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

// `side` = room for the tables the DELTA_BYTE_ARRAY kernels hand to each other: one DbaPageMeta and one DbaStreams per
// page, then one uint32 per 256 scratch slots (+ 1).
static uint64_t dba_side_bytes(uint32_t n_pages, uint64_t slots) {
    return (sizeof(p4s::DbaPageMeta) + sizeof(p4s::DbaStreams)) * (uint64_t)n_pages + 4ull * (slots / p4s::DBA_TILE + 1);
}
static_assert(sizeof(p4s::DbaPageMeta) + sizeof(p4s::DbaStreams) == P4S_STATIC_SCRATCH_PAGE_BYTES, "side table layout");

struct DbaSide {
    p4s::DbaAux aux;
    p4s::DbaStreams *recs;
};
static DbaSide dba_side(void *side, uint32_t n_pages) {
    uint8_t *p = static_cast<uint8_t *>(side);
    DbaSide s;
    s.aux.meta = reinterpret_cast<p4s::DbaPageMeta *>(p);
    s.recs = reinterpret_cast<p4s::DbaStreams *>(p + sizeof(p4s::DbaPageMeta) * (uint64_t)n_pages);
    s.aux.tile_min = reinterpret_cast<uint32_t *>(p + (sizeof(p4s::DbaPageMeta) + sizeof(p4s::DbaStreams)) * (uint64_t)n_pages);
    return s;
}

int p4s_delta_byte_array_sizes(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                               const p4s_page_desc *d_pages, uint32_t n_pages,
                               void *d_scratch, uint64_t scratch_bytes, p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_scratch) return fail(ctx, P4S_ERR_INVALID_ARG, "null scratch buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t slots = scratch_bytes / 8;
    int rc = ensure_workspace(ctx, P4S_WS_DBA, dba_side_bytes(n_pages, slots));
    if (rc != P4S_OK) return rc;
    const DbaSide sd = dba_side(ctx->ws[P4S_WS_DBA].ptr, n_pages);
    hipLaunchKernelGGL(p4s::k_dba_streams, dim3(2 * n_pages), dim3(p4s::DELTA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, static_cast<uint32_t *>(d_scratch), slots, sd.recs);
    hipLaunchKernelGGL(p4s::k_dba_scan<true>, dim3(n_pages), dim3(p4s::DELTA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, static_cast<int64_t *>(nullptr), (uint64_t)0,
                       static_cast<uint32_t *>(d_scratch), slots, sd.recs, sd.aux, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

// The three launches of a DELTA_BYTE_ARRAY decode.
static int dba_launch(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes, const p4s_page_desc *d_pages, uint32_t n_pages,
                      int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap, uint32_t *d_scratch, uint64_t slots,
                      void *side, p4s_page_status *d_status) {
    const DbaSide sd = dba_side(side, n_pages);
    const p4s::DbaAux aux = sd.aux;
    hipLaunchKernelGGL(p4s::k_dba_streams, dim3(2 * n_pages), dim3(p4s::DELTA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_scratch, slots, sd.recs);
    hipLaunchKernelGGL(p4s::k_dba_scan<false>, dim3(n_pages), dim3(p4s::DELTA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_offsets, bytes_cap, d_scratch, slots, sd.recs, aux, d_status);
    // the descriptors live on the device: a fixed number of single-wavefront workgroups per page, each with a run of
    // the page's strings (P4S_DBA_FAN overrides the number of workgroups the batch is spread over)
    static const uint32_t want = [] { const char *e = getenv("P4S_DBA_FAN"); long x = e ? atol(e) : 32768; return (uint32_t)(x < 1 ? 1 : x); }();
    const uint32_t fan = std::min(std::max(want / n_pages, 1u), 64u);
    hipLaunchKernelGGL(p4s::k_dba_assemble, dim3(n_pages * fan), dim3(64), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_offsets, d_bytes, d_scratch, slots, aux, d_status, fan);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_decode_delta_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                const p4s_page_desc *d_pages, uint32_t n_pages,
                                int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                void *d_scratch, uint64_t scratch_bytes, p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_offsets || (!d_bytes && bytes_cap) || !d_scratch) return fail(ctx, P4S_ERR_INVALID_ARG, "null buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const uint64_t slots = scratch_bytes / 8;
    int rc = ensure_workspace(ctx, P4S_WS_DBA, dba_side_bytes(n_pages, slots));
    if (rc != P4S_OK) return rc;
    return dba_launch(ctx, d_in, in_bytes, d_pages, n_pages, d_offsets, d_bytes, bytes_cap,
                      static_cast<uint32_t *>(d_scratch), slots, ctx->ws[P4S_WS_DBA].ptr, d_status);
}

int p4s_decode_delta_byte_array_static(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                       const p4s_page_desc *d_pages, uint32_t n_pages,
                                       int64_t *d_offsets, uint64_t n_offset_slots, uint8_t *d_bytes, uint64_t bytes_cap,
                                       p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_offsets || (!d_bytes && bytes_cap)) return fail(ctx, P4S_ERR_INVALID_ARG, "null buffer");
    if (n_offset_slots > P4S_STATIC_SCRATCH_SLOTS)
        return fail(ctx, P4S_ERR_INVALID_ARG,
                    "p4s_decode_delta_byte_array_static: the batch spans more offsets slots than the static scratch holds "
                    "(P4S_STATIC_SCRATCH_SLOTS); split the batch or use p4s_decode_delta_byte_array");
    if (n_pages > P4S_STATIC_SCRATCH_PAGES)
        return fail(ctx, P4S_ERR_INVALID_ARG,
                    "p4s_decode_delta_byte_array_static: more pages than the static scratch holds records for "
                    "(P4S_STATIC_SCRATCH_PAGES); split the batch or use p4s_decode_delta_byte_array");
    if (!ctx->static_scratch) return fail(ctx, P4S_ERR_HIP, "static scratch address not resolved");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    // the kernels take the device address kept in the context, never the array's host-side name; a page whose slots
    // (out_offset .. out_offset + values) lie past n_offset_slots reports P4S_E_OUT_CAPACITY.  The array holds the
    // per-string scratch first, the side tables of dba_launch behind it.
    uint8_t *base = static_cast<uint8_t *>(ctx->static_scratch);
    static_assert(P4S_STATIC_SCRATCH_BYTES >= 8ull * P4S_STATIC_SCRATCH_SLOTS + (uint64_t)P4S_STATIC_SCRATCH_PAGE_BYTES * P4S_STATIC_SCRATCH_PAGES +
                                                  4ull * (P4S_STATIC_SCRATCH_SLOTS / p4s::DBA_TILE + 1),
                  "static scratch too small for its parts");
    return dba_launch(ctx, d_in, in_bytes, d_pages, n_pages, d_offsets, d_bytes, bytes_cap,
                      reinterpret_cast<uint32_t *>(base), n_offset_slots, base + 8ull * P4S_STATIC_SCRATCH_SLOTS, d_status);
}

int p4s_decode_plain_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                const p4s_page_desc *d_pages, uint32_t n_pages,
                                int64_t *d_offsets, uint8_t *d_bytes, uint64_t bytes_cap,
                                p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_offsets || (!d_bytes && bytes_cap)) return fail(ctx, P4S_ERR_INVALID_ARG, "null output buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_plain_byte_array, dim3(n_pages), dim3(p4s::PBA_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_offsets, d_bytes, bytes_cap, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_decode_plain_fixed_len_byte_array(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                          const p4s_page_desc *d_pages, uint32_t n_pages,
                                          uint8_t *d_bytes, uint64_t bytes_cap,
                                          p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_bytes && bytes_cap) return fail(ctx, P4S_ERR_INVALID_ARG, "null output buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_plain_flba, dim3(n_pages), dim3(256), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_bytes, bytes_cap, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

}  // extern "C"
