// C ABI of the steps next to the decoders: Snappy page decompression and the DELTA_BINARY_PACKED encoder
// (contracts: include/p4s_gpu.h).
#include "p4s_abi_internal.hh"

using p4s_abi::check_common;
using p4s_abi::ensure_workspace;
using p4s_abi::fail;

#include "k_snappy.cuh"
#include "k_encode.cuh"
#include "k_rle_encode.cuh"

extern "C" {

int p4s_decompress_snappy(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_page_desc *d_pages, uint32_t n_pages,
                          uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_out && out_bytes) return fail(ctx, P4S_ERR_INVALID_ARG, "null output buffer");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_snappy_decompress, dim3(n_pages), dim3(p4s::SNAPPY_THREADS), 0, ctx->stream,
                       d_in, in_bytes, d_pages, d_out, out_bytes, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_encode_delta_binary_packed(p4s_ctx *ctx, const void *d_values, uint64_t n_values,
                                   const p4s_page_desc *d_pages, uint32_t n_pages,
                                   uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status, int elem_size) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_pages == 0) return P4S_OK;
    if (!d_pages || !d_status || !d_out) return fail(ctx, P4S_ERR_INVALID_ARG, "null argument");
    if (!d_values && n_values) return fail(ctx, P4S_ERR_INVALID_ARG, "null values");
    if (elem_size != 4 && elem_size != 8) return fail(ctx, P4S_ERR_INVALID_ARG, "elem_size must be 4 or 8");
    if (reinterpret_cast<uintptr_t>(d_values) & (uintptr_t)(elem_size - 1)) return fail(ctx, P4S_ERR_INVALID_ARG, "d_values must be aligned to its element size");
    if (n_pages > 65535u * 1024u) return fail(ctx, P4S_ERR_INVALID_ARG, "too many pages in one batch");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    if (elem_size == 4)
        hipLaunchKernelGGL(p4s::k_delta_encode<uint32_t>, dim3(n_pages), dim3(p4s::ENC_THREADS), 0, ctx->stream,
                           static_cast<const uint32_t *>(d_values), n_values, d_pages, d_out, out_bytes, d_status);
    else
        hipLaunchKernelGGL(p4s::k_delta_encode<uint64_t>, dim3(n_pages), dim3(p4s::ENC_THREADS), 0, ctx->stream,
                           static_cast<const uint64_t *>(d_values), n_values, d_pages, d_out, out_bytes, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

int p4s_encode_rle(p4s_ctx *ctx, const uint32_t *d_values, uint64_t n_values,
                   const p4s_page_desc *d_pages, uint32_t n_pages,
                   uint8_t *d_out, uint64_t out_bytes, p4s_page_status *d_status) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_pages == 0) return P4S_OK;
    if (!d_pages || !d_status || !d_out) return fail(ctx, P4S_ERR_INVALID_ARG, "null argument");
    if (!d_values && n_values) return fail(ctx, P4S_ERR_INVALID_ARG, "null values");
    if (reinterpret_cast<uintptr_t>(d_values) & 3u) return fail(ctx, P4S_ERR_INVALID_ARG, "d_values must be 4-byte aligned");
    if (n_pages > 65535u * 1024u) return fail(ctx, P4S_ERR_INVALID_ARG, "too many pages in one batch");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_rle_encode, dim3(n_pages), dim3(p4s::RLEENC_THREADS), 0, ctx->stream,
                       d_values, n_values, d_pages, d_out, out_bytes, d_status);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

}  // extern "C"
