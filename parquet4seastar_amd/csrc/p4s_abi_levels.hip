// C ABI of the hybrid-RLE family: p4s_decode_levels and the hand-over of its counts to the value pages (contracts: include/p4s_gpu.h).
#include "p4s_abi_rle_launch.hh"

extern "C" {

int p4s_decode_levels(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                      const p4s_page_desc *d_pages, uint32_t n_pages,
                      void *d_out, p4s_page_status *d_status, int out_elem_size) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *out = static_cast<uint8_t *>(d_out);
    switch (out_elem_size) {
    case 2: return rle_launch<p4s::RLE_KIND_LEVELS, 2>(ctx, d_in, in_bytes, d_pages, n_pages, nullptr, 0, out, d_status);
    case 4: return rle_launch<p4s::RLE_KIND_LEVELS, 4>(ctx, d_in, in_bytes, d_pages, n_pages, nullptr, 0, out, d_status);
    default: return fail(ctx, P4S_ERR_INVALID_ARG, "level element size must be 2 or 4");
    }
}

int p4s_pages_set_num_values_from_levels(p4s_ctx *ctx, p4s_page_desc *d_value_pages,
                                         const p4s_page_status *d_level_status, uint32_t n_pages) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_pages == 0) return P4S_OK;
    if (!d_value_pages || !d_level_status) return fail(ctx, P4S_ERR_INVALID_ARG, "null array");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(p4s::k_set_num_values, dim3((n_pages + 255) / 256), dim3(256), 0, ctx->stream,
                       d_value_pages, d_level_status, n_pages);
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

}  // extern "C"
