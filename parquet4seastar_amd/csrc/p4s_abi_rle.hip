// C ABI of the hybrid-RLE family: p4s_decode_rle (contracts: include/p4s_gpu.h).
#include "p4s_abi_rle_launch.hh"

extern "C" {

int p4s_decode_rle(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                   const p4s_page_desc *d_pages, uint32_t n_pages,
                   void *d_out, p4s_page_status *d_status, int out_elem_size) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    uint8_t *out = static_cast<uint8_t *>(d_out);
    switch (out_elem_size) {
    case 1: return rle_launch<p4s::RLE_KIND_RAW, 1>(ctx, d_in, in_bytes, d_pages, n_pages, nullptr, 0, out, d_status);
    case 2: return rle_launch<p4s::RLE_KIND_RAW, 2>(ctx, d_in, in_bytes, d_pages, n_pages, nullptr, 0, out, d_status);
    case 4: return rle_launch<p4s::RLE_KIND_RAW, 4>(ctx, d_in, in_bytes, d_pages, n_pages, nullptr, 0, out, d_status);
    default: return fail(ctx, P4S_ERR_INVALID_ARG, "out_elem_size must be 1, 2 or 4");
    }
}

}  // extern "C"
