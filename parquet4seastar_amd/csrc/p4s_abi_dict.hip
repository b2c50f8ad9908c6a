// C ABI of the hybrid-RLE family: p4s_decode_dictionary (contracts: include/p4s_gpu.h).
#include "p4s_abi_rle_launch.hh"

extern "C" {

int p4s_decode_dictionary(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                          const p4s_page_desc *d_pages, uint32_t n_pages,
                          const void *d_dict, uint64_t dict_size, int elem_size,
                          void *d_out, p4s_page_status *d_status) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (!d_dict) return fail(ctx, P4S_ERR_INVALID_ARG, "null dictionary");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    const uint8_t *dict = static_cast<const uint8_t *>(d_dict);
    uint8_t *out = static_cast<uint8_t *>(d_out);
    switch (elem_size) {
    case 1: return rle_launch<p4s::RLE_KIND_DICT, 1>(ctx, d_in, in_bytes, d_pages, n_pages, dict, dict_size, out, d_status);
    case 4: return rle_launch<p4s::RLE_KIND_DICT, 4>(ctx, d_in, in_bytes, d_pages, n_pages, dict, dict_size, out, d_status);
    case 8: return rle_launch<p4s::RLE_KIND_DICT, 8>(ctx, d_in, in_bytes, d_pages, n_pages, dict, dict_size, out, d_status);
    case 12: return rle_launch<p4s::RLE_KIND_DICT, 12>(ctx, d_in, in_bytes, d_pages, n_pages, dict, dict_size, out, d_status);
    default: return fail(ctx, P4S_ERR_INVALID_ARG, "dictionary element size must be 1, 4, 8 or 12");
    }
}

}  // extern "C"
