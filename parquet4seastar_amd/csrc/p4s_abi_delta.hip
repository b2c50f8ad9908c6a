// C ABI of the DELTA_BINARY_PACKED decoder (contract: include/p4s_gpu.h).
#include "p4s_abi_internal.hh"

using p4s_abi::check_common;
using p4s_abi::ensure_workspace;
using p4s_abi::fail;

#include "k_delta_split.cuh"

namespace {

// DELTA_BINARY_PACKED: batches of up to P4S_DELTA_SPLIT pages (default 64: the split's first passes follow a page's blocks serially,
// about 0.06 ms whatever the batch, and its last pass adds the parts' bases to values already written, so it pays only while most
// of the chip would otherwise idle; 0: never) are decoded with their pages split over several workgroups (k_delta_split.cuh).
uint32_t delta_split_max_pages() {
    static const uint32_t v = [] {
        const char *e = getenv("P4S_DELTA_SPLIT");
        long x = e ? atol(e) : 64;
        if (x < 0) x = 0;
        if (x > 65536) x = 65536;
        return (uint32_t)x;
    }();
    return v;
}

template <typename T>
int delta_launch(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes, const p4s_page_desc *d_pages, uint32_t n_pages,
                 T *d_out, p4s_page_status *d_status) {
    using namespace p4s;
    const dim3 block(DELTA_THREADS);
    hipStream_t st = ctx->stream;
    if (n_pages > 0 && n_pages <= delta_split_max_pages()) {
        const uint32_t fan = std::min(std::max(2048u / n_pages, 2u), 16u);
        const uint64_t o_parts = (sizeof(DeltaSplitPage) * (uint64_t)n_pages + 255) & ~255ull;
        const uint64_t bytes = o_parts + sizeof(DeltaPartRec) * (uint64_t)n_pages * fan;
        int rc = ensure_workspace(ctx, P4S_WS_DELTA, bytes);
        if (rc != P4S_OK) return rc;
        uint8_t *sc = static_cast<uint8_t *>(ctx->ws[P4S_WS_DELTA].ptr);
        DeltaSplitPage *split = reinterpret_cast<DeltaSplitPage *>(sc);
        DeltaPartRec *parts = reinterpret_cast<DeltaPartRec *>(sc + o_parts);
        hipLaunchKernelGGL(k_delta_split_guess<T>, dim3(n_pages), dim3(64), 0, st, d_in, in_bytes, d_pages, parts, split, fan);
        hipLaunchKernelGGL(k_delta_split_walk<T>, dim3(n_pages), block, 0, st, d_in, in_bytes, d_pages, parts, split, fan);
        hipLaunchKernelGGL(k_delta_binary_packed<T>, dim3(n_pages * fan), block, 0, st, d_in, in_bytes, d_pages, d_out, d_status,
                           parts, split, fan, 0u);
        hipLaunchKernelGGL(k_delta_split_finish<T>, dim3(n_pages * fan, 4), dim3(256), 0, st, d_pages, d_out, d_status, parts, split, fan);
        hipLaunchKernelGGL(k_delta_binary_packed<T>, dim3(n_pages), block, 0, st, d_in, in_bytes, d_pages, d_out, d_status,
                           parts, split, fan, 1u);
    } else {
        hipLaunchKernelGGL(k_delta_binary_packed<T>, dim3(n_pages), block, 0, st, d_in, in_bytes, d_pages, d_out, d_status,
                           static_cast<DeltaPartRec *>(nullptr), static_cast<DeltaSplitPage *>(nullptr), 1u, 0u);
    }
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

}  // namespace

extern "C" {

int p4s_decode_delta_binary_packed(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes,
                                   const p4s_page_desc *d_pages, uint32_t n_pages,
                                   void *d_out, p4s_page_status *d_status, int elem_size) {
    int c = check_common(ctx, d_in, d_pages, n_pages, d_status);
    if (c != 1) return c;
    if (elem_size != 4 && elem_size != 8) return fail(ctx, P4S_ERR_INVALID_ARG, "elem_size must be 4 or 8");
    P4S_HIP(ctx, hipSetDevice(ctx->device));
    if (elem_size == 4) return delta_launch<uint32_t>(ctx, d_in, in_bytes, d_pages, n_pages, static_cast<uint32_t *>(d_out), d_status);
    return delta_launch<uint64_t>(ctx, d_in, in_bytes, d_pages, n_pages, static_cast<uint64_t *>(d_out), d_status);
}

}  // extern "C"
