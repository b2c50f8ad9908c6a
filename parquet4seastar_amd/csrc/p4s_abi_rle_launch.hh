// Launch sequence of the hybrid-RLE family, shared by its translation units (one per kind of stream: raw,
// levels, dictionary indices, so that they compile side by side).
#pragma once
#include "p4s_abi_internal.hh"
#include "k_rle_split.cuh"

using p4s_abi::check_common;
using p4s_abi::ensure_workspace;
using p4s_abi::fail;

namespace {

// Whether a batch is decoded with its pages split over several workgroups (k_rle_split.cuh).  P4S_RLE_SPLIT=0
// never splits, P4S_RLE_SPLIT=<n> splits batches of up to n pages (default: up to 512 pages).
uint32_t rle_split_max_pages() {
    static const uint32_t v = [] {
        const char *e = getenv("P4S_RLE_SPLIT");
        long x = e ? atol(e) : 512;
        if (x < 0) x = 0;
        if (x > (long)p4s::RLE_SPLIT_MAX_PAGES) x = p4s::RLE_SPLIT_MAX_PAGES;
        return (uint32_t)x;
    }();
    return v;
}

template <int KIND, int ELEM>
int rle_launch(p4s_ctx *ctx, const uint8_t *d_in, uint64_t in_bytes, const p4s_page_desc *d_pages, uint32_t n_pages,
               const uint8_t *d_dict, uint64_t dict_size, uint8_t *d_out, p4s_page_status *d_status) {
    using namespace p4s;
    const dim3 block(RLE_THREADS);
    // windows the batch can span: every page's stream rounded up to whole windows (pages that overlap each other do
    // not fit and are decoded whole); batches of more than 1 GiB are decoded one workgroup per page
    const uint64_t cap64 = in_bytes / RLE_WIN + 2ull * n_pages + 8;
    // (dictionary pages: what they cost are the lookups, which the split does not make cheaper; its passes over 256
    // pages of 2 MiB cost more than the idle workgroups they fill: 2.1 ms against 1.5)
    const uint64_t cap_max = KIND == RLE_KIND_DICT ? 65536 : 262144;
    if (n_pages <= rle_split_max_pages() && n_pages > 0 && cap64 <= cap_max) {
        const uint32_t cap = (uint32_t)cap64;
        // parts per page: as many workgroups in all as the chip holds at once (2048; 4096 measured 4-10 % slower: a part that small is mostly start-up; P4S_RLE_SPLIT_WORKGROUPS overrides the total)
        static const uint32_t want = [] { const char *e = getenv("P4S_RLE_SPLIT_WORKGROUPS"); long x = e ? atol(e) : 2048; return (uint32_t)(x < 2 ? 2 : x); }();
        const uint32_t fan = std::min(std::max(want / n_pages, 2u), 64u);
        auto up = [](uint64_t x) { return (x + 255) & ~255ull; };
        uint64_t off = 0;
        const uint64_t o_pages = off; off += up(sizeof(RleSplitPage) * n_pages);
        const uint64_t o_total = off; off += 256;
        const uint64_t o_winpage = off; off += up(4ull * cap);
        const uint64_t o_entry = off; off += up(4ull * cap);
        const uint64_t o_wflag = off; off += up(4ull * cap);
        const uint64_t o_wcount = off; off += up(8ull * cap);
        const uint64_t o_wbase = off; off += up(8ull * cap);
        const uint64_t o_exits = off; off += up(2ull * RLE_WIN * cap);
        const uint64_t o_ckpos = off; off += up(2ull * RLE_CKS * cap);
        const uint64_t o_ckcnt = off; off += up(4ull * RLE_CKS * cap);
        const uint64_t o_parts = off; off += up(sizeof(RlePart) * (uint64_t)n_pages * fan);
        if (off > P4S_WORKSPACE_MAX_BYTES) {      // too large a batch to keep its exit tables: one workgroup per page
            hipLaunchKernelGGL((k_rle<KIND, ELEM>), dim3(n_pages), block, 0, ctx->stream, d_in, in_bytes, d_pages, d_dict, dict_size,
                               d_out, d_status, static_cast<const RlePart *>(nullptr), static_cast<RleSplitPage *>(nullptr), 1u);
            P4S_HIP(ctx, hipGetLastError());
            return P4S_OK;
        }
        int rc = ensure_workspace(ctx, P4S_WS_RLE, off);
        if (rc != P4S_OK) return rc;
        uint8_t *sc = static_cast<uint8_t *>(ctx->ws[P4S_WS_RLE].ptr);
        RleSplitArrays a;
        a.pages = reinterpret_cast<RleSplitPage *>(sc + o_pages);
        a.total_windows = reinterpret_cast<uint32_t *>(sc + o_total);
        a.win_page = reinterpret_cast<uint32_t *>(sc + o_winpage);
        a.entry = reinterpret_cast<uint32_t *>(sc + o_entry);
        a.wflag = reinterpret_cast<uint32_t *>(sc + o_wflag);
        a.wcount = reinterpret_cast<uint64_t *>(sc + o_wcount);
        a.wbase = reinterpret_cast<uint64_t *>(sc + o_wbase);
        a.exits = reinterpret_cast<uint16_t *>(sc + o_exits);
        a.ck_pos = reinterpret_cast<uint16_t *>(sc + o_ckpos);
        a.ck_cnt = reinterpret_cast<uint32_t *>(sc + o_ckcnt);
        a.parts = reinterpret_cast<RlePart *>(sc + o_parts);
        a.cap = cap;
        hipStream_t st = ctx->stream;
        hipLaunchKernelGGL((k_rle_split_init<KIND>), dim3(n_pages), block, 0, st, d_in, in_bytes, d_pages, n_pages, a);
        hipLaunchKernelGGL((k_rle_split_exits<KIND, ELEM>), dim3(cap), dim3(RLE_EXIT_THREADS), 0, st, d_in, in_bytes, d_pages, a);
        hipLaunchKernelGGL((k_rle_split_entry<KIND>), dim3(n_pages), dim3(64), 0, st, d_in, in_bytes, d_pages, n_pages, a);
        // a wave per window while the windows are few; a lane per window when they fill the chip anyway, except, up to
        // a point, for pages whose runs are short (narrow values)
        if (cap > 2048)
            hipLaunchKernelGGL((k_rle_split_count<KIND, ELEM>), dim3((cap + 255) / 256), dim3(256), 0, st, d_in, in_bytes, d_pages, a,
                               cap <= 16384 ? 1u : 0u);
        if (cap <= 16384)
            hipLaunchKernelGGL((k_rle_split_count_wave<KIND, ELEM>), dim3(cap), dim3(RLE_EXIT_THREADS), 0, st, d_in, in_bytes, d_pages, a,
                               cap > 2048 ? 1u : 0u);
        hipLaunchKernelGGL((k_rle_split_plan<KIND>), dim3(n_pages), block, 0, st, d_in, in_bytes, d_pages, a, fan);
        hipLaunchKernelGGL((k_rle<KIND, ELEM>), dim3(n_pages * fan), block, 0, st, d_in, in_bytes, d_pages, d_dict, dict_size,
                           d_out, d_status, a.parts, a.pages, fan);
        hipLaunchKernelGGL((k_rle_split_finish<KIND>), dim3((n_pages + 255) / 256), dim3(256), 0, st, d_in, in_bytes, d_pages,
                           n_pages, dict_size, a, d_status);
    } else {
        hipLaunchKernelGGL((k_rle<KIND, ELEM>), dim3(n_pages), block, 0, ctx->stream, d_in, in_bytes, d_pages, d_dict, dict_size,
                           d_out, d_status, static_cast<const RlePart *>(nullptr), static_cast<RleSplitPage *>(nullptr), 1u);
    }
    P4S_HIP(ctx, hipGetLastError());
    return P4S_OK;
}

}  // namespace

