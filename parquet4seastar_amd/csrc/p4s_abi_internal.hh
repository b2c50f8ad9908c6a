// What the translation units of libp4s_gpu.so share on the host side: the context, error plumbing, argument
// checks and the context-owned workspaces.  Each decoder family is compiled on its own (p4s_abi_*.hip) so that
// the library builds in parallel; p4s_abi.hip holds the context, memory, timer and error-text entry points.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <string>

#include "../../include/p4s_gpu.h"

// A workspace the library owns on behalf of one kernel family: side tables that the kernels of one call hand to
// each other.  Every family has its own (a call of one family never overwrites tables that queued launches of
// another family still read), and each remembers the stream it was last used on.
struct p4s_workspace {
    void *ptr = nullptr;
    uint64_t bytes = 0;
    hipStream_t last_stream = nullptr;
    bool used = false;
};

enum p4s_workspace_id : int { P4S_WS_RLE = 0, P4S_WS_DELTA = 1, P4S_WS_DBA = 2, P4S_WS_COUNT = 3 };

struct p4s_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string last_error;
    // workspaces owned by the context, grown on demand (never inside a steady-state loop: the second call with
    // the same shapes allocates nothing)
    p4s_workspace ws[P4S_WS_COUNT];
    // device address of the static scratch array of p4s_decode_delta_byte_array_static on this context's GPU
    void *static_scratch = nullptr;
};

namespace p4s_abi {

inline int fail(p4s_ctx *ctx, int code, const char *what, hipError_t e = hipSuccess) {
    if (ctx) {
        ctx->last_error = what;
        if (e != hipSuccess) {
            ctx->last_error += ": ";
            ctx->last_error += hipGetErrorString(e);
        }
    }
    return code;
}

#define P4S_HIP(ctx, call)                                                            \
    do {                                                                              \
        hipError_t e_ = (call);                                                       \
        if (e_ != hipSuccess) return ::p4s_abi::fail((ctx), P4S_ERR_HIP, #call, e_);  \
    } while (0)

inline int check_common(p4s_ctx *ctx, const void *d_in, const void *d_pages, uint32_t n_pages, const void *d_status) {
    if (!ctx) return P4S_ERR_INVALID_ARG;
    if (n_pages == 0) return P4S_OK;
    if (!d_pages || !d_status) return fail(ctx, P4S_ERR_INVALID_ARG, "null page or status array");
    if (!d_in) return fail(ctx, P4S_ERR_INVALID_ARG, "null input buffer");
    if (reinterpret_cast<uintptr_t>(d_in) & 15u) return fail(ctx, P4S_ERR_INVALID_ARG, "d_in must be 16-byte aligned");
    if (n_pages > 65535u * 1024u) return fail(ctx, P4S_ERR_INVALID_ARG, "too many pages in one batch");
    return 1;
}

// Room for `bytes` in workspace `id`, usable by launches on the context's current stream.  Growing it waits for
// the stream that used it last and reallocates (the only place a decode call may block; a call that fits the
// workspace it already has never does).  Using it from another stream than the last one first waits for that stream.
// (p4s_abi.hip)
int ensure_workspace(p4s_ctx *ctx, int id, uint64_t bytes);

// Device address of the static scratch array on the current device (p4s_abi_strings.hip, next to the array).
int static_scratch_address(void **out);

}  // namespace p4s_abi
