// Copy-kernel variants on 2 GiB in + 2 GiB out: what a streaming copy reaches on this GPU (HIP events).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_one(const v4u *__restrict__ s, v4u *__restrict__ d, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_stride(const v4u *__restrict__ s, v4u *__restrict__ d, uint64_t n) {
    const uint64_t nthr = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * nthr < n; i += U * nthr) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * nthr) : s[i + u * nthr];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * nthr); else d[i + u * nthr] = v[u]; }
    }
    for (; i < n; i += nthr) d[i] = s[i];
}
// each workgroup owns a contiguous chunk (as the page kernels do)
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_chunk(const v4u *__restrict__ s, v4u *__restrict__ d, uint64_t n, uint64_t per) {
    const uint64_t lo = (uint64_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint64_t i = lo + threadIdx.x;
    for (; i + (U - 1) * 256 < hi; i += U * 256) {
        v4u v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
    }
    for (; i < hi; i += 256) d[i] = s[i];
}
// one short workgroup per 256 * U elements: every load issued before the first store
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_block(const v4u *__restrict__ s, v4u *__restrict__ d, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 * U + threadIdx.x;
    v4u v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) v[u] = NT ? __builtin_nontemporal_load(s + i + u * 256) : s[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i + u * 256 < n) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * 256); else d[i + u * 256] = v[u]; }
}
int main() {
    const uint64_t bytes = 2ull << 30, n = bytes / 16;
    v4u *s, *d;
    CK(hipMalloc(&s, bytes)); CK(hipMalloc(&d, bytes));
    CK(hipMemset(s, 1, bytes)); CK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char *name, auto launch) {
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
        printf("%-28s %.3f ms  %.0f GB/s\n", name, ms, 2.0 * bytes / ms / 1e6);
    };
    run("one v4u per thread", [&] { hipLaunchKernelGGL(k_one, dim3((n + 255) / 256), dim3(256), 0, 0, s, d, n); });
    run("block U2", [&] { hipLaunchKernelGGL((k_block<2, false>), dim3((n + 511) / 512), dim3(256), 0, 0, s, d, n); });
    run("block U4", [&] { hipLaunchKernelGGL((k_block<4, false>), dim3((n + 1023) / 1024), dim3(256), 0, 0, s, d, n); });
    run("block U8", [&] { hipLaunchKernelGGL((k_block<8, false>), dim3((n + 2047) / 2048), dim3(256), 0, 0, s, d, n); });
    run("block U16", [&] { hipLaunchKernelGGL((k_block<16, false>), dim3((n + 4095) / 4096), dim3(256), 0, 0, s, d, n); });
    run("block U1 nt", [&] { hipLaunchKernelGGL((k_block<1, true>), dim3((n + 255) / 256), dim3(256), 0, 0, s, d, n); });
    run("block U4 nt", [&] { hipLaunchKernelGGL((k_block<4, true>), dim3((n + 1023) / 1024), dim3(256), 0, 0, s, d, n); });
    run("block U8 nt", [&] { hipLaunchKernelGGL((k_block<8, true>), dim3((n + 2047) / 2048), dim3(256), 0, 0, s, d, n); });
    for (int g : {8192, 65536, 262144}) {
        char nm[64];
        snprintf(nm, 64, "stride U4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_stride<4, false>), dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, 64, "stride U8 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_stride<8, false>), dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, 64, "stride U4 nt grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_stride<4, true>), dim3(g), dim3(256), 0, 0, s, d, n); });
        snprintf(nm, 64, "chunk U4 grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_chunk<4, false>), dim3(g), dim3(256), 0, 0, s, d, n, (n + g - 1) / g); });
        snprintf(nm, 64, "chunk U8 nt grid %d", g); run(nm, [&] { hipLaunchKernelGGL((k_chunk<8, true>), dim3(g), dim3(256), 0, 0, s, d, n, (n + g - 1) / g); });
    }
    CK(hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0));
    run("hipMemcpyAsync D2D", [&] { (void)hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
