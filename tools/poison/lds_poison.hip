// Test helper: fills the LDS of every CU (and a slab of per-lane private memory) with a word, so that a kernel
// launched afterwards that reads LDS or private memory it has not written sees that word instead of whatever
// the previous kernel left.  Not part of the product library.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void k_poison(uint32_t word, uint32_t words, uint32_t *sink) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < words; i += blockDim.x) lds[i] = word;
    volatile uint32_t priv[64];
    for (int i = 0; i < 64; ++i) priv[i] = word;
    __syncthreads();
    uint32_t acc = 0;
    for (int i = 0; i < 64; ++i) acc += priv[i];
    if (lds[(threadIdx.x * 7u) % words] + acc == 0x12345u) sink[0] = 1;     // keeps the stores alive
}

extern "C" int lds_poison(uint32_t word) {
    uint32_t *sink = nullptr;
    if (hipMalloc(&sink, 4) != hipSuccess) return 1;
    const size_t sizes[3] = {65536, 32768, 16384};
    for (int r = 0; r < 3; ++r) {
        if (hipFuncSetAttribute((const void *)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) return 2;
        hipLaunchKernelGGL(k_poison, dim3(256 * 16), dim3(256), sizes[r], 0, word, (uint32_t)(sizes[r] / 4), sink);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipFree(sink);
    return 0;
}
