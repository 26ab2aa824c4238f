// Experiment: is a workgroup's LDS ever modified from outside?  Each workgroup fills `kb` KiB of LDS with a per-word pattern,
// then re-reads all of it `rounds` times (tens of microseconds) and counts words that changed; mismatches are accumulated in
// a global counter together with the first few (offset, value) pairs.  Run beside other kernels on another stream.
#include <hip/hip_runtime.h>
#include <stdint.h>

extern "C" __global__ __launch_bounds__(256) void canary_kernel(int words, int rounds, unsigned* out /*[0]=mismatches, [1..]=log*/) {
    extern __shared__ unsigned lds[];
    const unsigned tag = 0xC0DE0000u;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = tag ^ (unsigned)i;
    __syncthreads();
    for (int r = 0; r < rounds; ++r) {
        for (int i = threadIdx.x; i < words; i += blockDim.x) {
            const unsigned v = lds[i];
            if (v != (tag ^ (unsigned)i)) {
                const unsigned k = atomicAdd(&out[0], 1u);
                if (k < 64) { out[1 + 2 * k] = (unsigned)i; out[2 + 2 * k] = v; }
                lds[i] = tag ^ (unsigned)i;
            }
        }
        __syncthreads();
    }
}

extern "C" int canary_launch(int blocks, int kb, int rounds, unsigned* d_out, void* stream) {
    const int words = kb * 256;
    hipFuncSetAttribute((const void*)canary_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
    hipLaunchKernelGGL(canary_kernel, dim3(blocks), dim3(256), kb * 1024, (hipStream_t)stream, words, rounds, d_out);
    return (int)hipGetLastError();
}
