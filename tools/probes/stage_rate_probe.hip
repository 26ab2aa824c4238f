// GPU diagnostic: how fast can ONE workgroup stage bytes from L2 into an LDS ring, and does the path matter?
// The convolution's main loop looked bound by staging at ~27-30 GB/s per workgroup whatever the tile shape
// (profiles/r02_conv_timeline.log: 8 / 12 / 16 KB chunks take 0.29 / 0.41 / 0.59 us).  This probe runs the loop's loader alone -
// a ring of 4 slots, 3 chunks in flight, one counted vmcnt wait + one s_barrier per chunk, no consumer - in two forms:
//   mode 0  `buffer_load_dwordx4 ... lds` (LDS-DMA, what the kernel does)
//   mode 1  `global_load_dwordx4` into registers + `ds_write_b128`
// for 1, 2, 4 or 8 loader waves, chunk sizes of 8 / 16 KB, and 1 or 2 workgroups per CU (grid 256 / 512).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stage_rate_probe.hip -o tools/probes/stage_rate_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lds_ptr;

constexpr int kRowBytes = 4096, kRows = 4096;      // 16 MB source: rows of 4 KB, a chunk takes 128 B of each of its rows

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// CHUNK_KB per iteration over WAVES loader waves: INST = CHUNK_KB / WAVES wave-instructions of 1 KB each per wave and chunk
template <int MODE, int WAVES, int CHUNK_KB>
__global__ __launch_bounds__(64 * WAVES) void stage_kernel(const char* __restrict__ src, int iters, float* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the host pass has no buffer-resource type)
    constexpr int INST = CHUNK_KB / WAVES, NBUF = 4, D = NBUF - 1;
    static_assert(INST >= 1 && INST * D <= 63, "vmcnt");
    __shared__ __attribute__((aligned(16))) char smem[NBUF * CHUNK_KB * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int row_in_inst = lane >> 3, seg = lane & 7;      // 8 rows x 128 B per wave-instruction
    // byte offset of this lane's 16 bytes in chunk 0, per instruction
    int off[INST];
#pragma unroll
    for (int i = 0; i < INST; ++i) {
        const int row = ((int)blockIdx.x * (CHUNK_KB * 8) + (wave * INST + i) * 8 + row_in_inst) % kRows;
        off[i] = row * kRowBytes + seg * 16;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, kRows * kRowBytes, 0x00020000);
    v4f regs[D][INST];
    auto issue = [&](const int c, const int slot, v4f (&r)[INST]) {
        const int koff = (c % (kRowBytes / 128)) * 128;
#pragma unroll
        for (int i = 0; i < INST; ++i) {
            if (MODE == 0)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + slot * CHUNK_KB * 1024 + (wave * INST + i) * 1024), 16, off[i] + koff, 0, 0, 0);
            else
                r[i] = *reinterpret_cast<const v4f*>(src + off[i] + koff);
        }
    };
    auto land = [&](const int slot, v4f (&r)[INST]) {      // mode 1: registers -> LDS
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < INST; ++i) *reinterpret_cast<v4f*>(smem + slot * CHUNK_KB * 1024 + (wave * INST + i) * 1024 + lane * 16) = r[i];
        }
    };
#pragma unroll
    for (int c = 0; c < D; ++c) issue(c, c, regs[c]);
    // the loop is unrolled by D so that the register sets of mode 1 are indexed statically
    int c = 0;
    for (; c + D <= iters; c += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            wait_vmcnt<INST*(D - 1)>();      // chunk c + u has landed (mode 1: in registers)
            land((c + u) % NBUF, regs[u]);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue(c + u + D, (c + u + D) % NBUF, regs[u]);
        }
    }
    wait_vmcnt<0>();
    __syncthreads();
    if (sink && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) sink[0] = reinterpret_cast<float*>(smem)[iters & 255];      // keeps the LDS image alive
#endif
}

template <int MODE, int WAVES, int CHUNK_KB>
void run(const char* src, int grid, float* sink) {
    const int iters = 300;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stage_kernel<MODE, WAVES, CHUNK_KB>), dim3(grid), dim3(64 * WAVES), 0, 0, src, iters, sink);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stage_kernel<MODE, WAVES, CHUNK_KB>), dim3(grid), dim3(64 * WAVES), 0, 0, src, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes_wg = (double)iters * CHUNK_KB * 1024;
    printf("mode %d (%s)  loader waves %d  chunk %2d KB  grid %3d : %7.1f us  %6.1f GB/s per workgroup  %6.1f GB/s per CU  %5.2f TB/s chip  (%.3f us per chunk)\n",
           MODE, MODE ? "registers + ds_write" : "LDS-DMA", WAVES, CHUNK_KB, grid, us, bytes_wg / us / 1e3, bytes_wg * grid / 256.0 / us / 1e3,
           bytes_wg * grid / us / 1e6, us / iters);
}

int main() {
    char* src;
    float* sink;
    hipMalloc(&src, (size_t)kRows * kRowBytes);
    hipMalloc(&sink, 1024);
    hipMemset(src, 1, (size_t)kRows * kRowBytes);
    for (int grid : {256, 512}) {
        run<0, 4, 8>(src, grid, sink);
        run<1, 4, 8>(src, grid, sink);
        run<0, 4, 16>(src, grid, sink);
        run<1, 4, 16>(src, grid, sink);
        run<0, 8, 8>(src, grid, sink);
        run<1, 8, 8>(src, grid, sink);
        run<0, 8, 16>(src, grid, sink);
        run<1, 8, 16>(src, grid, sink);
        run<0, 2, 8>(src, grid, sink);
        run<1, 2, 8>(src, grid, sink);
        run<0, 1, 8>(src, grid, sink);
    }
    return 0;
}
