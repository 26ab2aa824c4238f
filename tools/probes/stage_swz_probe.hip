// GPU diagnostic (round 3): does the XOR swizzle of the SOURCE address cost LDS-DMA staging rate?
// conv_stream_f16's loaders sustain 60-65 GB/s per CU where tools/probes/stage_rate_probe.hip (lanes in address order) measured
// 127.  An LDS-DMA wave-instruction writes 1 KiB lane-linearly, so the swizzle that keeps the fragment reads bank-conflict free
// is applied to the source: lane l of a row fetches 16-byte segment (l % SEGS) ^ swz(row).  This probe runs the loader alone
// (ring of NBUF slots, counted vmcnt wait + barrier per chunk, 4 loader waves of an 8-wave workgroup, one workgroup per CU):
//   swz 0: lanes in address order          swz 1: 8-slot swizzle (128-byte rows)     swz 2: 4-slot swizzle (64-byte rows)
//   swz 3: 128-byte rows, XOR with bit 2 only (64-byte halves swapped, order inside a half kept)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stage_swz_probe.hip -o tools/probes/stage_swz_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef void __attribute__((address_space(3))) * lds_ptr;
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int kRowBytes = 1024;      // a pixel row of the source: consecutive rows are consecutive in memory

template <int SWZ, int INST, int NBUF>
__global__ __launch_bounds__(512) void stage_kernel(const char* __restrict__ src, int rows_total, int iters, float* __restrict__ sink) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int D = NBUF - 1, CHUNK = 4 * INST * 1024;
    __shared__ __attribute__((aligned(16))) char smem[NBUF * CHUNK];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave < 4) {      // the multiplying waves of the real kernel: barriers only
        for (int c = 0; c < iters + 2; ++c) __builtin_amdgcn_s_barrier();
        return;
    }
    const int w = wave - 4;
    constexpr int RB = SWZ == 2 ? 64 : 128, SEGS = RB / 16, RPI = 1024 / RB;
    const int lrow = lane / SEGS;
    int off[INST];
#pragma unroll
    for (int i = 0; i < INST; ++i) {
        const int row = (w * INST + i) * RPI + lrow;      // row inside the chunk's tile
        const int sw = SWZ == 0 ? 0 : SWZ == 1 ? (row >> 1) & 7 : SWZ == 2 ? (row >> 2) & 3 : (row & 1) * 4;
        const int seg = (lane % SEGS) ^ sw;
        const int grow = ((int)blockIdx.x * (4 * INST * RPI) + row) % rows_total;
        off[i] = grow * kRowBytes + seg * 16;
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(src), 0, rows_total * kRowBytes, 0x00020000);
    auto issue = [&](const int c, const int slot) {
        const int koff = (c % (kRowBytes / RB)) * RB;
#pragma unroll
        for (int i = 0; i < INST; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr)(smem + slot * CHUNK + (w * INST + i) * 1024), 16, off[i] + koff, 0, 0, 0);
    };
    int slot = 0;
    for (int c = 0; c < D; ++c) { issue(c, slot); slot = slot + 1 == NBUF ? 0 : slot + 1; }
    wait_vmcnt<INST*(D - 1)>();
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < iters; ++c) {
        wait_vmcnt<INST*(D - 2)>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(c + D, slot);
        slot = slot + 1 == NBUF ? 0 : slot + 1;
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (sink && threadIdx.x == 256 && blockIdx.x == 0x7fffffff) sink[0] = reinterpret_cast<float*>(smem)[iters & 255];
#endif
}

template <int SWZ, int INST, int NBUF>
void run(const char* src, int rows_total, float* sink, const char* what) {
    const int iters = 300, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((stage_kernel<SWZ, INST, NBUF>), dim3(grid), dim3(512), 0, 0, src, rows_total, iters, sink);
    hipEventRecord(e0, 0);
    const int reps = 5;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stage_kernel<SWZ, INST, NBUF>), dim3(grid), dim3(512), 0, 0, src, rows_total, iters, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps, bytes_wg = (double)iters * 4 * INST * 1024;
    printf("swz %d  %2d KB chunks  ring %d  source %s : %7.1f us  %6.1f GB/s per CU  (%.3f us per chunk)\n", SWZ, 4 * INST, NBUF, what, us, bytes_wg / us / 1e3,
           us / iters);
}

int main() {
    char* src;
    float* sink;
    const int big = 64 << 10, small = 6 << 10;      // rows: 64 MB (beyond the L2s), 6 MB (every workgroup's rows stay in its XCD's L2)
    hipMalloc(&src, (size_t)big * kRowBytes);
    hipMalloc(&sink, 1024);
    hipMemset(src, 1, (size_t)big * kRowBytes);
    for (int pass = 0; pass < 2; ++pass) {
        const int rows = pass ? big : small;
        const char* what = pass ? "64 MB" : " 6 MB";
        run<0, 6, 6>(src, rows, sink, what);
        run<1, 6, 6>(src, rows, sink, what);
        run<2, 6, 6>(src, rows, sink, what);
        run<3, 6, 6>(src, rows, sink, what);
        run<0, 12, 3>(src, rows, sink, what);
        run<1, 12, 3>(src, rows, sink, what);
        run<0, 8, 4>(src, rows, sink, what);
        run<1, 8, 4>(src, rows, sink, what);
    }
    return 0;
}
