// GPU diagnostic: what does straight-line code that a wave executes ONCE cost, per instruction, at the start of a kernel?
// (every launch of the convolution kernels begins with ~200 instructions of prologue and ends with ~100 of epilogue)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/icache_probe.hip -o tools/probes/icache_probe.bin && tools/probes/icache_probe.bin
// Each workgroup (256 threads, one per CU) runs a block of N independent v_add_f32 twice: pass 1 from a cold instruction
// cache, pass 2 right after (warm).  Reported: shader cycles per instruction for both passes, first launch and a launch
// that follows an identical one back to back (is the cache still warm then?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))
#define REP512(x) REP8(REP64(x))

template <int N512>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, float* sink, int passes) {
    float a = threadIdx.x, b = 1.0f;
    unsigned long long t[4];
    int k = 0;
    for (int pass = 0; pass < passes; ++pass) {
        unsigned long long t0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int r = 0; r < N512; ++r) {
            REP512(asm volatile("v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));)
        }
        unsigned long long t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt lgkmcnt(0)");
        if (k < 4) t[k++] = t1 - t0;
    }
    if (threadIdx.x == 0)
        for (int i = 0; i < k; ++i) out[blockIdx.x * 4 + i] = t[i];
    if (a == 12345.f) sink[0] = a;
}

template <int N512>
void run(const char* name) {
    const int wg = 256;
    unsigned long long* d;
    float* sink;
    hipMalloc(&d, wg * 4 * 8);
    hipMalloc(&sink, 4);
    std::vector<unsigned long long> h(wg * 4);
    for (int launch = 0; launch < 3; ++launch) {
        hipMemset(d, 0, wg * 4 * 8);
        hipDeviceSynchronize();
        if (launch == 2) {      // back to back behind an identical launch
            hipLaunchKernelGGL(probe<N512>, dim3(wg), dim3(256), 0, 0, d, sink, 2);
        }
        hipLaunchKernelGGL(probe<N512>, dim3(wg), dim3(256), 0, 0, d, sink, 2);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), d, wg * 4 * 8, hipMemcpyDeviceToHost);
        std::vector<double> p1, p2;
        for (int i = 0; i < wg; ++i) { p1.push_back((double)h[i * 4]); p2.push_back((double)h[i * 4 + 1]); }
        std::sort(p1.begin(), p1.end());
        std::sort(p2.begin(), p2.end());
        const double n = 512.0 * N512;
        printf("%-10s launch %d%s: %5.0f instr  pass1 (cold) median %7.0f cyc = %5.2f cyc/instr (min %5.2f max %5.2f)   pass2 (warm) median %7.0f cyc = %5.2f cyc/instr\n",
               name, launch, launch == 2 ? " (back to back)" : "", n, p1[wg / 2], p1[wg / 2] / n, p1[0] / n, p1[wg - 1] / n, p2[wg / 2], p2[wg / 2] / n);
    }
    hipFree(d);
    hipFree(sink);
}

int main() {
    run<1>("512");
    run<2>("1024");
    run<4>("2048");
    return 0;
}
