// GPU diagnostic: does an out-of-range lane of `buffer_load_dwordx4 ... lds` (raw buffer, stride 0) WRITE ZEROS into LDS, or
// leave the old LDS bytes?  (the convolution loader wants zero-fill for halo / tail lanes without a per-lane address select)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/bufload_lds_probe.hip -o tools/probes/bufload_lds_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef void __attribute__((address_space(3))) * lds_ptr;

// second question: is the instruction's SCALAR offset part of the range check?  (lanes whose voffset is in range but whose
// voffset + soffset is not: zeros = yes, data from beyond num_records = no)
__global__ __launch_bounds__(64) void probe_soffset(const float* src, int src_bytes, int soff, float* out) {
    __shared__ __attribute__((aligned(16))) float smem[256];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.0f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, src_bytes, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)smem, 16, (int)threadIdx.x * 16, soff, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}

__global__ __launch_bounds__(64) void probe(const float* src, int src_bytes, float* out) {
    __shared__ __attribute__((aligned(16))) float smem[256];
    for (int i = threadIdx.x; i < 256; i += 64) smem[i] = -7.0f;      // stale pattern
    __syncthreads();
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0 /* stride */, src_bytes /* num_records */, 0x00020000);
    // lanes 0..31 in range (16 B each), lanes 32..63 far out of range
    const int voff = threadIdx.x < 32 ? threadIdx.x * 16 : 0x40000000 + threadIdx.x * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)smem, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = smem[i];
}

int main() {
    std::vector<float> h(128);
    for (int i = 0; i < 128; ++i) h[i] = 1.0f + i;
    float *d, *o;
    hipMalloc(&d, 512);
    hipMalloc(&o, 1024);
    hipMemcpy(d, h.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, 512, o);
    std::vector<float> r(256);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int ok_in = 0, zero_out = 0, stale_out = 0;
    for (int i = 0; i < 128; ++i) ok_in += r[i] == 1.0f + i;
    for (int i = 128; i < 256; ++i) { zero_out += r[i] == 0.0f; stale_out += r[i] == -7.0f; }
    printf("in-range floats correct: %d / 128;  out-of-range lanes: %d zeros, %d stale (of 128)\n", ok_in, zero_out, stale_out);
    printf("first out-of-range values: %g %g %g %g\n", r[128], r[129], r[130], r[131]);
    // buffer of 2048 bytes allocated, descriptor says 1024 bytes; voffset = lane * 16 (0..1008, all in range), soffset = 512:
    // lanes 32..63 address bytes 1024..1535 (inside the allocation, beyond num_records)
    std::vector<float> h2(512);
    for (int i = 0; i < 512; ++i) h2[i] = 100.0f + i;
    float* d2;
    hipMalloc(&d2, 2048);
    hipMemcpy(d2, h2.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_soffset, dim3(1), dim3(64), 0, 0, d2, 1024, 512, o);
    hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
    int in_ok = 0, z = 0, beyond = 0;
    for (int i = 0; i < 128; ++i) in_ok += r[i] == 100.0f + 128 + i;
    for (int i = 128; i < 256; ++i) { z += r[i] == 0.0f; beyond += r[i] == 100.0f + 128 + i; }
    printf("soffset test: lanes whose sum stays in range correct: %d / 128; lanes pushed out of range by soffset: %d zeros, %d read beyond num_records (of 128)\n", in_ok, z, beyond);
    // third question (round 3; csrc/train.hip prefetches chunks past a split's end with soffset = 0x7F000000 and in-range voffsets):
    // a scalar offset LARGER than num_records - does `num_records - soffset` wrap in the range check and let the lanes through?
    // Every lane's voffset is in range (0..1008 of 1024); expected: all zeros, nothing read.  Also with the largest offset the
    // kernels use for "out of range" lanes (voffset 0x80000000, soffset small).
    // (the allocation behind the 1024-byte descriptor is 1 MiB of non-zero pattern, so that a check that wrapped would read
    // pattern bytes from inside the allocation: no fault either way, and zeros can only come from the range check)
    float* d3;
    const int big = 1 << 20;
    hipMalloc(&d3, big);
    std::vector<float> h3(big / 4);
    for (int i = 0; i < big / 4; ++i) h3[i] = 5.0f + (i & 1023);
    hipMemcpy(d3, h3.data(), big, hipMemcpyHostToDevice);
    for (const int soff : {1024, 1040, 4096, 65536, 0x80000}) {
        hipLaunchKernelGGL(probe_soffset, dim3(1), dim3(64), 0, 0, d3, 1024, soff, o);
        if (hipDeviceSynchronize() != hipSuccess) { printf("soffset %#x: the launch faulted\n", soff); return 1; }
        hipMemcpy(r.data(), o, 1024, hipMemcpyDeviceToHost);
        int zeros = 0, stale = 0, other = 0;
        for (int i = 0; i < 256; ++i) { zeros += r[i] == 0.0f; stale += r[i] == -7.0f; other += r[i] != 0.0f && r[i] != -7.0f; }
        printf("soffset %#8x > num_records 1024, voffsets in range: %d zeros, %d stale, %d pattern values read past the descriptor (of 256 floats)\n",
               soff, zeros, stale, other);
    }
    return 0;
}
