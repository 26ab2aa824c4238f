// Inference pre-processing on the device for gfx950.
//
// Stands in for the host-side numpy/OpenCV chain of the reference's ROS node
// (scripts/fcn_object_detector.py:79-82 and demean_rgb_image :407-413):
//     im = frame.astype(float); im[:,:,c] -= mean[c]; im = (im - im.min()) / (im.max() - im.min())
//     im = cv.resize(im, (W, H))            # bilinear, float64 image, float coefficients
//     blob[...] = im.transpose(2, 0, 1)     # float64 -> float32
// The min/max of (px - mean[c]) over the frame equals min/max over c of (min/max_c(px) - mean[c]), so
// the reduction runs on the uint8 pixels with integer atomics (exact), and the rest is one pass that
// writes the network's NHWC input directly.  Arithmetic is float64 like the reference's.
#include <math.h>

#include "common.h"

using namespace fcn;

namespace {

__constant__ double kMeanBGR[3] = {104.0069879317889, 116.66876761696767, 122.6789143406786};

// blockIdx.y = frame of a batch of equally sized frames (frame_stride bytes apart, 8 ints of min/max scratch each)
__global__ void minmax_init_kernel(int* mm) {
    mm += 8 * blockIdx.y;
    if (threadIdx.x < 3) {
        mm[threadIdx.x] = 255;
        mm[3 + threadIdx.x] = 0;
    }
}

__global__ __launch_bounds__(256) void minmax_kernel(const uint8_t* __restrict__ frame, long long npix, int* mm, long long frame_stride) {
    frame += (size_t)blockIdx.y * frame_stride;
    mm += 8 * blockIdx.y;
    int lo[3] = {255, 255, 255}, hi[3] = {0, 0, 0};
    // Round 3: 48 bytes = 16 pixels per step as three 16-byte loads (one byte load per channel and pixel kept this pass at ~30 us for a
    // batch of 32 frames); byte b of the 48 belongs to channel b % 3.  Frames that are not 16-byte aligned, and the tail, go bytewise.
    const long long bytes = npix * 3;
    const bool vec = (((size_t)frame) & 15) == 0;
    const long long chunks = vec ? bytes / 48 : 0;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < chunks; q += (long long)gridDim.x * blockDim.x) {
        const uint4* src = reinterpret_cast<const uint4*>(frame + q * 48);
        const uint4 a = src[0], b = src[1], c = src[2];
        const unsigned dw[12] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w};
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            const int v = (int)((dw[k >> 2] >> (8 * (k & 3))) & 0xffu);
            lo[k % 3] = min(lo[k % 3], v);
            hi[k % 3] = max(hi[k % 3], v);
        }
    }
    for (long long p = chunks * 16 + (long long)blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int v = frame[p * 3 + c];
            lo[c] = min(lo[c], v);
            hi[c] = max(hi[c], v);
        }
    }
    // wave reduce, then one partial per wave through LDS, then SIX atomics per workgroup: thousands of same-address
    // atomics (one set per wave of a frame-sized grid) serialise at ~50 ns each and used to dominate this pre-processing
    __shared__ int part[6][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        for (int off = 32; off > 0; off >>= 1) {
            lo[c] = min(lo[c], __shfl_xor(lo[c], off));
            hi[c] = max(hi[c], __shfl_xor(hi[c], off));
        }
        if ((threadIdx.x & 63) == 0) {
            part[c][threadIdx.x >> 6] = lo[c];
            part[3 + c][threadIdx.x >> 6] = hi[c];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        atomicMin(&mm[c], min(min(part[c][0], part[c][1]), min(part[c][2], part[c][3])));
        atomicMax(&mm[3 + c], max(max(part[3 + c][0], part[3 + c][1]), max(part[3 + c][2], part[3 + c][3])));
    }
}

// Source windows of fcn_preprocess_bgr8_rois (detection_window_roi :257-277): x, y, w, h per image of the batch, by value in the
// kernel arguments.  n = 0: every image is its whole frame (blockIdx.y-th frame of the batch, own min/max).
constexpr int kMaxRois = 32;
struct RoiTable { int n; int r[kMaxRois][4]; };

template <typename D, bool ONES = false>
__global__ __launch_bounds__(256) void resize_norm_kernel(const uint8_t* __restrict__ frame, int h, int w, D* __restrict__ dst, int H,
                                                          int W, int cstride, float shift, const int* __restrict__ mm, long long frame_stride,
                                                          long long dst_stride, const RoiTable rois) {
    dst += (size_t)blockIdx.y * dst_stride;
    const int pitch = w;      // pixels per frame row
    if (rois.n > 0) {         // a window of the ONE frame, normalised by the frame's min/max (the node demeans before it crops, :198-200)
        const int* r = rois.r[blockIdx.y];
        frame += ((size_t)r[1] * pitch + r[0]) * 3;
        w = r[2];
        h = r[3];
    } else {
        frame += (size_t)blockIdx.y * frame_stride;
        mm += 8 * blockIdx.y;
    }
    double gmin = 1e300, gmax = -1e300;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        gmin = fmin(gmin, (double)mm[c] - kMeanBGR[c]);
        gmax = fmax(gmax, (double)mm[3 + c] - kMeanBGR[c]);
    }
    const double range = gmax - gmin;
    // Round 3: the normalised value of a pixel depends on its byte and its channel only - a table of 3 x 256 doubles per workgroup holds
    // (v - mean - min) / range, the SAME division the reference makes, once per value instead of twelve times per output pixel (four
    // taps x three channels in float64: the divisions were most of this pass).
    __shared__ double lut[3][256];
    for (int e = threadIdx.x; e < 768; e += blockDim.x) {
        const int c = e >> 8, v = e & 255;
        lut[c][v] = ((double)v - kMeanBGR[c] - gmin) / range;
    }
    __syncthreads();
    const double scale_x = (double)w / (double)W, scale_y = (double)h / (double)H;
    const long long total = (long long)H * W;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int X = (int)(t % W), Y = (int)(t / W);
        // OpenCV resize INTER_LINEAR coordinate + coefficient computation (coefficients are float)
        float fx = (float)(((double)X + 0.5) * scale_x - 0.5);
        int sx = (int)floorf(fx);
        fx -= (float)sx;
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= w - 1) { fx = 0.f; sx = w - 1; }
        float fy = (float)(((double)Y + 0.5) * scale_y - 0.5);
        int sy = (int)floorf(fy);
        fy -= (float)sy;
        if (sy < 0) { fy = 0.f; sy = 0; }
        if (sy >= h - 1) { fy = 0.f; sy = h - 1; }
        const int sx1 = min(sx + 1, w - 1), sy1 = min(sy + 1, h - 1);
        const double a0 = (double)(1.f - fx), a1 = (double)fx, b0 = (double)(1.f - fy), b1 = (double)fy;
        float out[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double s00 = lut[c][frame[((size_t)sy * pitch + sx) * 3 + c]];
            const double s01 = lut[c][frame[((size_t)sy * pitch + sx1) * 3 + c]];
            const double s10 = lut[c][frame[((size_t)sy1 * pitch + sx) * 3 + c]];
            const double s11 = lut[c][frame[((size_t)sy1 * pitch + sx1) * 3 + c]];
            const double r0 = s00 * a0 + s01 * a1;
            const double r1 = s10 * a0 + s11 * a1;
            out[c] = (float)(r0 * b0 + r1 * b1) + shift;   // blob value rounded to f32 first, as Caffe's Power layer sees it
        }
        if (ONES) {      // the half image of the f16 engine: (b, g, r, 1, 1, 0, 0, 0) in ONE 16-byte store (three 2-byte stores 16 bytes apart before)
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 px = {(_Float16)out[0], (_Float16)out[1], (_Float16)out[2], (_Float16)1.f, (_Float16)1.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            *reinterpret_cast<h8*>(dst + (size_t)t * 8) = px;
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) dst[(size_t)t * cstride + c] = (D)out[c];
        }
    }
}

}  // namespace

extern "C" {

static int preprocess_any(const uint8_t* frame, int n, int h, int w, void* dst, bool f16, int H, int W, int dst_cstride, float shift,
                          float* d_minmax, fcn_stream_t s, const int32_t* h_rois = nullptr, bool ones = false) {
    FCN_REQUIRE(frame && dst && d_minmax && n > 0 && n <= 65535 && h > 0 && w > 0 && H > 0 && W > 0 && dst_cstride >= 3, FCN_E_ARG,
                "preprocess: bad args");
    RoiTable rois = {};
    if (h_rois) {
        FCN_REQUIRE(n <= kMaxRois, FCN_E_ARG, "preprocess: at most %d windows per frame", kMaxRois);
        for (int i = 0; i < n; ++i) {
            const int32_t* r = h_rois + 4 * i;
            FCN_REQUIRE(r[0] >= 0 && r[1] >= 0 && r[2] > 0 && r[3] > 0 && (long long)r[0] + r[2] <= w && (long long)r[1] + r[3] <= h, FCN_E_ARG,
                        "preprocess: window %d (%d, %d, %d x %d) leaves the %d x %d frame", i, r[0], r[1], r[2], r[3], w, h);
            for (int k = 0; k < 4; ++k) rois.r[i][k] = r[k];
        }
        rois.n = n;
    }
    hipStream_t st = as_stream(s);
    int* mm = reinterpret_cast<int*>(d_minmax);
    const long long fstride = (long long)h * w * 3, dstride = (long long)H * W * dst_cstride;
    const int frames = h_rois ? 1 : n;      // windows share their frame's min/max
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1, frames), dim3(64), 0, st, mm);
    int mm_blocks = stream_grid((long long)h * w, 256 * 16);      // >= 16 pixels per lane: a few hundred workgroups at most
    if (mm_blocks > 256) mm_blocks = 256;
    hipLaunchKernelGGL(minmax_kernel, dim3(mm_blocks, frames), dim3(256), 0, st, frame, (long long)h * w, mm, fstride);
    FCN_REQUIRE(!ones || (f16 && dst_cstride == 8 && (((size_t)dst) & 15) == 0), FCN_E_ARG,
                "preprocess: the constant-channel form needs a half image of 8-half pixels, 16-byte aligned");
    // (the table of normalised values is built once per workgroup: 8 pixels per lane where the batch still fills the chip, fewer for a
    //  single frame, whose pre-processing is latency)
    int rn_blocks = stream_grid((long long)H * W, 256 * 8);
    const int fine = stream_grid((long long)H * W, 256), want = (1024 + n - 1) / n;
    if (rn_blocks < want) rn_blocks = want < fine ? want : fine;
    if (rn_blocks > 8192) rn_blocks = 8192;
    if (ones)
        hipLaunchKernelGGL((resize_norm_kernel<_Float16, true>), dim3(rn_blocks, n), dim3(256), 0, st, frame, h, w, reinterpret_cast<_Float16*>(dst), H, W,
                           dst_cstride, shift, mm, fstride, dstride, rois);
    else if (f16)
        hipLaunchKernelGGL(resize_norm_kernel<_Float16>, dim3(rn_blocks, n), dim3(256), 0, st, frame, h, w, reinterpret_cast<_Float16*>(dst), H, W,
                           dst_cstride, shift, mm, fstride, dstride, rois);
    else
        hipLaunchKernelGGL(resize_norm_kernel<float>, dim3(rn_blocks, n), dim3(256), 0, st, frame, h, w, reinterpret_cast<float*>(dst), H, W,
                           dst_cstride, shift, mm, fstride, dstride, rois);
    FCN_LAUNCH_CHECK("preprocess_bgr8");
    return 0;
}

int fcn_preprocess_bgr8(const uint8_t* frame, int h, int w, float* dst, int H, int W, int dst_cstride, float shift, float* d_minmax,
                        fcn_stream_t s) {
    return preprocess_any(frame, 1, h, w, dst, false, H, W, dst_cstride, shift, d_minmax, s);
}

int fcn_preprocess_bgr8_f16(const uint8_t* frame, int h, int w, void* dst, int H, int W, int dst_cstride, float shift, float* d_minmax,
                            fcn_stream_t s) {
    return preprocess_any(frame, 1, h, w, dst, true, H, W, dst_cstride, shift, d_minmax, s);
}

int fcn_preprocess_bgr8_rois(const uint8_t* frame, int h, int w, const int32_t* h_rois, int n, void* dst, int dst_f16, int H, int W,
                             int dst_cstride, float shift, float* d_minmax, fcn_stream_t s) {
    FCN_REQUIRE(h_rois, FCN_E_ARG, "preprocess_rois: null window list");
    return preprocess_any(frame, n, h, w, dst, dst_f16 != 0, H, W, dst_cstride, shift, d_minmax, s, h_rois, (dst_f16 & 2) != 0);
}

int fcn_preprocess_bgr8_batch(const uint8_t* frames, int n, int h, int w, void* dst, int dst_f16, int H, int W, int dst_cstride, float shift,
                              float* d_minmax, fcn_stream_t s) {
    return preprocess_any(frames, n, h, w, dst, dst_f16 != 0, H, W, dst_cstride, shift, d_minmax, s, nullptr, (dst_f16 & 2) != 0);
}

}  // extern "C"
