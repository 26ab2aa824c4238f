// Colour augmentation of composed training scenes on the device.
//
// Stands in for ArgumentationEngine.color_space_argumentation (scripts/data_argumentation_layer/argumentation_engine.py:
// 308-322), an imgaug Sequential: OneOf(GaussianBlur, AverageBlur, MedianBlur) -> Sharpen -> Add -> Multiply -> Grayscale.
// imgaug is an un-vendored, un-pinned submodule of the reference (.gitmodules), so the operators are restated from their
// documented definitions; oracle/scene_ref.py holds the same definitions in numpy and the kernels are checked against it
// bit for bit (every stage: float32 arithmetic in a fixed order, no contraction, round-half-even, saturate to uint8).
// The parameter draws stay on the host (fcn_object_detector_amd/data_layer.py::plan_color).
//
// All kernels: one lane per pixel (3 channels), images a few hundred KB: launch-latency bound, nothing to tile.
#include "common.h"

using namespace fcn;

namespace {

__device__ __forceinline__ unsigned char to_u8(float v) {
    const float r = rintf(v);
    return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

// BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
    return i;
}

__device__ __forceinline__ int clampi(int i, int n) { return i < 0 ? 0 : (i >= n ? n - 1 : i); }

struct GaussTaps { int r; float w[FCN_GAUSS_MAX_RADIUS + 1]; };

// horizontal pass: uint8 -> float32 (kept unrounded between the passes)
__global__ __launch_bounds__(256) void gauss_h_kernel(const unsigned char* __restrict__ src, float* __restrict__ tmp, int h, int w, GaussTaps t) {
    const int total = h * w;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        const unsigned char* row = src + (size_t)y * w * 3;
        float acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = t.w[0] * (float)row[x * 3 + c];
        for (int i = 1; i <= t.r; ++i) {
            const int xl = reflect101(x - i, w) * 3, xr = reflect101(x + i, w) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + t.w[i] * ((float)row[xl + c] + (float)row[xr + c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) tmp[(size_t)p * 3 + c] = acc[c];
    }
}

__global__ __launch_bounds__(256) void gauss_v_kernel(const float* __restrict__ tmp, unsigned char* __restrict__ dst, int h, int w, GaussTaps t) {
    const int total = h * w;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        float acc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] = t.w[0] * tmp[(size_t)p * 3 + c];
        for (int i = 1; i <= t.r; ++i) {
            const float* up = tmp + ((size_t)reflect101(y - i, h) * w + x) * 3;
            const float* dn = tmp + ((size_t)reflect101(y + i, h) * w + x) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = acc[c] + t.w[i] * (up[c] + dn[c]);
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(size_t)p * 3 + c] = to_u8(acc[c]);
    }
}

__global__ __launch_bounds__(256) void box_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int h, int w, int k) {
    const int total = h * w;
    const double scale = 1.0 / (double)(k * k);
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        int sum[3] = {0, 0, 0};
        for (int dy = 0; dy < k; ++dy) {
            const unsigned char* row = src + (size_t)reflect101(y - k / 2 + dy, h) * w * 3;
            for (int dx = 0; dx < k; ++dx) {
                const int xs = reflect101(x - k / 2 + dx, w) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) sum[c] += row[xs + c];
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(size_t)p * 3 + c] = (unsigned char)rint((double)sum[c] * scale);
    }
}

// median by bisection on the value: the smallest v with #(window <= v) >= (K*K+1)/2 -- 8 counting passes, no sort
template <int K>
__global__ __launch_bounds__(256) void median_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int h, int w) {
    const int total = h * w * 3;
    constexpr int NEED = (K * K + 1) / 2;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int c = p % 3, px = p / 3;
        const int y = px / w, x = px - y * w;
        unsigned char v[K * K];
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            const unsigned char* row = src + (size_t)clampi(y - K / 2 + dy, h) * w * 3 + c;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) v[dy * K + dx] = row[clampi(x - K / 2 + dx, w) * 3];
        }
        int lo = 0, hi = 255;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int mid = (lo + hi) >> 1;
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < K * K; ++i) cnt += v[i] <= mid;
            if (cnt >= NEED) hi = mid; else lo = mid + 1;
        }
        dst[p] = (unsigned char)lo;
    }
}

__global__ __launch_bounds__(256) void color_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, int h, int w,
                                                    fcn_color_params q) {
    const int total = h * w;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
        const int y = p / w, x = p - y * w;
        float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const unsigned char* row = src + (size_t)reflect101(y + dy, h) * w * 3;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int xs = reflect101(x + dx, w) * 3;
                const float coef = (dy == 0 && dx == 0) ? q.sharpen_centre : q.sharpen_off;
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] = acc[c] + coef * (float)row[xs + c];
            }
        }
        int v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int t = (int)to_u8(acc[c]) + q.add[c];                       // Sharpen, then Add (saturating)
            t = t < 0 ? 0 : (t > 255 ? 255 : t);
            v[c] = (int)to_u8((float)t * q.mul[c]);                       // Multiply
        }
        const int grey = (v[0] * 4899 + v[1] * 9617 + v[2] * 1868 + 8192) >> 14;
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(size_t)p * 3 + c] = to_u8(q.gray_keep * (float)v[c] + q.gray_alpha * (float)grey);
    }
}

int check_image(const void* src, const void* dst, int h, int w, const char* who) {
    FCN_REQUIRE(src && dst && src != dst && h > 0 && w > 0 && (long long)h * w < (1ll << 28), FCN_E_ARG, "%s: bad image arguments", who);
    return 0;
}

}  // namespace

extern "C" {

int fcn_blur_gauss_bgr8(const uint8_t* src, uint8_t* dst, float* tmp, int h, int w, const float* h_taps, int radius, fcn_stream_t s) {
    if (int rc = check_image(src, dst, h, w, "blur_gauss")) return rc;
    FCN_REQUIRE(tmp && h_taps && radius >= 0 && radius <= FCN_GAUSS_MAX_RADIUS, FCN_E_ARG, "blur_gauss: radius %d outside [0, %d]", radius,
                FCN_GAUSS_MAX_RADIUS);
    GaussTaps t;
    t.r = radius;
    for (int i = 0; i <= FCN_GAUSS_MAX_RADIUS; ++i) t.w[i] = i <= radius ? h_taps[i] : 0.f;
    const int grid = stream_grid((long long)h * w, 256);
    hipLaunchKernelGGL(gauss_h_kernel, dim3(grid), dim3(256), 0, as_stream(s), src, tmp, h, w, t);
    hipLaunchKernelGGL(gauss_v_kernel, dim3(grid), dim3(256), 0, as_stream(s), tmp, dst, h, w, t);
    FCN_LAUNCH_CHECK("blur_gauss");
    return 0;
}

int fcn_blur_box_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, int k, fcn_stream_t s) {
    if (int rc = check_image(src, dst, h, w, "blur_box")) return rc;
    FCN_REQUIRE(k >= 1 && k <= 15, FCN_E_ARG, "blur_box: k %d outside [1, 15]", k);
    hipLaunchKernelGGL(box_kernel, dim3(stream_grid((long long)h * w, 256)), dim3(256), 0, as_stream(s), src, dst, h, w, k);
    FCN_LAUNCH_CHECK("blur_box");
    return 0;
}

int fcn_blur_median_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, int k, fcn_stream_t s) {
    if (int rc = check_image(src, dst, h, w, "blur_median")) return rc;
    const dim3 grid(stream_grid((long long)h * w * 3, 256)), block(256);
    switch (k) {
        case 3: hipLaunchKernelGGL(median_kernel<3>, grid, block, 0, as_stream(s), src, dst, h, w); break;
        case 5: hipLaunchKernelGGL(median_kernel<5>, grid, block, 0, as_stream(s), src, dst, h, w); break;
        case 7: hipLaunchKernelGGL(median_kernel<7>, grid, block, 0, as_stream(s), src, dst, h, w); break;
        default: return set_err(FCN_E_UNSUPPORTED, "blur_median: k must be 3, 5 or 7 (got %d)", k);
    }
    FCN_LAUNCH_CHECK("blur_median");
    return 0;
}

int fcn_color_augment_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, const fcn_color_params* h_params, fcn_stream_t s) {
    if (int rc = check_image(src, dst, h, w, "color_augment")) return rc;
    FCN_REQUIRE(h_params, FCN_E_ARG, "color_augment: no parameters");
    hipLaunchKernelGGL(color_kernel, dim3(stream_grid((long long)h * w, 256)), dim3(256), 0, as_stream(s), src, dst, h, w, *h_params);
    FCN_LAUNCH_CHECK("color_augment");
    return 0;
}

}  // extern "C"
