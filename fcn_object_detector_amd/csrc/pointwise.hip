// HBM-bound layers of the detector for gfx950: pooling, LRN, element-wise, layout changes.
//
// Stands in for Caffe's PoolingLayer / LRNLayer / ReLULayer / SigmoidLayer / PowerLayer /
// EltwiseLayer / DeconvolutionLayer(group == channels) forward passes as executed by
// net.forward() (reference: scripts/fcn_object_detector.py:87) over models/deploy.prototxt
// (pool :54-64, LRN :65-75) and train/fcn_bbox/train_val.prototxt (deconv :544-565, eltwise
// :567-651).  All activations are NHWC: a pixel's channels are contiguous, so every kernel here
// moves 16 bytes per lane (4 channels) with consecutive lanes on consecutive addresses, and the
// LRN channel window is a contiguous run inside one pixel.
#include <float.h>
#include <math.h>

#include "common.h"

using namespace fcn;

namespace {

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------
// NCHW <-> NHWC (pycaffe boundary).  32x32 LDS tile transpose over (C, H*W) per image.
// ---------------------------------------------------------------------------------------------
template <typename D>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ src, D* __restrict__ dst,
                                                           int C, int HW, int dst_cstride, int dst_coffset, float shift) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const float* s = src + (size_t)n * C * HW;
    D* d = dst + (size_t)n * HW * dst_cstride;
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? s[(size_t)c * HW + p] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) d[(size_t)p * dst_cstride + dst_coffset + c] = (D)(tile[tx][i] + shift);
    }
}

template <typename S>
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const S* __restrict__ src, float* __restrict__ dst,
                                                           int C, int HW, int src_cstride, int src_coffset) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const S* s = src + (size_t)n * HW * src_cstride;
    float* d = dst + (size_t)n * C * HW;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        tile[i][tx] = (p < HW && c < C) ? (float)s[(size_t)p * src_cstride + src_coffset + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) d[(size_t)c * HW + p] = tile[tx][i];
    }
}

// The net's input: at most four channels into 4-float pixels (the image of models/deploy.prototxt is 3 + 1 pad).  The tile transpose
// above spends a 32 x 32 tile on three channels and stores 12 bytes per pixel; here a lane owns a pixel - C coalesced plane reads, one
// 16-byte store (the pad channels are written as zeros: they belong to this blob) - 9.6 -> ~3 us for a 448 x 448 frame, a layout
// kernel of every frame of SURVEY 8(d) config 2's region.
__global__ __launch_bounds__(256) void nchw_to_nhwc4_kernel(const float* __restrict__ src, float* __restrict__ dst, int C, int HW, float shift) {
    const int n = blockIdx.y;
    const float* s = src + (size_t)n * C * HW;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += gridDim.x * blockDim.x) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        v.x = s[p] + shift;
        if (C > 1) v.y = s[(size_t)HW + p] + shift;
        if (C > 2) v.z = s[2 * (size_t)HW + p] + shift;
        if (C > 3) v.w = s[3 * (size_t)HW + p] + shift;
        *reinterpret_cast<float4*>(dst + ((size_t)n * HW + p) * 4) = v;
    }
}

// Several small blobs NHWC -> NCHW in ONE launch (the two head blobs behind every forward: two launches of a few microseconds each
// were launch floor, not work).  One lane per output element.
constexpr int kMaxLayoutBlobs = 8;
struct LayoutBlob { const float* src; float* dst; int C, HW, cstride, coffset, count, end; };      // count = N*C*HW, end = exclusive prefix
struct LayoutArgs { int n; LayoutBlob b[kMaxLayoutBlobs]; };
__global__ __launch_bounds__(256) void nhwc_to_nchw_multi_kernel(const LayoutArgs a) {
    const int total = a.b[a.n - 1].end;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 0; j < kMaxLayoutBlobs - 1; ++j) k += (j + 1 < a.n && i >= a.b[j].end) ? 1 : 0;
        LayoutBlob q = a.b[0];
#pragma unroll
        for (int j = 1; j < kMaxLayoutBlobs; ++j)
            if (k == j) q = a.b[j];
        const int e = i - (q.end - q.count);            // element of this blob in NCHW order
        const int p = e % q.HW, nc = e / q.HW;
        const int c = nc % q.C, n = nc / q.C;
        q.dst[e] = q.src[((size_t)n * q.HW + p) * q.cstride + q.coffset + c];
    }
}

// ---------------------------------------------------------------------------------------------
// MAX pooling (Caffe semantics: window clipped to the image, strict '>' so the first maximum in
// raster order wins, start value -FLT_MAX).  One lane = 4 channels of one output pixel.
// ---------------------------------------------------------------------------------------------
template <bool VEC4, bool WITH_IDX>
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int32_t* __restrict__ idx,
                                                      int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                                                      int OH, int OW, int y_cstride, int y_coffset) {
    const int cg = VEC4 ? C / 4 : C;  // channel groups per pixel
    const long long total = (long long)N * OH * OW * cg;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t % cg);
        long long pix = t / cg;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        int hs = oy * stride - pad, ws = ox * stride - pad;
        const int he = min(hs + k, H), we = min(ws + k, W);
        hs = max(hs, 0);
        ws = max(ws, 0);
        const float* xb = x + (size_t)n * H * W * x_cstride + (VEC4 ? g * 4 : g);
        const size_t o = ((size_t)(n * OH + oy) * OW + ox);
        if (VEC4) {
            float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
            int4 mi = make_int4(-1, -1, -1, -1);
            if (k == 3) {
                // 3 x 3 windows (every pooling of the reference's nets): all nine loads in flight before the first compare - the loop below
                // waits for each load in turn, nine dependent round trips (pool3/3x3_s2 at batch 1: 5.8 us in the kernel trace for 6 MB).  Taps
                // outside the image load the nearest pixel inside and are skipped by the compares: same maximum, same first-maximum index.
                const int y0 = oy * stride - pad, x0 = ox * stride - pad;
                float4 v[9];
                int id[9];
                bool in[9];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const int iy = min(max(y0 + r, 0), H - 1);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const int ix = min(max(x0 + c, 0), W - 1);
                        in[3 * r + c] = (unsigned)(y0 + r) < (unsigned)H && (unsigned)(x0 + c) < (unsigned)W;
                        id[3 * r + c] = iy * W + ix;
                        v[3 * r + c] = ld4(xb + (size_t)id[3 * r + c] * x_cstride);
                    }
                }
#pragma unroll
                for (int t9 = 0; t9 < 9; ++t9) {
                    if (in[t9] & (v[t9].x > m.x)) { m.x = v[t9].x; mi.x = id[t9]; }
                    if (in[t9] & (v[t9].y > m.y)) { m.y = v[t9].y; mi.y = id[t9]; }
                    if (in[t9] & (v[t9].z > m.z)) { m.z = v[t9].z; mi.z = id[t9]; }
                    if (in[t9] & (v[t9].w > m.w)) { m.w = v[t9].w; mi.w = id[t9]; }
                }
            } else
            for (int iy = hs; iy < he; ++iy)
                for (int ix = ws; ix < we; ++ix) {
                    const float4 v = ld4(xb + ((size_t)iy * W + ix) * x_cstride);
                    const int id = iy * W + ix;
                    if (v.x > m.x) { m.x = v.x; mi.x = id; }
                    if (v.y > m.y) { m.y = v.y; mi.y = id; }
                    if (v.z > m.z) { m.z = v.z; mi.z = id; }
                    if (v.w > m.w) { m.w = v.w; mi.w = id; }
                }
            st4(y + o * y_cstride + y_coffset + g * 4, m);
            if (WITH_IDX) *reinterpret_cast<int4*>(idx + o * C + g * 4) = mi;
        } else {
            float m = -FLT_MAX;
            int mi = -1;
            for (int iy = hs; iy < he; ++iy)
                for (int ix = ws; ix < we; ++ix) {
                    const float v = xb[((size_t)iy * W + ix) * x_cstride];
                    if (v > m) { m = v; mi = iy * W + ix; }
                }
            y[o * y_cstride + y_coffset + g] = m;
            if (WITH_IDX) idx[o * C + g] = mi;
        }
    }
}

// AVE pooling: divisor = window area clipped to H+pad (Caffe counts the padding), sum over the image part
__global__ __launch_bounds__(256) void avepool_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C,
                                                      int x_cstride, int k, int stride, int pad, int OH, int OW, int y_cstride,
                                                      int y_coffset) {
    const long long total = (long long)N * OH * OW * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        long long pix = t / C;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        int hs = oy * stride - pad, ws = ox * stride - pad;
        int he = min(hs + k, H + pad), we = min(ws + k, W + pad);
        const float area = (float)((he - hs) * (we - ws));
        hs = max(hs, 0);
        ws = max(ws, 0);
        he = min(he, H);
        we = min(we, W);
        const float* xb = x + (size_t)n * H * W * x_cstride + c;
        float acc = 0.f;
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) acc += xb[((size_t)iy * W + ix) * x_cstride];
        y[((size_t)(n * OH + oy) * OW + ox) * y_cstride + y_coffset + c] = acc / area;
    }
}

// ---------------------------------------------------------------------------------------------
// LRN across channels, local_size 5 fast path: one lane = 4 channels; the 5-wide window of those
// 4 channels lives in the 12 floats c-4..c+7 of the same pixel (three 16-byte loads).
// ---------------------------------------------------------------------------------------------
// s^-beta; beta = 0.75 (every LRN of the reference nets) is two square roots and a reciprocal instead of powf
__device__ __forceinline__ float pow_neg_beta(float s, float beta) {
    if (beta == 0.75f) {
        const float r = sqrtf(s);
        return 1.f / (r * sqrtf(r));
    }
    return powf(s, -beta);
}

// The same for results that are rounded to half precision: the hardware reciprocal square root and square root (1 ulp of
// float32, far below half an f16 ulp) instead of the correctly rounded sequences - the f16 LRN is ALU-bound otherwise
// (8 outputs per lane, two IEEE square roots and a division each: 114 us for conv2/norm2 at batch 32, 2.7 TB/s).
__device__ __forceinline__ float pow_neg_beta_fast(float s, float beta) {
    if (beta == 0.75f) {
        const float r = __builtin_amdgcn_rsqf(s);
        return r * __builtin_amdgcn_sqrtf(r);
    }
    return powf(s, -beta);
}

// LRN of the 4 channels in `c` given their left / right neighbour groups (zeros outside the blob): scale -> s, returns x * s^-beta
template <bool FAST = false>
__device__ __forceinline__ float4 lrn5_apply(const float4 l, const float4 c, const float4 r, float alpha_over_n, float beta, float kk, float4& s) {
    const float q[12] = {l.x * l.x, l.y * l.y, l.z * l.z, l.w * l.w, c.x * c.x, c.y * c.y,
                         c.z * c.z, c.w * c.w, r.x * r.x, r.y * r.y, r.z * r.z, r.w * r.w};
    s.x = kk + alpha_over_n * (q[2] + q[3] + q[4] + q[5] + q[6]);
    s.y = kk + alpha_over_n * (q[3] + q[4] + q[5] + q[6] + q[7]);
    s.z = kk + alpha_over_n * (q[4] + q[5] + q[6] + q[7] + q[8]);
    s.w = kk + alpha_over_n * (q[5] + q[6] + q[7] + q[8] + q[9]);
    float4 o;
    o.x = c.x * (FAST ? pow_neg_beta_fast(s.x, beta) : pow_neg_beta(s.x, beta));
    o.y = c.y * (FAST ? pow_neg_beta_fast(s.y, beta) : pow_neg_beta(s.y, beta));
    o.z = c.z * (FAST ? pow_neg_beta_fast(s.z, beta) : pow_neg_beta(s.z, beta));
    o.w = c.w * (FAST ? pow_neg_beta_fast(s.w, beta) : pow_neg_beta(s.w, beta));
    return o;
}

__global__ __launch_bounds__(256) void lrn5_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ scale,
                                                   long long pixels, int C, int x_cstride, int y_cstride, float alpha_over_n,
                                                   float beta, float kk) {
    const int cg = C / 4;
    const long long total = pixels * cg;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t % cg);
        const long long pix = t / cg;
        const float* xp = x + (size_t)pix * x_cstride + g * 4;
        const float4 c = ld4(xp);
        const float4 l = g > 0 ? ld4(xp - 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 r = g + 1 < cg ? ld4(xp + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 s;
        const float4 o = lrn5_apply(l, c, r, alpha_over_n, beta, kk, s);
        st4(y + (size_t)pix * y_cstride + g * 4, o);
        if (scale) st4(scale + (size_t)pix * C + g * 4, s);
    }
}

// MAX pooling and LRN (local_size 5) of the same blob in one pass, inference only (nothing kept for backward):
// LRN_FIRST false: y = LRN(maxpool(x))  (pool1 -> norm1);  true: y = maxpool(LRN(x))  (norm2 -> pool2).
// The workgroup covers an 8 x 8 patch of output pixels x 32 channels (lane = 4 channels): the overlapping windows and the
// neighbour channel groups the LRN window needs are re-read from the CU's L1, and the blob in the middle never exists.
__device__ __forceinline__ float4 max4(const float4 a, const float4 b) {      // strict '>' like maxpool_kernel (first maximum stays)
    return make_float4(b.x > a.x ? b.x : a.x, b.y > a.y ? b.y : a.y, b.z > a.z ? b.z : a.z, b.w > a.w ? b.w : a.w);
}

template <bool LRN_FIRST>
__global__ __launch_bounds__(512) void maxpool_lrn5_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int x_cstride,
                                                           int k, int stride, int pad, int OH, int OW, int y_cstride, int cgroups,
                                                           float alpha_over_n, float beta, float kk) {
    const int cg32 = (int)blockIdx.x % cgroups, tx = (int)blockIdx.x / cgroups;
    const int g = cg32 * 8 + ((int)threadIdx.x & 7), cg = C / 4;
    const int pp = (int)threadIdx.x >> 3;
    const int oy = (int)blockIdx.y * 8 + (pp >> 3), ox = tx * 8 + (pp & 7), n = (int)blockIdx.z;
    if (g >= cg || oy >= OH || ox >= OW) return;
    int hs = oy * stride - pad, ws = ox * stride - pad;
    const int he = min(hs + k, H), we = min(ws + k, W);
    hs = max(hs, 0);
    ws = max(ws, 0);
    const float* xb = x + (size_t)n * H * W * x_cstride + g * 4;
    const bool hl = g > 0, hr = g + 1 < cg;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f), low = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    float4 m = low, ml = hl ? low : zero, mr = hr ? low : zero, s;
    for (int iy = hs; iy < he; ++iy)
        for (int ix = ws; ix < we; ++ix) {
            const float* xp = xb + ((size_t)iy * W + ix) * x_cstride;
            const float4 c = ld4(xp);
            const float4 l = hl ? ld4(xp - 4) : zero;
            const float4 r = hr ? ld4(xp + 4) : zero;
            if (LRN_FIRST) {
                // every input position is normalised once per window it lies in (2.25 times on average at stride 2): the
                // hardware rsq / sqrt (1 ulp) instead of the IEEE sequences of the stand-alone LRN kernel
                m = max4(m, lrn5_apply<true>(l, c, r, alpha_over_n, beta, kk, s));
            } else {
                m = max4(m, c);
                if (hl) ml = max4(ml, l);
                if (hr) mr = max4(mr, r);
            }
        }
    if (!LRN_FIRST) m = lrn5_apply(ml, m, mr, alpha_over_n, beta, kk, s);
    st4(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + g * 4, m);
}

// pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce (+ ReLU) of models/deploy.prototxt:54-104 as ONE launch (inference, float32, 64 -> 64
// channels, 3 x 3 window).  As a launch of its own the 1x1 convolution is two chunks of K behind the whole fixed cost of a convolution
// launch (5.6 us in the kernel trace at batch 1, 18 TF/s); here the workgroup that has just normalised a 4 x 8 patch of pixels multiplies
// them by the 64 x 64 filter bank before they leave the CU, so neither pooled nor normalised pixels ever reach HBM:
//   1. lane (pixel p = tid / 8, q = tid % 8) takes the maximum of channels 8q .. 8q+7 over its window - all 18 loads in flight, taps
//      outside the image re-read the nearest tap inside (the maximum does not change);
//   2. LRN: the two channels either side of the lane's eight come from lanes q - 1 / q + 1 (same pixel, same wave) by a lane exchange;
//      lrn5_apply<true>: the hardware rsq / sqrt (1 ulp), as in the LRN-first single pass;
//   3. the 32 x 64 normalised tile goes to LDS (pitch 68 floats), and wave w computes output channels 16w .. 16w+15 of the 32 pixels with
//      v_mfma_f32_16x16x4_f32 as out^T = W . act^T: A = 16 filters x 4 k-values (lane (i, h): filter 16w + i, k = 16h + s in step s -
//      any bijection of k serves a sum, this one makes a lane's filter values ONE 64-byte run, loaded at the top of the kernel),
//      B = 4 k-values x 16 pixels (four ds_read_b128 per pixel tile); D's register r of lane (j, h) is channel 16w + 4h + r of pixel j,
//      so bias, ReLU and a 16-byte store follow straight from the accumulators.
constexpr int kPLC_C = 64, kPLC_Pitch = 68;
__global__ __launch_bounds__(256) void pool3_lrn5_conv1x1_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 float* __restrict__ y, int H, int W, int x_cstride, int stride, int pad, int OH, int OW,
                                                                 int y_cstride, int y_coffset, int relu, float alpha_over_n, float beta, float kk) {
    __shared__ __attribute__((aligned(16))) float act[32 * kPLC_Pitch];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = (int)blockIdx.z;
    // filter fragments first: their latency hides behind the pooling loads
    const int fi = lane & 15, fh = lane >> 4;
    const float* wp = w + (size_t)(16 * wave + fi) * kPLC_C + 16 * fh;
    float4 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = ld4(wp + 4 * j);
    const float4 bv = bias ? ld4(bias + 16 * wave + 4 * fh) : make_float4(0.f, 0.f, 0.f, 0.f);
    // 1. pooling
    const int p = tid >> 3, q = tid & 7;
    const int oy = min((int)blockIdx.y * 4 + (p >> 3), OH - 1), ox = min((int)blockIdx.x * 8 + (p & 7), OW - 1);
    const int hs = oy * stride - pad, ws = ox * stride - pad;
    const int h0 = max(hs, 0), h1 = min(hs + 3, H) - 1, w0 = max(ws, 0), w1 = min(ws + 3, W) - 1;
    const float* xb = x + (size_t)n * H * W * x_cstride + 8 * q;
    float4 t0[9], t1[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int iy = min(max(hs + r, h0), h1), ix = min(max(ws + c, w0), w1);
            const float* xp = xb + ((size_t)iy * W + ix) * x_cstride;
            t0[3 * r + c] = ld4(xp);
            t1[3 * r + c] = ld4(xp + 4);
        }
    float4 m0 = t0[0], m1 = t1[0];
#pragma unroll
    for (int i = 1; i < 9; ++i) {
        m0 = max4(m0, t0[i]);
        m1 = max4(m1, t1[i]);
    }
    // 2. LRN over the 64 channels of the pixel: neighbours' groups from lanes tid - 1 / tid + 1 (zeros outside the blob)
    float4 lft, rgt;
    lft.x = __shfl_up(m1.x, 1); lft.y = __shfl_up(m1.y, 1); lft.z = __shfl_up(m1.z, 1); lft.w = __shfl_up(m1.w, 1);
    rgt.x = __shfl_down(m0.x, 1); rgt.y = __shfl_down(m0.y, 1); rgt.z = __shfl_down(m0.z, 1); rgt.w = __shfl_down(m0.w, 1);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (q == 0) lft = zero;
    if (q == 7) rgt = zero;
    float4 s;
    // (the hardware reciprocal square root / square root, 1 ulp each - what the LRN-first single pass uses - instead of the IEEE sequences: eight
    //  elements per lane at ~100 instructions each were 2 us of this kernel)
    const float4 a0 = lrn5_apply<true>(lft, m0, m1, alpha_over_n, beta, kk, s);
    const float4 a1 = lrn5_apply<true>(m0, m1, rgt, alpha_over_n, beta, kk, s);
    st4(act + p * kPLC_Pitch + 8 * q, a0);
    st4(act + p * kPLC_Pitch + 8 * q + 4, a1);
    __syncthreads();
    // 3. out^T = W . act^T on the matrix cores
    typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* ap = act + (16 * t + fi) * kPLC_Pitch + 16 * fh;
        float4 af[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) af[j] = ld4(ap + 4 * j);
        v4f acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].x, af[j].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].y, af[j].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].z, af[j].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j].w, af[j].w, acc, 0, 0, 0);
        }
        float4 o = make_float4(acc[0] + bv.x, acc[1] + bv.y, acc[2] + bv.z, acc[3] + bv.w);
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        // (16 bytes per lane, 64 contiguous bytes per pixel and wave: taking this 3.2 MB tile through LDS for whole 256-byte rows was tried
        //  and is slower - 5.6 -> 5.8 us, two more barriers; the half-float twin's 8-byte stores did gain from it)
        const int pp = 16 * t + fi;
        const int py = (int)blockIdx.y * 4 + (pp >> 3), px = (int)blockIdx.x * 8 + (pp & 7);
        if (py < OH && px < OW) st4(y + ((size_t)(n * OH + py) * OW + px) * y_cstride + y_coffset + 16 * wave + 4 * fh, o);
    }
}

// conv2/norm2 -> pool2/3x3_s2 of models/deploy.prototxt:137-158 (float32 inference, batch 1) through an LDS patch: the single pass above
// normalises every input pixel once per WINDOW it lies in (2.25 times on average, each time from three 16-byte loads) - 8.2 us for a
// 9.6 MB blob, 10.2 in the kernel trace.  Here a workgroup owns TH x 8 output pixels x ALL channels: the (2 TH + 1) x 17 input pixels
// under them go to LDS once by LDS-DMA (pixels outside the image - ceil-mode windows that hang over the edge - are replaced by the
// nearest pixel inside: same maximum), the LRN runs once per patch pixel in place (lrn5_apply<true>: the single pass's arithmetic, bit
// for bit), and the windows are pooled from LDS.  3 x 3 / stride 2 windows without padding; the patch must fit 64 KiB.
__global__ __launch_bounds__(512) void lrn5_pool3s2_lds_kernel(const float* __restrict__ x, float* __restrict__ y, int H, int W, int C, int x_cstride, int OH,
                                                             int OW, int y_cstride, int TH, float alpha_over_n, float beta, float kk) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const void __attribute__((address_space(1))) * gptr;
    typedef void __attribute__((address_space(3))) * lptr;
    constexpr int TW = 8, PWp = 2 * TW + 1, NT = 512, MAXI = 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int segs = C / 4, pitch = C * 4;
    const int oy0 = (int)blockIdx.y * TH, ox0 = (int)blockIdx.x * TW, n = (int)blockIdx.z;
    const int PH = 2 * TH + 1, npix = PH * PWp, nitems = npix * segs;
    const float* xn = x + (size_t)n * H * W * x_cstride;
    for (int i = wave; i * 64 < nitems; i += NT / 64) {
        int g = i * 64 + lane;
        g = g < nitems ? g : nitems - 1;
        const int p = g / segs, sg = g - p * segs;
        const int pr = p / PWp, pc = p - pr * PWp;
        int iy = 2 * oy0 + pr, ix = 2 * ox0 + pc;
        iy = iy >= H ? H - 1 : iy;
        ix = ix >= W ? W - 1 : ix;
        __builtin_amdgcn_global_load_lds((gptr)(xn + ((size_t)iy * W + ix) * x_cstride + sg * 4), (lptr)(lds + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 res[MAXI];      // normalise in place: all results first (they read their neighbours' raw values), then all writes
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int g = tid + NT * j;
        if (g < nitems) {
            const int sg = g % segs;
            const char* at = lds + (size_t)g * 16;
            const float4 c = *reinterpret_cast<const float4*>(at);
            const float4 l = sg > 0 ? *reinterpret_cast<const float4*>(at - 16) : zero;
            const float4 r = sg + 1 < segs ? *reinterpret_cast<const float4*>(at + 16) : zero;
            float4 sc;
            res[j] = lrn5_apply<true>(l, c, r, alpha_over_n, beta, kk, sc);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
        const int g = tid + NT * j;
        if (g < nitems) *reinterpret_cast<float4*>(lds + (size_t)g * 16) = res[j];
    }
    __syncthreads();
    const int nout = TH * TW * segs;
    for (int g = tid; g < nout; g += NT) {
        const int op = g / segs, sg = g - op * segs;
        const int oyl = op / TW, oxl = op - oyl * TW;
        const int oy = oy0 + oyl, ox = ox0 + oxl;
        if (oy >= OH || ox >= OW) continue;
        const char* w0 = lds + (size_t)((2 * oyl) * PWp + 2 * oxl) * pitch + sg * 16;
        float4 m = *reinterpret_cast<const float4*>(w0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (dy || dx) m = max4(m, *reinterpret_cast<const float4*>(w0 + (size_t)(dy * PWp + dx) * pitch));
        st4(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + sg * 4, m);
    }
}

// generic window (any odd/even local_size, any C)
__global__ __launch_bounds__(256) void lrn_generic_kernel(const float* __restrict__ x, float* __restrict__ y, float* __restrict__ scale,
                                                          long long pixels, int C, int x_cstride, int y_cstride, int local_size,
                                                          float alpha_over_n, float beta, float kk) {
    const long long total = pixels * C;
    const int pre = (local_size - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long pix = t / C;
        const float* xp = x + (size_t)pix * x_cstride;
        float acc = 0.f;
        for (int j = c - pre; j < c - pre + local_size; ++j)
            if (j >= 0 && j < C) acc += xp[j] * xp[j];
        const float s = kk + alpha_over_n * acc;
        y[(size_t)pix * y_cstride + c] = xp[c] * powf(s, -beta);
        if (scale) scale[(size_t)pix * C + c] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// flat element-wise ops (count floats, 16 B per lane when count and pointers allow)
// ---------------------------------------------------------------------------------------------
enum { OP_RELU = 0, OP_SIGMOID = 1, OP_POWER = 2 };

template <int OP>
__device__ __forceinline__ float unary(float v, float a, float b, float c) {
    if (OP == OP_RELU) return v > 0.f ? v : v * a;
    if (OP == OP_SIGMOID) return 1.f / (1.f + expf(-v));
    const float t = c + b * v;  // Power: (shift + scale * x) ^ power
    return a == 1.f ? t : powf(t, a);
}

template <int OP>
__global__ __launch_bounds__(256) void unary_kernel(const float* __restrict__ x, float* __restrict__ y, size_t count, float a, float b,
                                                    float c) {
    const size_t n4 = count / 4;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = ld4(x + i * 4);
        v.x = unary<OP>(v.x, a, b, c);
        v.y = unary<OP>(v.y, a, b, c);
        v.z = unary<OP>(v.z, a, b, c);
        v.w = unary<OP>(v.w, a, b, c);
        st4(y + i * 4, v);
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) y[i] = unary<OP>(x[i], a, b, c);
}

__global__ __launch_bounds__(256) void unary_scalar_kernel(const float* __restrict__ x, float* __restrict__ y, size_t count, int op,
                                                           float a, float b, float c) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float v = x[i];
        y[i] = op == OP_RELU ? unary<OP_RELU>(v, a, b, c) : op == OP_SIGMOID ? unary<OP_SIGMOID>(v, a, b, c) : unary<OP_POWER>(v, a, b, c);
    }
}

__global__ __launch_bounds__(256) void eltwise_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y,
                                                      size_t count, int op, float ca, float cb) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float u = a[i], v = b[i];
        y[i] = op == FCN_ELT_PROD ? u * v : op == FCN_ELT_SUM ? ca * u + cb * v : fmaxf(u, v);
    }
}

__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst, long long pixels,
                                                            int C, int src_cstride, int src_coffset, int dst_cstride, int dst_coffset) {
    const long long total = pixels * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long pix = t / C;
        dst[(size_t)pix * dst_cstride + dst_coffset + c] = src[(size_t)pix * src_cstride + src_coffset + c];
    }
}

// depthwise transposed convolution (Caffe Deconvolution with group == channels), gather form:
// y[oy][ox][c] = b[c] + sum over (r, q) with (oy + p - r) % s == 0, (ox + p - q) % s == 0 of
//                x[(oy+p-r)/s][(ox+p-q)/s][c] * w[c][r][q]
// Softmax over the channels of every pixel (Caffe SoftmaxLayer, axis 1): channels are contiguous in NHWC, one lane per pixel
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, long long pixels, int C,
                                                      int x_cstride, int y_cstride) {
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < pixels; pix += (long long)gridDim.x * blockDim.x) {
        const float* xp = x + (size_t)pix * x_cstride;
        float* yp = y + (size_t)pix * y_cstride;
        float m = xp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, xp[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(xp[c] - m);
        for (int c = 0; c < C; ++c) yp[c] = expf(xp[c] - m) / sum;
    }
}

__global__ __launch_bounds__(256) void deconv_dw_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int N, int H, int W, int C,
                                                        int x_cstride, int k, int stride, int pad, int OH, int OW, int y_cstride,
                                                        int y_coffset) {
    const long long total = (long long)N * OH * OW * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        long long pix = t / C;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        float acc = bias ? bias[c] : 0.f;
        const float* xb = x + (size_t)n * H * W * x_cstride + c;
        const float* wc = w + (size_t)c * k * k;
        for (int r = (oy + pad) % stride; r < k; r += stride) {
            const int iy = (oy + pad - r) / stride;
            if (oy + pad - r < 0 || iy >= H) continue;
            for (int q = (ox + pad) % stride; q < k; q += stride) {
                const int ix = (ox + pad - q) / stride;
                if (ox + pad - q < 0 || ix >= W) continue;
                acc += xb[((size_t)iy * W + ix) * x_cstride] * wc[r * k + q];
            }
        }
        y[((size_t)(n * OH + oy) * OW + ox) * y_cstride + y_coffset + c] = acc;
    }
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int launch_unary(int op, const float* x, float* y, size_t count, float a, float b, float c, fcn_stream_t s) {
    FCN_REQUIRE(x && y, FCN_E_ARG, "unary: null");
    if (count == 0) return 0;
    hipStream_t st = as_stream(s);
    if (aligned16(x) && aligned16(y)) {
        const int grid = stream_grid((long long)(count / 4 + 1), 256);
        if (op == OP_RELU) hipLaunchKernelGGL(unary_kernel<OP_RELU>, dim3(grid), dim3(256), 0, st, x, y, count, a, b, c);
        else if (op == OP_SIGMOID) hipLaunchKernelGGL(unary_kernel<OP_SIGMOID>, dim3(grid), dim3(256), 0, st, x, y, count, a, b, c);
        else hipLaunchKernelGGL(unary_kernel<OP_POWER>, dim3(grid), dim3(256), 0, st, x, y, count, a, b, c);
    } else {
        hipLaunchKernelGGL(unary_scalar_kernel, dim3(stream_grid((long long)count, 256)), dim3(256), 0, st, x, y, count, op, a, b, c);
    }
    FCN_LAUNCH_CHECK("unary");
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// half-float variants for the f16 inference path (BASELINE configs[4]): 8 channels per lane (one 16-byte access),
// arithmetic in f32.  MAX pooling needs no argmax here (inference only).
// ---------------------------------------------------------------------------------------------
typedef _Float16 h8_t __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(256) void maxpool_f16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int N, int H, int W, int C,
                                                          int x_cstride, int k, int stride, int pad, int OH, int OW, int y_cstride,
                                                          int y_coffset) {
    const int cg = C / 8;
    const long long total = (long long)N * OH * OW * cg;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t % cg);
        long long pix = t / cg;
        const int ox = (int)(pix % OW);
        pix /= OW;
        const int oy = (int)(pix % OH);
        const int n = (int)(pix / OH);
        int hs = oy * stride - pad, ws = ox * stride - pad;
        const int he = min(hs + k, H), we = min(ws + k, W);
        hs = max(hs, 0);
        ws = max(ws, 0);
        const _Float16* xb = x + (size_t)n * H * W * x_cstride + g * 8;
        h8_t m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (_Float16)-65504.f;
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const h8_t v = *(const h8_t*)(xb + ((size_t)iy * W + ix) * x_cstride);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        *(h8_t*)(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + y_coffset + g * 8) = m;
    }
}

// The same, one lane = 8 channels of a 2 x 2 QUAD of output pixels, a workgroup = 7 x 8 quads (14 x 16 pixels) x 64 channels.
// A 3x3 / stride-1 pooling in output-per-lane form issues nine 16-byte loads per output and is bound by the L1's 64 B/clk
// (batch-32 inception poolings: 22-40 us each, 2.3 TB/s of algorithmic traffic); a quad shares its (K + S)^2 inputs - 16 loads
// for four outputs at stride 1, 25 at stride 2 - and the patch keeps the overlap between quads inside one CU's L1.
template <int K, int S>
__global__ __launch_bounds__(448) void maxpool_f16_quad_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int H, int W, int C,
                                                               int x_cstride, int pad, int OH, int OW, int y_cstride, int y_coffset, int cgroups) {
    const int cg64 = (int)blockIdx.x % cgroups, tx = (int)blockIdx.x / cgroups;
    const int c = cg64 * 64 + ((int)threadIdx.x & 7) * 8;
    const int q = (int)threadIdx.x >> 3;                       // quad 0..55: row q >> 3, column q & 7
    const int oy = ((int)blockIdx.y * 7 + (q >> 3)) * 2, ox = (tx * 8 + (q & 7)) * 2, n = (int)blockIdx.z;
    if (c >= C || oy >= OH || ox >= OW) return;
    const _Float16* xb = x + (size_t)n * H * W * x_cstride + c;
    const int iy0 = oy * S - pad, ix0 = ox * S - pad;
    h8_t m[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 8; ++e) m[a][b][e] = (_Float16)-65504.f;
#pragma unroll
    for (int dy = 0; dy < K + S; ++dy) {
        const int iy = iy0 + dy;
        if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
        for (int dx = 0; dx < K + S; ++dx) {
            const int ix = ix0 + dx;
            if ((unsigned)ix >= (unsigned)W) continue;
            const h8_t v = *(const h8_t*)(xb + ((size_t)iy * W + ix) * x_cstride);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    if (dy >= a * S && dy < a * S + K && dx >= b * S && dx < b * S + K)      // (compile-time: which outputs see this input)
#pragma unroll
                        for (int e = 0; e < 8; ++e) m[a][b][e] = v[e] > m[a][b][e] ? v[e] : m[a][b][e];
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
            if (oy + a < OH && ox + b < OW)
                *(h8_t*)(y + ((size_t)(n * OH + oy + a) * OW + ox + b) * y_cstride + y_coffset + c) = m[a][b];
}

__device__ __forceinline__ h8_t max8(const h8_t a, const h8_t b);
// 3x3 / stride 1 / pad 1 MAX pooling (the nine inception poolings of models/deploy.prototxt:325-336 etc.) through an LDS patch.
// The quad kernel above issues 16 global loads per four outputs and runs the batch-32 poolings at ~3 TB/s of algorithmic traffic;
// here a workgroup owns TH x TW output pixels x 64 channels: the (TH + 2) x (TW + 2) input pixels under them go to LDS ONCE by
// LDS-DMA (a pixel's 64 channels are one 128-byte row; pixels outside the image load the nearest pixel INSIDE it - the maximum
// over a window with replicated edges is the maximum over the clipped window, exactly, for any values), and a lane (column x,
// 8 channels) forms the maximum separably: per input row the maximum of three neighbours, then a sliding maximum of three rows -
// 3.4 LDS reads per output instead of 9 global ones, every input byte fetched 1.3x (the halo, from L2) instead of 4x.
constexpr int kP3MaxTH = 8, kP3MaxTW = 32;
constexpr int kP3LdsBytes = ((kP3MaxTH + 2) * (kP3MaxTW + 2) * 128 + 1023) / 1024 * 1024;      // (whole 1 KiB pieces: the last one may be partly padding)

__global__ __launch_bounds__(256) void maxpool3s1_f16_lds_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int H, int W, int C, int x_cstride,
                                                                 int y_cstride, int y_coffset, int cgroups, int TH, int TW) {
    extern __shared__ __attribute__((aligned(16))) char patch[];      // (TH + 2) x (TW + 2) pixels in whole 1 KiB pieces: sized by the launch
    typedef const void __attribute__((address_space(1))) * gptr;
    typedef void __attribute__((address_space(3))) * lptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg64 = (int)blockIdx.x % cgroups, tx = (int)blockIdx.x / cgroups;
    const int y0 = (int)blockIdx.y * TH, x0 = tx * TW, n = (int)blockIdx.z;
    const int PWp = TW + 2, npix = (TH + 2) * PWp;
    const int c = cg64 * 64 + (lane & 7) * 8;                  // this lane's 8 channels
    const int cc = c < C ? c : C - 8;                           // (lanes past C fetch valid bytes; they store nothing)
    const _Float16* xn = x + (size_t)n * H * W * x_cstride + cc;
    // ---- stage: 8 pixels (1 KiB) per wave-instruction, lane -> pixel 8 i + lane / 8, segment lane % 8
    for (int i = wave; i * 8 < npix; i += 4) {
        int p = i * 8 + (lane >> 3);
        p = p < npix ? p : npix - 1;
        const int pr = p / PWp, pc = p - pr * PWp;
        int iy = y0 - 1 + pr, ix = x0 - 1 + pc;
        iy = iy < 0 ? 0 : iy >= H ? H - 1 : iy;
        ix = ix < 0 ? 0 : ix >= W ? W - 1 : ix;
        __builtin_amdgcn_global_load_lds((gptr)(xn + ((size_t)iy * W + ix) * x_cstride), (lptr)(patch + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- compute: lane (column xl, channel group) walks the patch rows
    const int xl = tid >> 3;
    if (xl >= TW || x0 + xl >= W || c >= C) return;
    const char* col = patch + (xl * 8 + (lane & 7)) * 16;      // patch[row 0][xl][segment]
    auto row_max = [&](const int r) {
        const h8_t a = *reinterpret_cast<const h8_t*>(col + (r * PWp) * 128);
        const h8_t b = *reinterpret_cast<const h8_t*>(col + (r * PWp + 1) * 128);
        const h8_t d = *reinterpret_cast<const h8_t*>(col + (r * PWp + 2) * 128);
        return max8(max8(a, b), d);
    };
    h8_t r0 = row_max(0), r1 = row_max(1);
    _Float16* yn = y + ((size_t)n * H * W + x0 + xl) * y_cstride + y_coffset + c;
    for (int oy = 0; oy < TH && y0 + oy < H; ++oy) {
        const h8_t r2 = row_max(oy + 2);
        *reinterpret_cast<h8_t*>(yn + (size_t)(y0 + oy) * W * y_cstride) = max8(max8(r0, r1), r2);
        r0 = r1;
        r1 = r2;
    }
}

// LRN across channels, local_size 5: the window of 8 channels lives in the 24 halves c-8..c+15 of the pixel
// LRN of the 8 channels in `c` given their neighbour groups (zeros outside the blob), f32 arithmetic, one rounding to half.
// Round 3: at batch 32 the two LRN layers are bound by the vector ALU, not by HBM (77 M elements at ~20 instructions each), so the
// arithmetic is trimmed: the window sums come from v_dot2_f32_f16 (exact products, f32 accumulation; pair sums P are shared between
// neighbouring windows: 27 instructions per 8 channels instead of 56), the scale is one fma, and B075 (beta == 0.75, every LRN layer
// of the reference's nets: models/deploy.prototxt:66-75,127-136) is a template argument - the general powf expansion used to sit in
// the middle of the unrolled loop (a 60 KB kernel, jumped over 64 times per workgroup).  A window sum differs from the plain
// left-to-right f32 sum in its last bit at most: far below the half ulp the result is rounded to.
#ifndef FCN_LRN_DOT2
#define FCN_LRN_DOT2 1
#endif
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
template <bool B075>
__device__ __forceinline__ h8_t lrn5_h8(const h8_t l, const h8_t c, const h8_t r, float alpha_over_n, float beta, float kk) {
    float S[8];      // sum of squares over channels e-2 .. e+2
#if FCN_LRN_DOT2
    const _Float16 z = (_Float16)0.f;
    const h2_t l3 = {l[6], l[7]}, c0 = {c[0], c[1]}, c1 = {c[2], c[3]}, c2 = {c[4], c[5]}, c3 = {c[6], c[7]}, r0 = {r[0], r[1]};
    auto lo = [&](const h2_t p) __attribute__((always_inline)) { return h2_t{p[0], z}; };
    auto hi = [&](const h2_t p) __attribute__((always_inline)) { return h2_t{z, p[1]}; };
    const float Pl = __builtin_amdgcn_fdot2(l3, l3, 0.f, false), P0 = __builtin_amdgcn_fdot2(c0, c0, 0.f, false),
                P1 = __builtin_amdgcn_fdot2(c1, c1, 0.f, false), P2 = __builtin_amdgcn_fdot2(c2, c2, 0.f, false),
                P3 = __builtin_amdgcn_fdot2(c3, c3, 0.f, false), Pr = __builtin_amdgcn_fdot2(r0, r0, 0.f, false);
    const float Tl0 = Pl + P0, T01 = P0 + P1, T12 = P1 + P2, T23 = P2 + P3, T3r = P3 + Pr;
    S[0] = __builtin_amdgcn_fdot2(c1, lo(c1), Tl0, false);      // l6 l7 c0 c1 | c2
    S[1] = __builtin_amdgcn_fdot2(l3, hi(l3), T01, false);      // l7 | c0 c1 c2 c3
    S[2] = __builtin_amdgcn_fdot2(c2, lo(c2), T01, false);      // c0 c1 c2 c3 | c4
    S[3] = __builtin_amdgcn_fdot2(c0, hi(c0), T12, false);      // c1 | c2 c3 c4 c5
    S[4] = __builtin_amdgcn_fdot2(c3, lo(c3), T12, false);      // c2 c3 c4 c5 | c6
    S[5] = __builtin_amdgcn_fdot2(c1, hi(c1), T23, false);      // c3 | c4 c5 c6 c7
    S[6] = __builtin_amdgcn_fdot2(r0, lo(r0), T23, false);      // c4 c5 c6 c7 | r0
    S[7] = __builtin_amdgcn_fdot2(c2, hi(c2), T3r, false);      // c5 | c6 c7 r0 r1
#else
    float q[12];      // squares of channels c-2 .. c+9
    q[0] = (float)l[6] * (float)l[6];
    q[1] = (float)l[7] * (float)l[7];
#pragma unroll
    for (int e = 0; e < 8; ++e) q[2 + e] = (float)c[e] * (float)c[e];
    q[10] = (float)r[0] * (float)r[0];
    q[11] = (float)r[1] * (float)r[1];
#pragma unroll
    for (int e = 0; e < 8; ++e) S[e] = q[e] + q[e + 1] + q[e + 2] + q[e + 3] + q[e + 4];
#endif
    h8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float sc = __builtin_fmaf(alpha_over_n, S[e], kk);
        float f;
        if (B075) {
            const float rs = __builtin_amdgcn_rsqf(sc);      // sc^-0.75 = sc^-0.5 * sqrt(sc^-0.5), both 1 ulp of f32
            f = rs * __builtin_amdgcn_sqrtf(rs);
        } else {
            f = powf(sc, -beta);
        }
        o[e] = (_Float16)((float)c[e] * f);
    }
    return o;
}

template <bool B075>
__global__ __launch_bounds__(256) void lrn5_f16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, long long pixels, int C,
                                                       int x_cstride, int y_cstride, float alpha_over_n, float beta, float kk) {
    const int cg = C / 8;
    const long long total = pixels * cg;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(t % cg);
        const long long pix = t / cg;
        const _Float16* xp = x + (size_t)pix * x_cstride + g * 8;
        const h8_t c = *(const h8_t*)xp;
        h8_t l, r;
#pragma unroll
        for (int e = 0; e < 8; ++e) l[e] = r[e] = (_Float16)0.f;
        if (g > 0) l = *(const h8_t*)(xp - 8);
        if (g + 1 < cg) r = *(const h8_t*)(xp + 8);
        *(h8_t*)(y + (size_t)pix * y_cstride + g * 8) = lrn5_h8<B075>(l, c, r, alpha_over_n, beta, kk);
    }
}

// MAX pooling and LRN of one half-float blob in a single pass (the half twin of maxpool_lrn5_kernel; bit-identical to the two
// stand-alone kernels: the maximum of halves is exact and every normalised value is rounded to a half before it is compared)
__device__ __forceinline__ h8_t max8(const h8_t a, const h8_t b) {
    h8_t m;
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = b[e] > a[e] ? b[e] : a[e];
    return m;
}

template <bool LRN_FIRST, bool B075>
__global__ __launch_bounds__(512) void maxpool_lrn5_f16_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int H, int W, int C,
                                                               int x_cstride, int k, int stride, int pad, int OH, int OW, int y_cstride,
                                                               int cgroups, float alpha_over_n, float beta, float kk) {
    const int cg64 = (int)blockIdx.x % cgroups, tx = (int)blockIdx.x / cgroups;
    const int g = cg64 * 8 + ((int)threadIdx.x & 7), cg = C / 8;
    const int pp = (int)threadIdx.x >> 3;
    const int oy = (int)blockIdx.y * 8 + (pp >> 3), ox = tx * 8 + (pp & 7), n = (int)blockIdx.z;
    if (g >= cg || oy >= OH || ox >= OW) return;
    int hs = oy * stride - pad, ws = ox * stride - pad;
    const int he = min(hs + k, H), we = min(ws + k, W);
    hs = max(hs, 0);
    ws = max(ws, 0);
    const _Float16* xb = x + (size_t)n * H * W * x_cstride + g * 8;
    const bool hl = g > 0, hr = g + 1 < cg;
    h8_t zero, low;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        zero[e] = (_Float16)0.f;
        low[e] = (_Float16)-65504.f;
    }
    h8_t m = low, ml = hl ? low : zero, mr = hr ? low : zero;
    for (int iy = hs; iy < he; ++iy)
        for (int ix = ws; ix < we; ++ix) {
            const _Float16* xp = xb + ((size_t)iy * W + ix) * x_cstride;
            const h8_t c = *(const h8_t*)xp;
            const h8_t l = hl ? *(const h8_t*)(xp - 8) : zero;
            const h8_t r = hr ? *(const h8_t*)(xp + 8) : zero;
            if (LRN_FIRST) {
                m = max8(m, lrn5_h8<B075>(l, c, r, alpha_over_n, beta, kk));
            } else {
                m = max8(m, c);
                if (hl) ml = max8(ml, l);
                if (hr) mr = max8(mr, r);
            }
        }
    if (!LRN_FIRST) m = lrn5_h8<B075>(ml, m, mr, alpha_over_n, beta, kk);
    *(h8_t*)(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + g * 8) = m;
}

// 3x3 / stride 2 MAX pooling and LRN (5 channels) of one half-float blob through an LDS patch, either order (pool1 -> norm1 and
// norm2 -> pool2 of models/deploy.prototxt:54-75,137-158 at batch 32).  maxpool_lrn5_f16_kernel above recomputes the normalisation
// for every window element (9 / 4 times per input) and was slower than the two stand-alone launches once the blobs are tens of MB;
// here a workgroup owns 4 x TW outputs x ALL channels: the 9 x (2 TW + 1) input pixels under them go to LDS once by LDS-DMA
// (pixels outside the image - ceil-mode windows that hang over the edge - are replaced by the nearest pixel inside: same maximum),
// LRN runs once per pixel on the patch (LRN first) or on the pooled tile (pool first), and the blob in between never exists.
// Same lrn5_h8 and exact maxima as the stand-alone kernels; equal to them bit for bit except at float32 values within ~1e-7 of a
// half-way point between two halves (tests/test_gpu_f16.py: 2e-5 of the elements, one f16 ulp).
// Measured at batch 32: pool1 + norm1 73 -> 69 us, norm2 + pool2 105 -> 92 us: the LRN arithmetic (two hardware square roots per
// element) is what bounds both forms - the fused one saves the blob's round trip but serialises stage / normalise / pool per workgroup.
constexpr int kPL_TH = 4;      // output rows per workgroup (the default; FCN_PL_TH / FCN_PL_NT: experiments)
constexpr int kPLLdsBytes = 64 * 1024;      // (dynamic LDS: what a launch may ask for without a function attribute)

template <bool LRN_FIRST, bool B075>
__global__ __launch_bounds__(512) void pool_lrn5_f16_lds_kernel(const _Float16* __restrict__ x, _Float16* __restrict__ y, int H, int W, int C, int x_cstride,
                                                                int OH, int OW, int y_cstride, int TW, int TH, float alpha_over_n, float beta, float kk) {
    // (sized by the launch: patch + pooled tile rounded up to the 1 KiB staging pieces - a fixed 80 KiB held a CU to two workgroups,
    //  and a workgroup stages, waits, normalises and pools one step after the other: what hides the wait is the workgroups beside it)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const void __attribute__((address_space(1))) * gptr;
    typedef void __attribute__((address_space(3))) * lptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NT = (int)blockDim.x, nwaves = NT >> 6;      // 256 or 512 threads
    const int segs = C / 8, pitch = C * 2;                        // 16-byte segments / bytes per pixel
    const int oy0 = (int)blockIdx.y * TH, ox0 = (int)blockIdx.x * TW, n = (int)blockIdx.z;
    const int PWp = 2 * TW + 1, PH = 2 * TH + 1, npix = PH * PWp, nitems = npix * segs;
    const _Float16* xn = x + (size_t)n * H * W * x_cstride;
    // ---- stage the patch: item g = (pixel, segment) in patch order, 64 items (1 KiB) per wave-instruction
    for (int i = wave; i * 64 < nitems; i += nwaves) {
        int g = i * 64 + lane;
        g = g < nitems ? g : nitems - 1;
        const int p = g / segs, sg = g - p * segs;
        const int pr = p / PWp, pc = p - pr * PWp;
        int iy = 2 * oy0 + pr, ix = 2 * ox0 + pc;
        iy = iy >= H ? H - 1 : iy;
        ix = ix >= W ? W - 1 : ix;
        __builtin_amdgcn_global_load_lds((gptr)(xn + ((size_t)iy * W + ix) * x_cstride + sg * 8), (lptr)(lds + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    h8_t zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.f;
    auto lrn_at = [&](const char* base, const int sg) {        // LRN of segment sg of the pixel at `base` (neighbour segments of the same pixel)
        const h8_t c = *reinterpret_cast<const h8_t*>(base + sg * 16);
        const h8_t l = sg > 0 ? *reinterpret_cast<const h8_t*>(base + sg * 16 - 16) : zero;
        const h8_t r = sg + 1 < segs ? *reinterpret_cast<const h8_t*>(base + sg * 16 + 16) : zero;
        return lrn5_h8<B075>(l, c, r, alpha_over_n, beta, kk);
    };
    if (LRN_FIRST) {      // normalise the patch in place: all results first (they read their neighbours' raw values), then all writes
        constexpr int MAXI = 8;
        h8_t res[MAXI];
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const int g = tid + NT * j;
            if (g < nitems) res[j] = lrn_at(lds + (g / segs) * pitch, g % segs);
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < MAXI; ++j) {
            const int g = tid + NT * j;
            if (g < nitems) *reinterpret_cast<h8_t*>(lds + (size_t)g * 16) = res[j];
        }
        __syncthreads();
    }
    // ---- pool: item = (output pixel of the tile, segment)
    const int nout = TH * TW * segs;
    char* const pooled = lds + (size_t)npix * pitch;              // (pool first: the pooled tile, normalised in a second step)
    for (int g = tid; g < nout; g += NT) {
        const int op = g / segs, sg = g - op * segs;
        const int oyl = op / TW, oxl = op - oyl * TW;
        const char* w0 = lds + ((2 * oyl) * PWp + 2 * oxl) * pitch + sg * 16;
        h8_t m = *reinterpret_cast<const h8_t*>(w0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (dy || dx) m = max8(m, *reinterpret_cast<const h8_t*>(w0 + (dy * PWp + dx) * pitch));
        if (LRN_FIRST) {
            const int oy = oy0 + oyl, ox = ox0 + oxl;
            if (oy < OH && ox < OW) *reinterpret_cast<h8_t*>(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + sg * 8) = m;
        } else {
            *reinterpret_cast<h8_t*>(pooled + (size_t)g * 16) = m;
        }
    }
    if (!LRN_FIRST) {
        __syncthreads();
        for (int g = tid; g < nout; g += NT) {
            const int op = g / segs, sg = g - op * segs;
            const int oy = oy0 + op / TW, ox = ox0 + op % TW;
            if (oy < OH && ox < OW)
                *reinterpret_cast<h8_t*>(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + sg * 8) = lrn_at(pooled + (size_t)op * pitch, sg);
        }
    }
}

// pool1/3x3_s2 -> pool1/norm1 -> conv2/3x3_reduce (+ ReLU) of models/deploy.prototxt:54-104 at batch 32, half floats, ONE launch: the pool-first
// form of pool_lrn5_f16_lds_kernel for 64 channels (a workgroup owns 4 x 16 output pixels), whose last step writes the normalised tile
// to LDS instead of HBM and multiplies it by the 64 x 64 filter bank with v_mfma_f32_16x16x16_f16 as out^T = W . act^T (A = 16 filters x
// 16 k, B = 16 k x 16 pixels; the k index of (step, lane group, element) is 16 * group + 4 * step + element, so a lane's filter values are one
// 32-byte run loaded at the top of the kernel and its pixel values two ds_read_b128; D's register r of lane (pixel j, group h) is output
// channel 16 * tile + 4 h + r): bias, ReLU, one rounding to half, an 8-byte store per lane.  The convolution as a launch of its own was
// 29 us of the 1.39 ms forward (a 51 MB blob written and read back); pooling and LRN are the stand-alone kernel's arithmetic.
template <bool B075>
__global__ __launch_bounds__(512) void pool_lrn5_conv1x1_f16_lds_kernel(const _Float16* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
                                                                        _Float16* __restrict__ y, int H, int W, int x_cstride, int OH, int OW, int y_cstride,
                                                                        int y_coffset, int relu, float alpha_over_n, float beta, float kk) {
    constexpr int C = 64, TW = 16, TH = 4, NT = 512, segs = C / 8, pitch = C * 2;
    constexpr int PWp = 2 * TW + 1, PH = 2 * TH + 1, npix = PH * PWp, nitems = npix * segs;      // 9 x 33 pixels, 2376 staging items
    constexpr int apitch = pitch + 16;                                                            // normalised tile: 144 bytes per pixel (bank spread)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef const void __attribute__((address_space(1))) * gptr;
    typedef void __attribute__((address_space(3))) * lptr;
    typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
    typedef float f4_t __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int oy0 = (int)blockIdx.y * TH, ox0 = (int)blockIdx.x * TW, n = (int)blockIdx.z;
    const _Float16* xn = x + (size_t)n * H * W * x_cstride;
    // wave -> output-channel tile ct = wave & 3; lane (i = lane & 15, h = lane >> 4) holds W[16 ct + i][16 h .. 16 h + 15]
    const int ct = wave & 3, fi = lane & 15, fh = lane >> 4;
    for (int i = wave; i * 64 < nitems; i += NT / 64) {
        int g = i * 64 + lane;
        g = g < nitems ? g : nitems - 1;
        const int p = g / segs, sg = g - p * segs;
        const int pr = p / PWp, pc = p - pr * PWp;
        int iy = 2 * oy0 + pr, ix = 2 * ox0 + pc;
        iy = iy >= H ? H - 1 : iy;
        ix = ix >= W ? W - 1 : ix;
        __builtin_amdgcn_global_load_lds((gptr)(xn + ((size_t)iy * W + ix) * x_cstride + sg * 8), (lptr)(lds + i * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (the filter fragments are needed last: their latency hides behind the pooling and the normalisation)
    const h8_t wf0 = *reinterpret_cast<const h8_t*>(w + (size_t)(16 * ct + fi) * C + 16 * fh);
    const h8_t wf1 = *reinterpret_cast<const h8_t*>(w + (size_t)(16 * ct + fi) * C + 16 * fh + 8);
    const f4_t bv = bias ? *reinterpret_cast<const f4_t*>(bias + 16 * ct + 4 * fh) : f4_t{0.f, 0.f, 0.f, 0.f};
    h8_t zero;
#pragma unroll
    for (int e = 0; e < 8; ++e) zero[e] = (_Float16)0.f;
    constexpr int nout = TH * TW * segs;                          // 512 items: one per thread
    constexpr int staged = (nitems + 63) / 64 * 1024;
    char* const pooled = lds + staged;                             // [64 pixels][128 bytes]
    char* const act = pooled + TH * TW * pitch;                    // [64 pixels][144 bytes]
    char* const outt = act + TH * TW * apitch;                     // the output tile, same shape
    {
        const int op = tid / segs, sg = tid - op * segs;
        const int oyl = op / TW, oxl = op - oyl * TW;
        const char* w0 = lds + ((2 * oyl) * PWp + 2 * oxl) * pitch + sg * 16;
        h8_t m = *reinterpret_cast<const h8_t*>(w0);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
                if (dy || dx) m = max8(m, *reinterpret_cast<const h8_t*>(w0 + (dy * PWp + dx) * pitch));
        *reinterpret_cast<h8_t*>(pooled + (size_t)tid * 16) = m;
        __syncthreads();
        const char* base = pooled + (size_t)op * pitch;
        const h8_t c = *reinterpret_cast<const h8_t*>(base + sg * 16);
        const h8_t l = sg > 0 ? *reinterpret_cast<const h8_t*>(base + sg * 16 - 16) : zero;
        const h8_t r = sg + 1 < segs ? *reinterpret_cast<const h8_t*>(base + sg * 16 + 16) : zero;
        *reinterpret_cast<h8_t*>(act + (size_t)op * apitch + sg * 16) = lrn5_h8<B075>(l, c, r, alpha_over_n, beta, kk);
        static_assert(nout == NT, "one pooling / LRN item per thread");
    }
    __syncthreads();
    // out^T = W . act^T: wave -> channel tile ct and the pixel tiles 2 (wave >> 2), 2 (wave >> 2) + 1
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int pt = 2 * (wave >> 2) + t;
        const char* ap = act + (size_t)(16 * pt + fi) * apitch + 32 * fh;
        const h8_t a0 = *reinterpret_cast<const h8_t*>(ap), a1 = *reinterpret_cast<const h8_t*>(ap + 16);
        f4_t acc = {0.f, 0.f, 0.f, 0.f};
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4_t{wf0[0], wf0[1], wf0[2], wf0[3]}, h4_t{a0[0], a0[1], a0[2], a0[3]}, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4_t{wf0[4], wf0[5], wf0[6], wf0[7]}, h4_t{a0[4], a0[5], a0[6], a0[7]}, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4_t{wf1[0], wf1[1], wf1[2], wf1[3]}, h4_t{a1[0], a1[1], a1[2], a1[3]}, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x16f16(h4_t{wf1[4], wf1[5], wf1[6], wf1[7]}, h4_t{a1[4], a1[5], a1[6], a1[7]}, acc, 0, 0, 0);
        const int op = 16 * pt + fi;
        h4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e] + bv[e];
            if (relu) v = fmaxf(v, 0.f);
            o[e] = (_Float16)v;
        }
        // (through LDS: an 8-byte store per lane would write 32 bytes to each of 16 pixels per instruction - the first-layer kernel's
        //  experiment with such stores cost it 15 %; the tile leaves as whole 128-byte pixel rows instead)
        *reinterpret_cast<h4_t*>(outt + (size_t)op * apitch + (16 * ct + 4 * fh) * 2) = o;
    }
    __syncthreads();
    {
        const int op = tid >> 3, sg = tid & 7;
        const int oy = oy0 + op / TW, ox = ox0 + op % TW;
        if (oy < OH && ox < OW)
            *reinterpret_cast<h8_t*>(y + ((size_t)(n * OH + oy) * OW + ox) * y_cstride + y_coffset + sg * 8) = *reinterpret_cast<const h8_t*>(outt + (size_t)op * apitch + sg * 16);
    }
}

extern "C" {

int fcn_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int H, int W, int dst_cstride, int dst_coffset, float shift,
                         fcn_stream_t s) {
    FCN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, FCN_E_ARG, "nchw_to_nhwc: bad args");
    FCN_REQUIRE(dst_coffset >= 0 && dst_cstride >= dst_coffset + C, FCN_E_ARG, "nchw_to_nhwc: slice exceeds dst_cstride");
    FCN_REQUIRE(N <= 65535, FCN_E_UNSUPPORTED, "nchw_to_nhwc: batch too large");
    const int HW = H * W;
    if (C <= 4 && dst_cstride == 4 && dst_coffset == 0 && ((uintptr_t)dst & 15) == 0) {      // the image: a lane per pixel
        hipLaunchKernelGGL(nchw_to_nhwc4_kernel, dim3(cdiv(HW, 256) < 4096 ? cdiv(HW, 256) : 4096, N), dim3(256), 0, as_stream(s), src, dst, C, HW, shift);
        FCN_LAUNCH_CHECK("nchw_to_nhwc4");
        return 0;
    }
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, as_stream(s), src, dst, C, HW, dst_cstride, dst_coffset, shift);
    FCN_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}

int fcn_nhwc_to_nchw_multi_f32(const fcn_layout_desc* h_descs, int n, fcn_stream_t s) {
    FCN_REQUIRE(h_descs && n > 0 && n <= kMaxLayoutBlobs, FCN_E_ARG, "nhwc_to_nchw_multi: 1..%d blobs", kMaxLayoutBlobs);
    LayoutArgs a;
    a.n = n;
    long long end = 0;
    for (int i = 0; i < kMaxLayoutBlobs; ++i) {
        const fcn_layout_desc& d = h_descs[i < n ? i : n - 1];
        if (i < n) {
            FCN_REQUIRE(d.src && d.dst && d.N > 0 && d.C > 0 && d.H > 0 && d.W > 0, FCN_E_ARG, "nhwc_to_nchw_multi: bad blob %d", i);
            FCN_REQUIRE(d.src_coffset >= 0 && d.src_cstride >= d.src_coffset + d.C, FCN_E_ARG, "nhwc_to_nchw_multi: slice of blob %d exceeds its stride", i);
            end += (long long)d.N * d.C * d.H * d.W;
            FCN_REQUIRE(end < (1ll << 30), FCN_E_UNSUPPORTED, "nhwc_to_nchw_multi: meant for small blobs");
        }
        LayoutBlob& b = a.b[i];
        b.src = d.src; b.dst = d.dst; b.C = d.C; b.HW = d.H * d.W; b.cstride = d.src_cstride; b.coffset = d.src_coffset;
        b.count = i < n ? d.N * d.C * d.H * d.W : 0;
        b.end = (int)end;
    }
    const int blocks = cdiv(end, 256) < 2048 ? cdiv(end, 256) : 2048;
    hipLaunchKernelGGL(nhwc_to_nchw_multi_kernel, dim3(blocks), dim3(256), 0, as_stream(s), a);
    FCN_LAUNCH_CHECK("nhwc_to_nchw_multi");
    return 0;
}

int fcn_nchw_f32_to_nhwc_f16(const float* src, void* dst, int N, int C, int H, int W, int dst_cstride, int dst_coffset, float shift,
                             fcn_stream_t s) {
    FCN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, FCN_E_ARG, "nchw_to_nhwc_f16: bad args");
    FCN_REQUIRE(dst_coffset >= 0 && dst_cstride >= dst_coffset + C, FCN_E_ARG, "nchw_to_nhwc_f16: slice exceeds dst_cstride");
    FCN_REQUIRE(N <= 65535, FCN_E_UNSUPPORTED, "nchw_to_nhwc_f16: batch too large");
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<_Float16>, grid, dim3(256), 0, as_stream(s), src, reinterpret_cast<_Float16*>(dst), C, HW, dst_cstride,
                       dst_coffset, shift);
    FCN_LAUNCH_CHECK("nchw_to_nhwc_f16");
    return 0;
}

int fcn_nhwc_f16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, int src_cstride, int src_coffset, fcn_stream_t s) {
    FCN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, FCN_E_ARG, "nhwc_f16_to_nchw: bad args");
    FCN_REQUIRE(src_coffset >= 0 && src_cstride >= src_coffset + C, FCN_E_ARG, "nhwc_f16_to_nchw: slice exceeds src_cstride");
    FCN_REQUIRE(N <= 65535, FCN_E_UNSUPPORTED, "nhwc_f16_to_nchw: batch too large");
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<_Float16>, grid, dim3(256), 0, as_stream(s), reinterpret_cast<const _Float16*>(src), dst, C, HW,
                       src_cstride, src_coffset);
    FCN_LAUNCH_CHECK("nhwc_f16_to_nchw");
    return 0;
}

int fcn_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W, int src_cstride, int src_coffset, fcn_stream_t s) {
    FCN_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0, FCN_E_ARG, "nhwc_to_nchw: bad args");
    FCN_REQUIRE(src_coffset >= 0 && src_cstride >= src_coffset + C, FCN_E_ARG, "nhwc_to_nchw: slice exceeds src_cstride");
    FCN_REQUIRE(N <= 65535, FCN_E_UNSUPPORTED, "nhwc_to_nchw: batch too large");
    const int HW = H * W;
    dim3 grid(cdiv(HW, 32), cdiv(C, 32), N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, as_stream(s), src, dst, C, HW, src_cstride, src_coffset);
    FCN_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}

int fcn_maxpool_fwd_f32(const float* x, float* y, int32_t* idx, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                        int OH, int OW, int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0, FCN_E_ARG,
                "maxpool: bad args");
    FCN_REQUIRE(pad < k, FCN_E_ARG, "maxpool: pad must be smaller than the kernel");
    FCN_REQUIRE((OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG,
                "maxpool: last window starts outside the image (OH/OW too large)");
    FCN_REQUIRE(x_cstride >= C && y_coffset >= 0 && y_cstride >= y_coffset + C, FCN_E_ARG, "maxpool: channel slice out of range");
    const bool vec = C % 4 == 0 && x_cstride % 4 == 0 && y_cstride % 4 == 0 && y_coffset % 4 == 0 && aligned16(x) && aligned16(y) &&
                     (!idx || aligned16(idx));
    const long long work = (long long)N * OH * OW * (vec ? C / 4 : C);
    const int grid = stream_grid(work, 256);
    hipStream_t st = as_stream(s);
#define FCN_POOL_ARGS x, y, idx, N, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride, y_coffset
    if (vec && idx) hipLaunchKernelGGL((maxpool_kernel<true, true>), dim3(grid), dim3(256), 0, st, FCN_POOL_ARGS);
    else if (vec) hipLaunchKernelGGL((maxpool_kernel<true, false>), dim3(grid), dim3(256), 0, st, FCN_POOL_ARGS);
    else if (idx) hipLaunchKernelGGL((maxpool_kernel<false, true>), dim3(grid), dim3(256), 0, st, FCN_POOL_ARGS);
    else hipLaunchKernelGGL((maxpool_kernel<false, false>), dim3(grid), dim3(256), 0, st, FCN_POOL_ARGS);
#undef FCN_POOL_ARGS
    FCN_LAUNCH_CHECK("maxpool");
    return 0;
}

int fcn_avepool_fwd_f32(const float* x, float* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                        int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0, FCN_E_ARG,
                "avepool: bad args");
    FCN_REQUIRE((OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG, "avepool: OH/OW too large");
    FCN_REQUIRE(x_cstride >= C && y_coffset >= 0 && y_cstride >= y_coffset + C, FCN_E_ARG, "avepool: channel slice out of range");
    const long long work = (long long)N * OH * OW * C;
    hipLaunchKernelGGL(avepool_kernel, dim3(stream_grid(work, 256)), dim3(256), 0, as_stream(s), x, y, N, H, W, C, x_cstride, k, stride, pad,
                       OH, OW, y_cstride, y_coffset);
    FCN_LAUNCH_CHECK("avepool");
    return 0;
}

int fcn_lrn_fwd_f32(const float* x, float* y, float* scale, int pixels, int C, int x_cstride, int y_cstride, int local_size, float alpha,
                    float beta, float k, fcn_stream_t s) {
    FCN_REQUIRE(x && y && pixels > 0 && C > 0 && local_size > 0, FCN_E_ARG, "lrn: bad args");
    FCN_REQUIRE(x_cstride >= C && y_cstride >= C, FCN_E_ARG, "lrn: channel stride smaller than C");
    const float aon = alpha / (float)local_size;
    hipStream_t st = as_stream(s);
    const bool fast = local_size == 5 && C % 4 == 0 && x_cstride % 4 == 0 && y_cstride % 4 == 0 && aligned16(x) && aligned16(y) &&
                      (!scale || aligned16(scale));
    if (fast) {
        hipLaunchKernelGGL(lrn5_kernel, dim3(stream_grid((long long)pixels * (C / 4), 256)), dim3(256), 0, st, x, y, scale,
                           (long long)pixels, C, x_cstride, y_cstride, aon, beta, k);
    } else {
        hipLaunchKernelGGL(lrn_generic_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, st, x, y, scale,
                           (long long)pixels, C, x_cstride, y_cstride, local_size, aon, beta, k);
    }
    FCN_LAUNCH_CHECK("lrn");
    return 0;
}

int fcn_maxpool_lrn5_fwd_f32(const float* x, float* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                             int y_cstride, int lrn_first, float alpha, float beta, float lrn_k, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0, FCN_E_ARG,
                "maxpool_lrn5: bad args");
    FCN_REQUIRE(pad < k && (OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG, "maxpool_lrn5: OH/OW too large or pad >= kernel");
    FCN_REQUIRE(x_cstride >= C && y_cstride >= C, FCN_E_ARG, "maxpool_lrn5: channel stride smaller than C");
    FCN_REQUIRE(C % 4 == 0 && x_cstride % 4 == 0 && y_cstride % 4 == 0 && aligned16(x) && aligned16(y), FCN_E_UNSUPPORTED,
                "maxpool_lrn5: needs 16-byte channel groups (run the two layers separately)");
    const int cgroups = cdiv(C, 32);
    const long long gx = (long long)cgroups * cdiv(OW, 8);
    FCN_REQUIRE(gx < (1ll << 31) && cdiv(OH, 8) <= 65535 && N <= 65535, FCN_E_UNSUPPORTED, "maxpool_lrn5: grid too large");
    const dim3 grid((unsigned)gx, cdiv(OH, 8), N);
    const float aon = alpha / 5.f;
    if (lrn_first && k == 3 && stride == 2 && pad == 0) {
        // The LDS-patch form: OPT-IN ($FCN_LRN_POOL_LDS=1).  Repeated back to back it is faster than the single pass below (7.4 against 8.3 us
        // for conv2/norm2 -> pool2 at batch 1), but inside a forward pass - inputs cold, rocprofv3's kernel trace - it is SLOWER (11.2 against
        // 10.2 us): its workgroups stage, wait, normalise and pool one step after the other, the single pass has every load in flight at once.
        const char* lds_env = getenv("FCN_LRN_POOL_LDS");      // (read per call: the tests run both forms in one process)
        const bool lds_ok = lds_env && atoi(lds_env) == 1;
        static const int th_env = getenv("FCN_LP_TH") ? atoi(getenv("FCN_LP_TH")) : 0;
        // (a workgroup stages, normalises and pools one step after the other: what hides the waits is the workgroups beside it - patches of at
        //  most 40 KiB leave room for three or four per CU.  192 channels: one output row per workgroup 7.4 us, two rows (64 KiB, one workgroup
        //  per CU) 7.8, the single pass without LDS 8.3)
        int TH = th_env >= 1 && th_env <= 4 ? th_env : 4;
        while (TH > 1 && (long long)(2 * TH + 1) * 17 * C * 4 > (th_env ? 64 : 40) * 1024) TH >>= 1;
        const long long patch = (long long)(2 * TH + 1) * 17 * C * 4, items = patch / 16;
        if (lds_ok && TH >= 1 && patch <= 64 * 1024 && items <= 8 * 512 && cdiv(OH, TH) <= 65535 && (long long)H * W * C >= (1 << 18)) {
            const unsigned lds_bytes = (unsigned)((items + 63) / 64 * 1024);
            hipLaunchKernelGGL(lrn5_pool3s2_lds_kernel, dim3(cdiv(OW, 8), cdiv(OH, TH), N), dim3(512), lds_bytes, as_stream(s), x, y, H, W, C, x_cstride, OH, OW,
                               y_cstride, TH, aon, beta, lrn_k);
            FCN_LAUNCH_CHECK("lrn5_pool3s2_lds");
            return 0;
        }
    }
    if (lrn_first)
        hipLaunchKernelGGL(maxpool_lrn5_kernel<true>, grid, dim3(512), 0, as_stream(s), x, y, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride,
                           cgroups, aon, beta, lrn_k);
    else
        hipLaunchKernelGGL(maxpool_lrn5_kernel<false>, grid, dim3(512), 0, as_stream(s), x, y, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride,
                           cgroups, aon, beta, lrn_k);
    FCN_LAUNCH_CHECK("maxpool_lrn5");
    return 0;
}

int fcn_maxpool_lrn5_conv1x1_fwd_f32(const float* x, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                                     float alpha, float beta, float lrn_k, const float* w, const float* bias, int Cout, int relu, float* y,
                                     int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && w && N > 0 && H > 0 && W > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0 && y_coffset >= 0, FCN_E_ARG,
                "maxpool_lrn5_conv1x1: bad args");
    FCN_REQUIRE(pad < k && (OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG,
                "maxpool_lrn5_conv1x1: OH/OW too large or pad >= kernel");
    FCN_REQUIRE(x_cstride >= C && y_cstride >= y_coffset + Cout, FCN_E_ARG, "maxpool_lrn5_conv1x1: channel stride smaller than the channels");
    FCN_REQUIRE(C == kPLC_C && Cout == 64 && k == 3, FCN_E_UNSUPPORTED,
                "maxpool_lrn5_conv1x1: 3 x 3 windows, 64 -> 64 channels only (run the three layers separately)");
    FCN_REQUIRE(x_cstride % 4 == 0 && y_cstride % 4 == 0 && y_coffset % 4 == 0 && aligned16(x) && aligned16(y) && aligned16(w) &&
                (!bias || aligned16(bias)), FCN_E_UNSUPPORTED, "maxpool_lrn5_conv1x1: needs 16-byte channel groups");
    FCN_REQUIRE(cdiv(OH, 4) <= 65535 && N <= 65535, FCN_E_UNSUPPORTED, "maxpool_lrn5_conv1x1: grid too large");
    const dim3 grid(cdiv(OW, 8), cdiv(OH, 4), N);
    hipLaunchKernelGGL(pool3_lrn5_conv1x1_kernel, grid, dim3(256), 0, as_stream(s), x, w, bias, y, H, W, x_cstride, stride, pad, OH, OW, y_cstride,
                       y_coffset, relu, alpha / 5.f, beta, lrn_k);
    FCN_LAUNCH_CHECK("maxpool_lrn5_conv1x1");
    return 0;
}

int fcn_relu_fwd_f32(const float* x, float* y, size_t count, float negative_slope, fcn_stream_t s) {
    return launch_unary(OP_RELU, x, y, count, negative_slope, 0.f, 0.f, s);
}

int fcn_sigmoid_fwd_f32(const float* x, float* y, size_t count, fcn_stream_t s) { return launch_unary(OP_SIGMOID, x, y, count, 0.f, 0.f, 0.f, s); }

int fcn_power_fwd_f32(const float* x, float* y, size_t count, float power, float scale, float shift, fcn_stream_t s) {
    return launch_unary(OP_POWER, x, y, count, power, scale, shift, s);
}

int fcn_eltwise_fwd_f32(const float* a, const float* b, float* y, size_t count, int op, float ca, float cb, fcn_stream_t s) {
    FCN_REQUIRE(a && b && y, FCN_E_ARG, "eltwise: null");
    FCN_REQUIRE(op == FCN_ELT_PROD || op == FCN_ELT_SUM || op == FCN_ELT_MAX, FCN_E_ARG, "eltwise: bad op %d", op);
    if (count == 0) return 0;
    hipLaunchKernelGGL(eltwise_kernel, dim3(stream_grid((long long)count, 256)), dim3(256), 0, as_stream(s), a, b, y, count, op, ca, cb);
    FCN_LAUNCH_CHECK("eltwise");
    return 0;
}

int fcn_copy_channels_f32(const float* src, float* dst, int pixels, int C, int src_cstride, int src_coffset, int dst_cstride,
                          int dst_coffset, fcn_stream_t s) {
    FCN_REQUIRE(src && dst && pixels > 0 && C > 0, FCN_E_ARG, "copy_channels: bad args");
    FCN_REQUIRE(src_coffset >= 0 && dst_coffset >= 0 && src_cstride >= src_coffset + C && dst_cstride >= dst_coffset + C, FCN_E_ARG,
                "copy_channels: slice out of range");
    hipLaunchKernelGGL(copy_channels_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, as_stream(s), src, dst,
                       (long long)pixels, C, src_cstride, src_coffset, dst_cstride, dst_coffset);
    FCN_LAUNCH_CHECK("copy_channels");
    return 0;
}

int fcn_softmax_fwd_f32(const float* x, float* y, int pixels, int C, int x_cstride, int y_cstride, fcn_stream_t s) {
    FCN_REQUIRE(x && y && pixels > 0 && C > 0 && x_cstride >= C && y_cstride >= C, FCN_E_ARG, "softmax: bad args");
    hipLaunchKernelGGL(softmax_kernel, dim3(stream_grid(pixels, 256)), dim3(256), 0, as_stream(s), x, y, (long long)pixels, C, x_cstride,
                       y_cstride);
    FCN_LAUNCH_CHECK("softmax");
    return 0;
}

int fcn_deconv_depthwise_fwd_f32(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int C, int x_cstride,
                                 int k, int stride, int pad, int OH, int OW, int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0, FCN_E_ARG, "deconv: bad args");
    FCN_REQUIRE(OH == stride * (H - 1) + k - 2 * pad && OW == stride * (W - 1) + k - 2 * pad, FCN_E_ARG,
                "deconv: OH/OW do not match s(H-1)+k-2p");
    FCN_REQUIRE(x_cstride >= C && y_coffset >= 0 && y_cstride >= y_coffset + C, FCN_E_ARG, "deconv: channel slice out of range");
    hipLaunchKernelGGL(deconv_dw_kernel, dim3(stream_grid((long long)N * OH * OW * C, 256)), dim3(256), 0, as_stream(s), x, w, bias, y, N,
                       H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride, y_coffset);
    FCN_LAUNCH_CHECK("deconv_depthwise");
    return 0;
}


int fcn_maxpool_fwd_f16(const void* x, void* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                        int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && pad < k && OH > 0 && OW > 0, FCN_E_ARG,
                "maxpool_f16: bad args");
    FCN_REQUIRE((OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG, "maxpool_f16: last window starts outside the image");
    FCN_REQUIRE(C % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0 && y_coffset % 8 == 0 && x_cstride >= C && y_coffset >= 0 &&
                    y_cstride >= y_coffset + C && aligned16(x) && aligned16(y), FCN_E_ALIGN, "maxpool_f16: channels / strides must be multiples of 8");
    const int cgroups = cdiv(C, 64);
    const long long gx = (long long)cgroups * cdiv(OW, 16);
    const bool quad = k == 3 && (stride == 1 || stride == 2) && gx < (1ll << 31) && cdiv(OH, 14) <= 65535 && N <= 65535;
    const _Float16* xh = reinterpret_cast<const _Float16*>(x);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
    static const bool lds_pool = !(getenv("FCN_POOL_LDS") && atoi(getenv("FCN_POOL_LDS")) == 0);      // (experiments: the quad kernel)
    if (lds_pool && k == 3 && stride == 1 && pad == 1 && OH == H && OW == W && W >= 8 && H >= 4 && N <= 65535 && (long long)N * H * W * 8 >= (1 << 16)) {
        // tile extents that divide the image where they can (28 = 4 x 7 rows of 28 columns; 56 = 7 x 8 rows of 2 x 28 columns)
        const int TW = W % 32 == 0 ? 32 : W % 28 == 0 ? 28 : W % 24 == 0 ? 24 : W < 32 ? W : 32;
        const int TH = H % 8 == 0 ? 8 : H % 7 == 0 ? 7 : 8;
        const unsigned p3_bytes = (unsigned)(((TH + 2) * (TW + 2) * 128 + 1023) / 1024 * 1024);      // (<= kP3LdsBytes)
        hipLaunchKernelGGL(maxpool3s1_f16_lds_kernel, dim3((unsigned)(cgroups * cdiv(W, TW)), cdiv(H, TH), N), dim3(256), p3_bytes, as_stream(s), xh, yh, H, W,
                           C, x_cstride, y_cstride, y_coffset, cgroups, TH, TW);
        FCN_LAUNCH_CHECK("maxpool3s1_f16_lds");
        return 0;
    }
    if (quad && stride == 1)
        hipLaunchKernelGGL((maxpool_f16_quad_kernel<3, 1>), dim3((unsigned)gx, cdiv(OH, 14), N), dim3(448), 0, as_stream(s), xh, yh, H, W, C, x_cstride,
                           pad, OH, OW, y_cstride, y_coffset, cgroups);
    else if (quad)
        hipLaunchKernelGGL((maxpool_f16_quad_kernel<3, 2>), dim3((unsigned)gx, cdiv(OH, 14), N), dim3(448), 0, as_stream(s), xh, yh, H, W, C, x_cstride,
                           pad, OH, OW, y_cstride, y_coffset, cgroups);
    else
        hipLaunchKernelGGL(maxpool_f16_kernel, dim3(stream_grid((long long)N * OH * OW * (C / 8), 256)), dim3(256), 0, as_stream(s), xh, yh, N, H, W, C,
                           x_cstride, k, stride, pad, OH, OW, y_cstride, y_coffset);
    FCN_LAUNCH_CHECK("maxpool_f16");
    return 0;
}

int fcn_maxpool_lrn5_fwd_f16(const void* x, void* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                             int y_cstride, int lrn_first, float alpha, float beta, float lrn_k, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && OH > 0 && OW > 0, FCN_E_ARG,
                "maxpool_lrn5_f16: bad args");
    FCN_REQUIRE(pad < k && (OH - 1) * stride - pad < H && (OW - 1) * stride - pad < W, FCN_E_ARG, "maxpool_lrn5_f16: OH/OW too large or pad >= kernel");
    FCN_REQUIRE(C % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0 && x_cstride >= C && y_cstride >= C && aligned16(x) && aligned16(y), FCN_E_ALIGN,
                "maxpool_lrn5_f16: channels / strides must be multiples of 8");
    const int cgroups = cdiv(C, 64);
    const long long gx = (long long)cgroups * cdiv(OW, 8);
    FCN_REQUIRE(gx < (1ll << 31) && cdiv(OH, 8) <= 65535 && N <= 65535, FCN_E_UNSUPPORTED, "maxpool_lrn5_f16: grid too large");
    const dim3 grid((unsigned)gx, cdiv(OH, 8), N);
    const _Float16* xh = reinterpret_cast<const _Float16*>(x);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
    const float aon = alpha / 5.f;
    {   // large blobs, 3x3 / stride 2 without padding: the LDS-patch kernel (all channels of a pixel in one workgroup)
        static const bool lds_ok = !(getenv("FCN_POOL_LDS") && atoi(getenv("FCN_POOL_LDS")) == 0);
        // (192 channels: 7 columns make a 52 KiB patch - three workgroups per CU where 8 columns (58 KiB) leave room for two)
        const int TW = C <= 96 ? 16 : (OW % 7 == 0 && getenv("FCN_PL_TW8") == nullptr) ? 7 : 8;
        static const int th_env = getenv("FCN_PL_TH") ? atoi(getenv("FCN_PL_TH")) : 0, nt_env = getenv("FCN_PL_NT") ? atoi(getenv("FCN_PL_NT")) : 0;
        const int TH = th_env >= 1 && th_env <= 8 ? th_env : kPL_TH, NT = nt_env == 256 ? 256 : 512;
        const long long patch = (long long)(2 * TH + 1) * (2 * TW + 1) * C * 2 + (lrn_first ? 0 : (long long)TH * TW * C * 2);
        const long long items = (long long)(2 * TH + 1) * (2 * TW + 1) * (C / 8);
        if (lds_ok && k == 3 && stride == 2 && pad == 0 && patch + 1024 <= kPLLdsBytes && items <= 8 * NT && (long long)N * H * W * C >= (1 << 22) &&
            cdiv(OH, TH) <= 65535) {
            const dim3 g2(cdiv(OW, TW), cdiv(OH, TH), N);
            // staging writes whole 1 KiB pieces: the patch is rounded up to them, the pooled tile sits behind the patch's own bytes
            const long long staged = (items + 63) / 64 * 1024;
            const long long need = lrn_first ? staged : std::max(staged, (long long)(2 * TH + 1) * (2 * TW + 1) * C * 2 + (long long)TH * TW * C * 2);
            const unsigned lds_bytes = (unsigned)((need + 1023) / 1024 * 1024);
            const bool b075 = beta == 0.75f;
#define FCN_PL_LAUNCH(F, B)                                                                                                             \
    hipLaunchKernelGGL((pool_lrn5_f16_lds_kernel<F, B>), g2, dim3(NT), lds_bytes, as_stream(s), xh, yh, H, W, C, x_cstride, OH, OW, y_cstride, \
                       TW, TH, aon, beta, lrn_k)
            if (lrn_first && b075) FCN_PL_LAUNCH(true, true);
            else if (lrn_first) FCN_PL_LAUNCH(true, false);
            else if (b075) FCN_PL_LAUNCH(false, true);
            else FCN_PL_LAUNCH(false, false);
#undef FCN_PL_LAUNCH
            FCN_LAUNCH_CHECK("pool_lrn5_f16_lds");
            return 0;
        }
    }
#define FCN_ML_LAUNCH(F, B)                                                                                                                        \
    hipLaunchKernelGGL((maxpool_lrn5_f16_kernel<F, B>), grid, dim3(512), 0, as_stream(s), xh, yh, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride, \
                       cgroups, aon, beta, lrn_k)
    if (lrn_first && beta == 0.75f) FCN_ML_LAUNCH(true, true);
    else if (lrn_first) FCN_ML_LAUNCH(true, false);
    else if (beta == 0.75f) FCN_ML_LAUNCH(false, true);
    else FCN_ML_LAUNCH(false, false);
#undef FCN_ML_LAUNCH
    FCN_LAUNCH_CHECK("maxpool_lrn5_f16");
    return 0;
}

int fcn_maxpool_lrn5_conv1x1_fwd_f16(const void* x, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                                     float alpha, float beta, float lrn_k, const void* w, const float* bias, int Cout, int relu, void* y,
                                     int y_cstride, int y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && w && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && y_coffset >= 0, FCN_E_ARG, "maxpool_lrn5_conv1x1_f16: bad args");
    FCN_REQUIRE(C == 64 && Cout == 64 && k == 3 && stride == 2 && pad == 0, FCN_E_UNSUPPORTED,
                "maxpool_lrn5_conv1x1_f16: 3 x 3 / stride 2 windows without padding, 64 -> 64 channels only (run the three layers separately)");
    FCN_REQUIRE(2 * (OH - 1) < H && 2 * (OW - 1) < W, FCN_E_ARG, "maxpool_lrn5_conv1x1_f16: OH/OW too large");
    FCN_REQUIRE(x_cstride >= C && x_cstride % 8 == 0 && y_cstride >= y_coffset + Cout && y_cstride % 8 == 0 && y_coffset % 8 == 0 && aligned16(x) &&
                aligned16(w) && aligned16(y) && (!bias || aligned16(bias)), FCN_E_ALIGN, "maxpool_lrn5_conv1x1_f16: channel groups / alignment");
    FCN_REQUIRE(cdiv(OH, 4) <= 65535 && N <= 65535, FCN_E_UNSUPPORTED, "maxpool_lrn5_conv1x1_f16: grid too large");
    const dim3 grid(cdiv(OW, 16), cdiv(OH, 4), N);
    const unsigned lds_bytes = (9 * 33 * 8 + 63) / 64 * 1024 + 64 * 128 + 2 * 64 * 144;      // staged patch + pooled tile + normalised tile + output tile (64 KiB)
    const _Float16* xh = reinterpret_cast<const _Float16*>(x);
    const _Float16* wh = reinterpret_cast<const _Float16*>(w);
    _Float16* yh = reinterpret_cast<_Float16*>(y);
    if (beta == 0.75f)
        hipLaunchKernelGGL((pool_lrn5_conv1x1_f16_lds_kernel<true>), grid, dim3(512), lds_bytes, as_stream(s), xh, wh, bias, yh, H, W, x_cstride, OH, OW, y_cstride,
                           y_coffset, relu, alpha / 5.f, beta, lrn_k);
    else
        hipLaunchKernelGGL((pool_lrn5_conv1x1_f16_lds_kernel<false>), grid, dim3(512), lds_bytes, as_stream(s), xh, wh, bias, yh, H, W, x_cstride, OH, OW, y_cstride,
                           y_coffset, relu, alpha / 5.f, beta, lrn_k);
    FCN_LAUNCH_CHECK("maxpool_lrn5_conv1x1_f16");
    return 0;
}

int fcn_lrn_fwd_f16(const void* x, void* y, int pixels, int C, int x_cstride, int y_cstride, int local_size, float alpha, float beta, float k,
                    fcn_stream_t s) {
    FCN_REQUIRE(x && y && pixels > 0 && C > 0, FCN_E_ARG, "lrn_f16: bad args");
    FCN_REQUIRE(local_size == 5, FCN_E_UNSUPPORTED, "lrn_f16: local_size %d (the reference nets use 5)", local_size);
    FCN_REQUIRE(C % 8 == 0 && x_cstride % 8 == 0 && y_cstride % 8 == 0 && x_cstride >= C && y_cstride >= C && aligned16(x) && aligned16(y),
                FCN_E_ALIGN, "lrn_f16: channels / strides must be multiples of 8");
    const dim3 lgrid(stream_grid((long long)pixels * (C / 8), 256));
    if (beta == 0.75f)
        hipLaunchKernelGGL(lrn5_f16_kernel<true>, lgrid, dim3(256), 0, as_stream(s), reinterpret_cast<const _Float16*>(x), reinterpret_cast<_Float16*>(y),
                           (long long)pixels, C, x_cstride, y_cstride, alpha / (float)local_size, beta, k);
    else
        hipLaunchKernelGGL(lrn5_f16_kernel<false>, lgrid, dim3(256), 0, as_stream(s), reinterpret_cast<const _Float16*>(x), reinterpret_cast<_Float16*>(y),
                           (long long)pixels, C, x_cstride, y_cstride, alpha / (float)local_size, beta, k);
    FCN_LAUNCH_CHECK("lrn_f16");
    return 0;
}
}  // extern "C"
