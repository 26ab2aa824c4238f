// Training-scene synthesis on the device.
//
// Stands in for the pixel work of the reference's data layer:
// ArgumentationEngineMapping.argument (scripts/data_argumentation_layer/argumentation_engine.py:651-746: background crop
// + resize, per object flip / crop / optional rescale / masked paste in a per-pixel Python loop) and the whole-image flip of
// random_argumentation (:143-188).  The RANDOM DECISIONS stay on the host (fcn_object_detector_amd/data_layer.py replays
// the reference's draws); this kernel renders a decided scene: every object image / mask and the background are resident
// in HBM (288 GB: a dataset is uploaded once), a sample costs one 64-byte record per object and one launch.
//
// One lane per output pixel, gather form: background first, then the objects in paste order (later objects overwrite
// earlier ones, as in the reference).  Bilinear resampling reproduces the oracle renderer's float32 arithmetic bit for
// bit (oracle/scene_ref.py::resize_bilinear: half-pixel centres computed in double, weights in
// float32, round-half-even to uint8), so the kernel is checked bit-exactly against the oracle renderer.
#include "common.h"

using namespace fcn;

namespace {

struct Tap { int i0, i1; float f; };

// oracle/scene_ref.py::resize_bilinear coords(): source taps of destination index d for n_in -> n_out
__device__ __forceinline__ Tap tap(int d, int n_in, int n_out) {
    Tap t;
    if (n_in == n_out) {
        t.i0 = d; t.i1 = d < n_in - 1 ? d + 1 : n_in - 1; t.f = 0.f;
        return t;
    }
    const double fd = ((double)d + 0.5) * ((double)n_in / (double)n_out) - 0.5;
    const float f = (float)fd;
    const float fl = floorf(f);
    int s = (int)fl;
    float fr = f - fl;
    if (s < 0) { fr = 0.f; s = 0; }
    if (s >= n_in - 1) { fr = 0.f; s = n_in - 1; }
    t.i0 = s;
    t.i1 = s + 1 < n_in - 1 ? s + 1 : n_in - 1;
    t.f = fr;
    return t;
}

__device__ __forceinline__ float lerp2(float a00, float a01, float a10, float a11, float fx, float fy) {
    const float top = a00 * (1.f - fx) + a01 * fx;
    const float bot = a10 * (1.f - fx) + a11 * fx;
    return top * (1.f - fy) + bot * fy;
}

__device__ __forceinline__ unsigned char to_u8(float v) {
    const float r = rintf(v);
    return (unsigned char)(r < 0.f ? 0.f : (r > 255.f ? 255.f : r));
}

// coordinates in the FLIPPED source image -> the stored (unflipped) image.  cv.flip codes: 0 = around x, 1 = around y, -1 = both
__device__ __forceinline__ void unflip(int flip, int H, int W, int& x, int& y) {
    if (flip == 0 || flip == -1) y = H - 1 - y;
    if (flip == 1 || flip == -1) x = W - 1 - x;
}

__global__ __launch_bounds__(256) void compose_scene_kernel(const unsigned char* __restrict__ bg, int bg_h, int bg_w, int crop_x, int crop_y,
                                                            int crop_w, int crop_h, const fcn_scene_obj* __restrict__ objs, int nobj,
                                                            int final_flip, unsigned char* __restrict__ out_img,
                                                            unsigned char* __restrict__ out_mask, int H, int W, int vx, int vy,
                                                            int VH, int VW) {
    // the output is the VH x VW window at (vx, vy) of the (flipped) H x W scene: the zoom crop of random_argumentation is a
    // change of origin for a gather kernel, not a copy
    const int total = VH * VW;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int Y = t / VW + vy, X = t - (t / VW) * VW + vx;
        int x = X, y = Y;                       // position in the scene BEFORE the whole-image flip
        if (final_flip >= -1 && final_flip <= 1) unflip(final_flip, H, W, x, y);
        // background: bilinear resize of the crop to H x W
        const Tap tx = tap(x, crop_w, W), ty = tap(y, crop_h, H);
        const unsigned char* b0 = bg + ((size_t)(crop_y + ty.i0) * bg_w + crop_x) * 3;
        const unsigned char* b1 = bg + ((size_t)(crop_y + ty.i1) * bg_w + crop_x) * 3;
        unsigned char px[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            px[c] = to_u8(lerp2((float)b0[tx.i0 * 3 + c], (float)b0[tx.i1 * 3 + c], (float)b1[tx.i0 * 3 + c], (float)b1[tx.i1 * 3 + c], tx.f, ty.f));
        unsigned char lab = 0;
        for (int o = 0; o < nobj; ++o) {
            const fcn_scene_obj q = objs[o];
            const int dx = x - q.cx, dy = y - q.cy;
            if ((unsigned)dx >= (unsigned)q.out_w || (unsigned)dy >= (unsigned)q.out_h) continue;
            const Tap ox = tap(dx, q.roi_w, q.out_w), oy = tap(dy, q.roi_h, q.out_h);
            int xs[2] = {q.roi_x + ox.i0, q.roi_x + ox.i1}, ys[2] = {q.roi_y + oy.i0, q.roi_y + oy.i1};
            size_t off[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    int sx = xs[b], sy = ys[a];
                    unflip(q.flip, q.src_h, q.src_w, sx, sy);
                    off[a][b] = (size_t)sy * q.src_w + sx;
                }
            const float m = lerp2((float)q.mask[off[0][0]], (float)q.mask[off[0][1]], (float)q.mask[off[1][0]], (float)q.mask[off[1][1]], ox.f, oy.f);
            if (to_u8(m) == 0) continue;
#pragma unroll
            for (int c = 0; c < 3; ++c)
                px[c] = to_u8(lerp2((float)q.img[off[0][0] * 3 + c], (float)q.img[off[0][1] * 3 + c], (float)q.img[off[1][0] * 3 + c],
                                    (float)q.img[off[1][1] * 3 + c], ox.f, oy.f));
            lab = (unsigned char)q.label1;
        }
        if (out_img) {
            out_img[(size_t)t * 3 + 0] = px[0];
            out_img[(size_t)t * 3 + 1] = px[1];
            out_img[(size_t)t * 3 + 2] = px[2];
        }
        if (out_mask) out_mask[t] = lab;
    }
}

// class-id mask (h x w uint8) -> one float per pixel of an H x W label blob, nearest neighbour (src = floor(dst * scale))
__global__ __launch_bounds__(256) void mask_to_label_kernel(const unsigned char* __restrict__ mask, int h, int w, float* __restrict__ dst, int H,
                                                            int W, int dst_cstride) {
    const int total = H * W;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int Y = t / W, X = t - Y * W;
        int sy = (int)((double)Y * ((double)h / (double)H)), sx = (int)((double)X * ((double)w / (double)W));
        sy = sy < h - 1 ? sy : h - 1;
        sx = sx < w - 1 ? sx : w - 1;
        dst[(size_t)t * dst_cstride] = (float)mask[(size_t)sy * w + sx];
    }
}

}  // namespace

extern "C" {

int fcn_compose_scene_view_bgr8(const uint8_t* bg, int bg_h, int bg_w, int crop_x, int crop_y, int crop_w, int crop_h, const fcn_scene_obj* d_objs,
                                int nobj, int final_flip, uint8_t* out_img, uint8_t* out_mask, int H, int W, int view_x, int view_y, int view_w,
                                int view_h, fcn_stream_t s) {
    FCN_REQUIRE(bg && (out_img || out_mask) && bg_h > 0 && bg_w > 0 && H > 0 && W > 0 && nobj >= 0 && (nobj == 0 || d_objs), FCN_E_ARG,
                "compose_scene: bad args");
    FCN_REQUIRE(view_x >= 0 && view_y >= 0 && view_w > 0 && view_h > 0 && view_x + view_w <= W && view_y + view_h <= H, FCN_E_ARG,
                "compose_scene: view (%d,%d,%d,%d) outside the %dx%d scene", view_x, view_y, view_w, view_h, W, H);
    FCN_REQUIRE(crop_x >= 0 && crop_y >= 0 && crop_w > 0 && crop_h > 0 && crop_x + crop_w <= bg_w && crop_y + crop_h <= bg_h, FCN_E_ARG,
                "compose_scene: background crop (%d,%d,%d,%d) outside the %dx%d image", crop_x, crop_y, crop_w, crop_h, bg_w, bg_h);
    FCN_REQUIRE((long long)H * W < (1ll << 30), FCN_E_UNSUPPORTED, "compose_scene: scene too large");
    hipLaunchKernelGGL(compose_scene_kernel, dim3(stream_grid((long long)view_h * view_w, 256)), dim3(256), 0, as_stream(s), bg, bg_h, bg_w,
                       crop_x, crop_y, crop_w, crop_h, d_objs, nobj, final_flip, out_img, out_mask, H, W, view_x, view_y, view_h, view_w);
    FCN_LAUNCH_CHECK("compose_scene");
    return 0;
}

int fcn_compose_scene_bgr8(const uint8_t* bg, int bg_h, int bg_w, int crop_x, int crop_y, int crop_w, int crop_h, const fcn_scene_obj* d_objs,
                           int nobj, int final_flip, uint8_t* out_img, uint8_t* out_mask, int H, int W, fcn_stream_t s) {
    FCN_REQUIRE(out_img, FCN_E_ARG, "compose_scene: bad args");
    return fcn_compose_scene_view_bgr8(bg, bg_h, bg_w, crop_x, crop_y, crop_w, crop_h, d_objs, nobj, final_flip, out_img, out_mask, H, W, 0, 0, W,
                                       H, s);
}

int fcn_mask_to_label_f32(const uint8_t* mask, int h, int w, float* dst, int H, int W, int dst_cstride, fcn_stream_t s) {
    FCN_REQUIRE(mask && dst && h > 0 && w > 0 && H > 0 && W > 0 && dst_cstride >= 1, FCN_E_ARG, "mask_to_label: bad args");
    hipLaunchKernelGGL(mask_to_label_kernel, dim3(stream_grid((long long)H * W, 256)), dim3(256), 0, as_stream(s), mask, h, w, dst, H, W, dst_cstride);
    FCN_LAUNCH_CHECK("mask_to_label");
    return 0;
}

}  // extern "C"
