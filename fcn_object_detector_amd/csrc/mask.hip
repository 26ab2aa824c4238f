// What the node does with its score maps AFTER net.forward() in run_detector2 - integer / index work, bit-exact.
//
// Reference: scripts/fcn_object_detector.py:208-236 (threshold, x255, cv.resize to the window, uint8 cast, OR into the frame-sized
// pmap) and create_mask_labels :279-303 (cv.findContours(RETR_CCOMP, CHAIN_APPROX_SIMPLE), the contour with the largest
// cv.contourArea, cv.boundingRect).  OpenCV is not vendored by the reference; its published algorithms are restated in
// oracle/mask_ref.py, whose header also derives what the selection comes to: the bounding box of the 8-connected component whose
// OUTER border polygon (Suzuki border following through the pixel centres) has the largest positive area, the component found LAST
// in raster order among equals.  Four launches for all (window, class) maps of a frame:
//   1. score_mask_kernel   one lane per mask pixel: threshold, x255, bilinear resize with OpenCV's float coefficients and float
//                          products / sums (no contraction), truncating cast, atomicOr into pmap, label initialisation;
//   2. ccl_merge_kernel    8-connected components: lock-free union-find over the labels, roots = smallest pixel index of a
//                          component = the pixel cvFindContours starts that component's outer border at;
//   3. contour_kernel      one lane per ROOT follows its outer border (sequential by nature; components are independent) and offers
//                          (twice the polygon area, start index) to a 64-bit atomicMax: largest area, then latest start;
//   4. select_kernel       one lane per map follows the winner's border once more for its bounding rectangle.
#include "common.h"

using namespace fcn;

namespace {

struct MaskP {
    const float* score;      // NHWC score blob: N windows x H x W x cstride floats, classes at coffset ..
    int N, C, H, W, cstride, coffset;
    int w, h;                // window size in frame pixels = mask size
    int frame_h, frame_w;
    float thresh;
    double scale_x, scale_y; // 1 / (w / W), 1 / (h / H) as cv::resize computes them
    unsigned char* pmap;
    unsigned char* feat;     // [maps][h][w]
    int* labels;             // [maps][h * w]: pixel index of a foreground pixel's parent, -1 for background
    unsigned long long* keys;// [maps]
    int rect_xy[2 * 32];     // window origins (x, y) in the frame, at most 32 windows
};

__device__ __forceinline__ void src_coord(int d, double scale, int n_in, int* s_out, float* f_out) {
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_in - 1) { f = 0.f; s = n_in - 1; }
    *s_out = s;
    *f_out = f;
}

__global__ __launch_bounds__(256) void score_mask_kernel(const MaskP p) {
    const int m = blockIdx.y;                       // map = window * (C - 1) + (class - 1)
    const int n = m / (p.C - 1), c = 1 + m % (p.C - 1);
    const int px = p.w * p.h;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < px; i += gridDim.x * blockDim.x) {
        const int y = i / p.w, x = i - y * p.w;
        int sx, sy;
        float fx, fy;
        src_coord(x, p.scale_x, p.W, &sx, &fx);
        src_coord(y, p.scale_y, p.H, &sy, &fy);
        const int sx1 = min(sx + 1, p.W - 1), sy1 = min(sy + 1, p.H - 1);
        const float* base = p.score + (size_t)n * p.H * p.W * p.cstride + p.coffset + c;
        auto at = [&](int yy, int xx) {
            float v = base[((size_t)yy * p.W + xx) * p.cstride];
            v = v < p.thresh ? 0.f : v;                 // feature_maps[feature_maps < prob_thresh] = 0
            return __fmul_rn(v, 255.f);                 // fmaps[index] * 255 (float32)
        };
        const float a0 = __fsub_rn(1.f, fx), a1 = fx, b0 = __fsub_rn(1.f, fy), b1 = fy;
        const float r0 = __fadd_rn(__fmul_rn(at(sy, sx), a0), __fmul_rn(at(sy, sx1), a1));       // horizontal pass, the two source rows
        const float r1 = __fadd_rn(__fmul_rn(at(sy1, sx), a0), __fmul_rn(at(sy1, sx1), a1));
        const float v = __fadd_rn(__fmul_rn(r0, b0), __fmul_rn(r1, b1));                            // vertical pass
        // ndarray.astype(np.uint8): the C cast numpy performs on x86-64 - through a 32-bit integer, then the low byte
        const float vc = fminf(fmaxf(v, -2147483648.f), 2147483520.f);
        const unsigned char u = (unsigned char)((int)truncf(vc) & 0xFF);
        p.feat[(size_t)m * px + i] = u;
        p.labels[(size_t)m * px + i] = u ? i : -1;
        if (u) {
            const int fy_ = p.rect_xy[2 * n + 1] + y, fx_ = p.rect_xy[2 * n] + x;
            if ((unsigned)fy_ < (unsigned)p.frame_h && (unsigned)fx_ < (unsigned)p.frame_w) {
                const size_t off = (size_t)fy_ * p.frame_w + fx_;
                atomicOr(reinterpret_cast<unsigned*>(p.pmap + (off & ~(size_t)3)), (unsigned)u << (8 * (off & 3)));      // pmap |= feat
            }
        }
    }
}

// lock-free union-find in global memory, roots = smallest index (as detect.hip does in LDS); loads are agent-scope atomics: another
// compute unit's CAS must be seen
__device__ __forceinline__ int g_find(int* lab, int x) {
    while (true) {
        const int p = __hip_atomic_load(&lab[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        const int gp = __hip_atomic_load(&lab[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (gp == p) return p;
        __hip_atomic_store(&lab[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // path halving: gp is an ancestor too
        x = gp;
    }
}

__device__ __forceinline__ void g_union(int* lab, int a, int b) {
    while (true) {
        a = g_find(lab, a);
        b = g_find(lab, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }      // hook the larger root under the smaller
        if (atomicCAS(&lab[a], a, b) == a) return;
    }
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(const MaskP p) {
    const int m = blockIdx.y;
    const int px = p.w * p.h;
    const unsigned char* f = p.feat + (size_t)m * px;
    int* lab = p.labels + (size_t)m * px;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < px; i += gridDim.x * blockDim.x) {
        if (!f[i]) continue;
        const int y = i / p.w, x = i - y * p.w;
        if (x > 0 && f[i - 1]) g_union(lab, i, i - 1);                                   // W
        if (y > 0) {
            if (f[i - p.w]) g_union(lab, i, i - p.w);                                    // N
            if (x > 0 && f[i - p.w - 1]) g_union(lab, i, i - p.w - 1);                   // NW
            if (x + 1 < p.w && f[i - p.w + 1]) g_union(lab, i, i - p.w + 1);             // NE
        }
    }
}

// direction codes of cvFindContours (x right, y down): 0 = E, counter-clockwise on the screen in steps of 45 degrees
__device__ __constant__ int kDx[8] = {1, 1, 0, -1, -1, -1, 0, 1};
__device__ __constant__ int kDy[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// Suzuki border following of the outer border that starts at pixel (x0, y0) (icvFetchContour, is_hole = 0): twice the signed
// polygon area and the bounding box of the visited points; `cap` bounds the walk (a border has at most 4 * pixels steps)
__device__ void follow_outer(const unsigned char* f, int w, int h, int x0, int y0, long long* area2, int* bb, long long cap) {
    auto at = [&](int x, int y) { return (unsigned)x < (unsigned)w && (unsigned)y < (unsigned)h && f[(size_t)y * w + x] != 0; };
    int s = 4, x1 = x0, y1 = y0;
    do {                                            // clockwise from the left neighbour for the first nonzero pixel
        s = (s - 1) & 7;
        x1 = x0 + kDx[s];
        y1 = y0 + kDy[s];
    } while (!at(x1, y1) && s != 4);
    bb[0] = bb[2] = x0;
    bb[1] = bb[3] = y0;
    *area2 = 0;
    if (s == 4) return;                             // an isolated pixel: one point, area 0
    long long acc = 0;
    int x3 = x0, y3 = y0;
    for (long long step = 0; step < cap; ++step) {
        int x4, y4;
        do {                                        // counter-clockwise from the neighbour after the one we came from
            s = (s + 1) & 7;
            x4 = x3 + kDx[s];
            y4 = y3 + kDy[s];
        } while (!at(x4, y4));
        acc += (long long)x3 * y4 - (long long)x4 * y3;      // point (x3, y3) and its successor (x4, y4)
        bb[0] = min(bb[0], x3); bb[1] = min(bb[1], y3); bb[2] = max(bb[2], x3); bb[3] = max(bb[3], y3);
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
    }
    *area2 = acc < 0 ? -acc : acc;
}

__global__ __launch_bounds__(256) void contour_kernel(const MaskP p) {
    const int m = blockIdx.y;
    const int px = p.w * p.h;
    const unsigned char* f = p.feat + (size_t)m * px;
    const int* lab = p.labels + (size_t)m * px;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < px; i += gridDim.x * blockDim.x) {
        if (lab[i] != i) continue;                  // roots only: the first pixel of a component in raster order
        long long a2;
        int bb[4];
        follow_outer(f, p.w, p.h, i % p.w, i / p.w, &a2, bb, 4ll * px + 16);
        if (a2 > 0) atomicMax(&p.keys[m], ((unsigned long long)a2 << 32) | (unsigned)i);      // largest area, then the LATEST start
    }
}

__global__ __launch_bounds__(64) void select_kernel(const MaskP p, int maps, int* out) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= maps) return;
    const unsigned long long key = p.keys[m];
    int* o = out + 5 * (size_t)m;
    if (key == 0) {
        o[0] = o[1] = o[2] = o[3] = o[4] = 0;
        return;
    }
    const int i = (int)(key & 0xFFFFFFFFull);
    const int px = p.w * p.h;
    long long a2;
    int bb[4];
    follow_outer(p.feat + (size_t)m * px, p.w, p.h, i % p.w, i / p.w, &a2, bb, 4ll * px + 16);
    o[0] = 1;
    o[1] = bb[0];
    o[2] = bb[1];
    o[3] = bb[2] - bb[0] + 1;
    o[4] = bb[3] - bb[1] + 1;
}

inline size_t align256(size_t v) { return (v + 255) / 256 * 256; }

}  // namespace

extern "C" {

size_t fcn_score_masks_workspace_bytes(int n_windows, int num_classes, int w, int h) {
    if (n_windows <= 0 || num_classes < 2 || w <= 0 || h <= 0) return 0;
    const size_t maps = (size_t)n_windows * (num_classes - 1), px = (size_t)w * h;
    return align256(maps * px) + align256(maps * px * sizeof(int)) + align256(maps * sizeof(unsigned long long));
}

int fcn_score_masks(const float* score, int N, int C, int H, int W, int cstride, int coffset, const int32_t* h_rects, float prob_thresh,
                    uint8_t* pmap, int frame_h, int frame_w, void* d_workspace, int32_t* out, fcn_stream_t s) {
    FCN_REQUIRE(score && h_rects && pmap && d_workspace && out, FCN_E_ARG, "score_masks: null");
    FCN_REQUIRE(N > 0 && N <= 32 && C >= 2 && H > 0 && W > 0 && cstride >= coffset + C && coffset >= 0 && frame_h > 0 && frame_w > 0,
                FCN_E_ARG, "score_masks: bad extents (1..32 windows, at least two classes)");
    const int w = h_rects[2], h = h_rects[3];
    FCN_REQUIRE(w > 0 && h > 0 && (long long)w * h < (1ll << 30), FCN_E_ARG, "score_masks: bad window size %dx%d", w, h);
    FCN_REQUIRE(((uintptr_t)pmap & 3) == 0, FCN_E_ALIGN, "score_masks: pmap must be 4-byte aligned (32-bit atomic OR)");
    MaskP p;
    p.score = score;
    p.N = N; p.C = C; p.H = H; p.W = W; p.cstride = cstride; p.coffset = coffset;
    p.w = w; p.h = h; p.frame_h = frame_h; p.frame_w = frame_w;
    p.thresh = prob_thresh;
    p.scale_x = 1.0 / ((double)w / (double)W);
    p.scale_y = 1.0 / ((double)h / (double)H);
    for (int i = 0; i < N; ++i) {
        FCN_REQUIRE(h_rects[4 * i + 2] == w && h_rects[4 * i + 3] == h, FCN_E_ARG, "score_masks: all windows must have the same size");
        p.rect_xy[2 * i] = h_rects[4 * i];
        p.rect_xy[2 * i + 1] = h_rects[4 * i + 1];
    }
    const size_t maps = (size_t)N * (C - 1), px = (size_t)w * h;
    char* ws = reinterpret_cast<char*>(d_workspace);
    p.pmap = pmap;
    p.feat = reinterpret_cast<unsigned char*>(ws);
    p.labels = reinterpret_cast<int*>(ws + align256(maps * px));
    p.keys = reinterpret_cast<unsigned long long*>(ws + align256(maps * px) + align256(maps * px * sizeof(int)));
    FCN_HIP(hipMemsetAsync(p.keys, 0, maps * sizeof(unsigned long long), as_stream(s)));
    const int bx = (int)((px + 255) / 256 < 4096 ? (px + 255) / 256 : 4096);
    const dim3 grid(bx, (unsigned)maps);
    hipLaunchKernelGGL(score_mask_kernel, grid, dim3(256), 0, as_stream(s), p);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, as_stream(s), p);
    hipLaunchKernelGGL(contour_kernel, grid, dim3(256), 0, as_stream(s), p);
    hipLaunchKernelGGL(select_kernel, dim3((unsigned)((maps + 63) / 64)), dim3(64), 0, as_stream(s), p, (int)maps, out);
    FCN_LAUNCH_CHECK("score_masks");
    return 0;
}

}  // extern "C"
