/* ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/caffe_ref.py).  Parity unpinned, like the numpy file this one accelerates.
 *
 * Plain-C restatement of the memory-bound layers of Caffe's CPU path (BVLC / NVIDIA-Caffe 0.15, the un-vendored dependency
 * of scripts/fcn_object_detector.py:87 `net.forward()`): im2col_cpu (src/caffe/util/im2col.cpp), PoolingLayer::Forward_cpu
 * MAX (src/caffe/layers/pooling_layer.cpp), LRNLayer::CrossChannelForward_cpu (src/caffe/layers/lrn_layer.cpp) - the same
 * loops in the same order, one OpenMP loop over the outermost independent index, so that the CPU baseline bench.py reports
 * is a compiled Caffe-like program and not numpy indexing overhead.  oracle/caffe_ref.py loads
 * libcaffe_cpu.so when it has been built (oracle/Makefile) and tests/test_oracle.py checks every function here against the
 * numpy statement of the same layer.  The GEMM stays OpenBLAS' sgemm (numpy matmul), as in Caffe.
 * All tensors NCHW float32. */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* threads for a loop over `work` elements: small layers run on the calling thread (forking a team costs more than the loop),
 * large ones on at most 16 threads (these loops are memory-bound; a 256-thread team on a big host only adds fork/join time) */
static int team(size_t work) {
#ifdef _OPENMP
    const int m = omp_get_max_threads();
    return work < 65536 ? 1 : (m < 16 ? m : 16);
#else
    (void)work;
    return 1;
#endif
}

/* (C, H, W) -> (C*kh*kw, OH*OW): row (c, r, q), zero padding */
void oracle_im2col_f32(const float* x, int C, int H, int W, int kh, int kw, int ph, int pw, int sh, int sw, int OH, int OW, float* col) {
    const int rows = C * kh * kw;
#pragma omp parallel for schedule(static) num_threads(team((size_t)rows * OH * OW))
    for (int row = 0; row < rows; ++row) {
        const int q = row % kw, r = (row / kw) % kh, c = row / (kw * kh);
        const float* xc = x + (size_t)c * H * W;
        float* out = col + (size_t)row * OH * OW;
        for (int oy = 0; oy < OH; ++oy) {
            const int iy = oy * sh - ph + r;
            float* o = out + (size_t)oy * OW;
            if (iy < 0 || iy >= H) {
                memset(o, 0, sizeof(float) * OW);
                continue;
            }
            const float* xr = xc + (size_t)iy * W;
            for (int ox = 0; ox < OW; ++ox) {
                const int ix = ox * sw - pw + q;
                o[ox] = (ix >= 0 && ix < W) ? xr[ix] : 0.f;
            }
        }
    }
}

/* MAX pooling: window clipped to the image, -FLT_MAX start, strict `>` (first maximum in raster order wins);
 * idx (may be NULL) = iy * W + ix of the winner, -1 where the window is empty.  Without idx (TEST phase) the window is
 * walked one element (r, q) at a time over whole output rows - the same compares in the same order, in a form the
 * compiler turns into vector max instructions. */
void oracle_maxpool_f32(const float* x, int planes, int H, int W, int k, int s, int p, int OH, int OW, float* y, int64_t* idx) {
#pragma omp parallel for schedule(static) num_threads(team((size_t)planes * OH * OW * k))
    for (int pl = 0; pl < planes; ++pl) {
        const float* xp = x + (size_t)pl * H * W;
        float* yp = y + (size_t)pl * OH * OW;
        if (!idx) {
            for (int oy = 0; oy < OH; ++oy) {
                float* yr = yp + (size_t)oy * OW;
                for (int ox = 0; ox < OW; ++ox) yr[ox] = -FLT_MAX;
                for (int r = 0; r < k; ++r) {
                    const int iy = oy * s - p + r;
                    if (iy < 0 || iy >= H) continue;
                    const float* xr = xp + (size_t)iy * W;
                    for (int q = 0; q < k; ++q) {
                        /* outputs whose element q lies inside the row: 0 <= ox * s - p + q < W */
                        int lo = p - q > 0 ? (p - q + s - 1) / s : 0;
                        int hi = (W - 1 + p - q) / s + 1;
                        if (hi > OW) hi = OW;
                        const float* xq = xr - p + q;
                        if (s == 1)
                            for (int ox = lo; ox < hi; ++ox) yr[ox] = xq[ox] > yr[ox] ? xq[ox] : yr[ox];
                        else
                            for (int ox = lo; ox < hi; ++ox) yr[ox] = xq[ox * s] > yr[ox] ? xq[ox * s] : yr[ox];
                    }
                }
            }
            continue;
        }
        int64_t* ip = idx + (size_t)pl * OH * OW;
        for (int oy = 0; oy < OH; ++oy)
            for (int ox = 0; ox < OW; ++ox) {
                int hs = oy * s - p, ws = ox * s - p;
                const int he = hs + k < H ? hs + k : H, we = ws + k < W ? ws + k : W;
                hs = hs > 0 ? hs : 0;
                ws = ws > 0 ? ws : 0;
                float best = -FLT_MAX;
                int64_t arg = -1;
                for (int iy = hs; iy < he; ++iy)
                    for (int ix = ws; ix < we; ++ix)
                        if (xp[iy * W + ix] > best) {
                            best = xp[iy * W + ix];
                            arg = iy * W + ix;
                        }
                yp[oy * OW + ox] = best;
                ip[oy * OW + ox] = arg;
            }
    }
}

/* across-channel LRN: scale = k + alpha/n * sum_{window} x^2 (zero padded), y = x * scale^-beta.  The window sum is
 * accumulated in ascending channel order from zero, like oracle/caffe_ref.py::lrn_across. */
void oracle_lrn_f32(const float* x, int N, int C, int H, int W, int local_size, float alpha_over_n, float beta, float k, float* y,
                    float* scale) {
    const int pre = (local_size - 1) / 2;
    const size_t hw = (size_t)H * W;
#pragma omp parallel for schedule(static) collapse(2) num_threads(team((size_t)N * C * hw))
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < C; ++c) {
            const float* xn = x + (size_t)n * C * hw;
            float* sc = scale + ((size_t)n * C + c) * hw;
            float* yc = y + ((size_t)n * C + c) * hw;
            for (size_t i = 0; i < hw; ++i) sc[i] = 0.f;
            for (int j = 0; j < local_size; ++j) {
                const int cc = c - pre + j;
                if (cc < 0 || cc >= C) continue;
                const float* xc = xn + (size_t)cc * hw;
                for (size_t i = 0; i < hw; ++i) sc[i] += xc[i] * xc[i];
            }
            const float* x0 = xn + (size_t)c * hw;
            for (size_t i = 0; i < hw; ++i) {
                sc[i] = k + alpha_over_n * sc[i];
                yc[i] = x0[i] * powf(sc[i], -beta);
            }
        }
}
