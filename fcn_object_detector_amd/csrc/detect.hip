// DetectNet post-processing and target generation for gfx950 — integer / index work, bit-exact.
//
// (1) fcn_detect_decode_group: gridbox_to_boxes + vote_boxes -> cv.groupRectangles
//     (reference: scripts/fcn_object_detector.py:337-394; OpenCV 3 objdetect
//     groupRectangles()/partition()/SimilarRects, un-vendored — semantics restated in DESIGN.md).
//     One workgroup per (image, class):
//       a. threshold the coverage map and compact the positive cells in row-major order
//          (wavefront ballot + popcount prefix, so candidate order == np.where order);
//       b. candidate = (double)bbox + cell origin, converted to int like the Python->Rect converter;
//       c. connected components of the SimilarRects relation with a lock-free union-find in LDS
//          (path halving) whose roots are always the smallest member index, so "classes numbered by
//          first member" falls out as the ascending order of roots; the M^2/2 similarity tests run
//          from LDS in 64 x 64 blocks dealt round-robin to the 16 waves;
//       d. integer sums per component (LDS atomics: exact and order-independent), float mean with
//          round-half-even, the n <= groupThreshold and containment filters, the height filter of
//          vote_boxes, and an ordered compaction of the survivors.
// (2) fcn_gen_targets: ArgumentationEngine.bounding_box_parameterized_labels
//     (reference: scripts/data_argumentation_layer/argumentation_engine.py:69-109 with
//     JaccardCoeff.iou :26-55, generate_box_labels :272-278, grid_region :283-292).
#include <math.h>

#include <atomic>

#include "common.h"

using namespace fcn;

namespace {

constexpr int kDetThreads = 1024;
constexpr int kDetWaves = kDetThreads / 64;
// CANDIDATES (cells at or above the threshold) of one (image, class) the LDS holds: parent + count + 4 sums = 6 x 20 KiB, plus
// 2 x 10 KiB of containment margins = 140 of the CU's 160 KiB.  Round 4: the bound is on the candidates, no longer on the grid -
// every LDS structure is indexed by candidate, the per-cell planes live in the workspace - and 5120 covers the largest grid the
// reference's own vectors hold with EVERY cell firing (640 x 480 at stride 8 = 4800 cells, fcn_object_detector.py:357-394).  A
// class with more candidates than this reports out_count = -1 (FCN_DETECT_OVERFLOW): the caller raises, nothing is dropped silently.
constexpr int kMaxCand = 5120;
constexpr int kMaxBig = kMaxCand / 2;   // classes with n > groupThreshold >= 1 members: at most M / 2
constexpr int kSliceMinCand = 192;      // below this many candidates one workgroup does the whole problem
constexpr int kMaxSlices = 8;

// workgroups per (image, class): enough to spread a lone frame's tests over the chip, one when the batch already fills it
inline int detect_slices(long long problems) {
    long long s = 256 / (problems > 0 ? problems : 1);
    return (int)(s < 1 ? 1 : (s > kMaxSlices ? kMaxSlices : s));
}

struct DetP {
    fcn_detect_params p;
    double eps;
    const float* cvg;
    const float* bbox;
    size_t cvg_image_stride, box_image_stride;
    int* ws;            // per problem: rects[4*G] + list[G] + slice forests[S*G]; behind all problems: one arrival word each
    int slices;         // workgroups per (image, class): the SimilarRects tests of a problem are dealt to S workgroups
    unsigned seq;       // launch sequence number (24 bits used): the arrival words need no reset and no zero-initialised workspace
    int32_t* out_rects;
    int32_t* out_weights;
    int32_t* out_count;
};

__device__ __forceinline__ int round_coord(double v, int mode) {
    double r = mode == FCN_RECT_ROUND_TRUNCATE ? trunc(v) : rint(v);  // rint: round-half-even (cvRound)
    r = fmin(fmax(r, -2147483648.0), 2147483647.0);
    return (int)r;
}

// exclusive prefix of `flag` over the workgroup in thread order; returns the total through *total
__device__ __forceinline__ int block_excl_scan(bool flag, int* wsum /*[kDetWaves]*/, int* total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    const int within = __popcll(m & ((1ull << lane) - 1ull));
    __syncthreads();  // previous users of wsum are done
    if (lane == 0) wsum[wid] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kDetWaves; ++w) {
        const int v = wsum[w];
        if (w < wid) base += v;
        tot += v;
    }
    *total = tot;
    return base + within;
}

// Lock-free find with path halving.  Invariant: parent[x] <= x and only a ROOT is ever hooked (CAS expecting parent[a] == a),
// so re-pointing a non-root at its grandparent - another ancestor - races with nothing: concurrent halvings all write
// ancestors, and a CAS on a non-root fails whatever it holds.  Without it the chains of a dense cluster grow to O(M).
__device__ __forceinline__ int uf_find(int* parent, int x) {
    while (true) {
        const int p = __hip_atomic_load(&parent[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (p == x) return x;
        const int gp = __hip_atomic_load(&parent[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (gp == p) return p;
        __hip_atomic_store(&parent[x], gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        x = gp;
    }
}

// union keeping the smaller index as root
__device__ __forceinline__ void uf_union(int* parent, int a, int b) {
    while (true) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }  // a > b: hook a under b
        const int old = atomicCAS(&parent[a], a, b);
        if (old == a) return;
    }
}

__global__ __launch_bounds__(kDetThreads) void detect_kernel(DetP d) {
    __shared__ int parent[kMaxCand];
    __shared__ int cnt[kMaxCand];
    __shared__ int sum[4][kMaxCand];
    __shared__ int bdx[kMaxBig], bdy[kMaxBig];      // containment margins of the classes with n > groupThreshold
    __shared__ int wsum[kDetWaves];
    __shared__ int s_any;
    __shared__ int s_last;

    const fcn_detect_params& P = d.p;
    const int S = d.slices;
    const int prob = blockIdx.x / S;                 // (image, class)
    const int slice = blockIdx.x - prob * S;         // which share of the problem's SimilarRects tests this workgroup runs
    const int cls = prob % P.num_classes;
    const int img = prob / P.num_classes;
    const int G = P.gy * P.gx;
    const int tid = threadIdx.x;
    const float* cvg = d.cvg + (size_t)img * d.cvg_image_stride;
    const float* box = d.bbox + (size_t)img * d.box_image_stride;
    int* rects = d.ws + (size_t)prob * (5 + S) * G;  // [4][G] as x | y | w | h planes (every slice writes the same values)
    int* list = rects + 4 * (size_t)G;               // compacted root / survivor lists (used by the finishing workgroup only)
    int* forests = list + G;                         // [S][G]: each slice's component roots
    unsigned* arrive = reinterpret_cast<unsigned*>(d.ws + (size_t)gridDim.x / S * (5 + S) * G) + prob;
    int32_t* o_rects = d.out_rects + (size_t)prob * P.max_out * 4;
    int32_t* o_w = d.out_weights + (size_t)prob * P.max_out;

    if (tid == 0) s_any = 0;

    // ---- a+b: threshold, ordered compaction, candidate rectangles ----------------------------
    int M = 0;
    bool any_nonzero = false;
    for (int base = 0; base < G; base += kDetThreads) {
        const int cell = base + tid;
        bool pos = false;
        if (cell < G) pos = cvg[(size_t)cell * P.cvg_cstride + P.cvg_coffset + cls] >= P.prob_thresh;
        int tot;
        const int at = M + block_excl_scan(pos, wsum, &tot);
        if (pos) {
            const int y = cell / P.gx, x = cell - y * P.gx;
            const float* b = box + (size_t)cell * P.box_cstride + P.box_coffset + 4 * cls;
            const double x1 = (double)b[0] + (double)(x * P.cell_w);
            const double y1 = (double)b[1] + (double)(y * P.cell_h);
            const double x2 = (double)b[2] + (double)(x * P.cell_w);
            const double y2 = (double)b[3] + (double)(y * P.cell_h);
            any_nonzero |= (x1 != 0.0) | (y1 != 0.0) | (x2 != 0.0) | (y2 != 0.0);
            rects[0 * G + at] = round_coord(x1, P.round_mode);
            rects[1 * G + at] = round_coord(y1, P.round_mode);
            rects[2 * G + at] = round_coord(x2, P.round_mode);  // read back as width  (reference passes x2)
            rects[3 * G + at] = round_coord(y2, P.round_mode);  // read back as height (reference passes y2)
        }
        M += tot;
    }
    if (M > kMaxCand) {                              // (every slice of the problem sees the same M and takes the same exit)
        if (tid == 0 && slice == 0) d.out_count[prob] = -1;
        return;
    }
    if (any_nonzero) atomicOr(&s_any, 1);
    for (int i = tid; i < M; i += kDetThreads) {
        parent[i] = i;
        cnt[i] = 0;
        sum[0][i] = sum[1][i] = sum[2][i] = sum[3][i] = 0;
    }
    __syncthreads();
    // vote_boxes: `if not propose_boxes.any(): return []`
    if (M == 0 || s_any == 0) {                      // (every slice of the problem sees the same M and takes the same exit)
        if (tid == 0 && slice == 0) d.out_count[prob] = 0;
        return;
    }

    // groupRectangles: groupThreshold <= 0 returns the input untouched with weight 1
    if (P.group_thresh <= 0) {
        if (slice != 0) return;
        int outn = 0;
        for (int base = 0; base < M; base += kDetThreads) {
            const int i = base + tid;
            const bool keep = i < M && rects[3 * G + i] - rects[1 * G + i] >= P.min_height;
            int tot;
            const int at = outn + block_excl_scan(keep, wsum, &tot);
            if (keep && at < P.max_out) {
                o_rects[at * 4 + 0] = rects[0 * G + i];
                o_rects[at * 4 + 1] = rects[1 * G + i];
                o_rects[at * 4 + 2] = rects[2 * G + i];
                o_rects[at * 4 + 3] = rects[3 * G + i];
                o_w[at] = 1;
            }
            outn += tot;
        }
        if (tid == 0) d.out_count[prob] = outn;
        return;
    }
    // Few candidates: one workgroup is faster than the hand-over between several
    const bool sliced = S > 1 && M > kSliceMinCand;
    if (!sliced && slice != 0) return;

    // ---- c: partition(): union every SimilarRects pair ----------------------------------------
    // M^2/2 tests (307 k for a full 28x28 grid): the candidates are staged in LDS - the four sum planes are free until
    // step d - and lane j keeps its own rectangle in registers while all lanes of a wave walk the same i (an LDS
    // broadcast), so a test costs four LDS reads and a handful of double operations instead of eight global loads.
    int4* cand = reinterpret_cast<int4*>(&sum[0][0]);           // 4 planes x kMaxCand ints == kMaxCand int4
    for (int i = tid; i < M; i += kDetThreads) cand[i] = make_int4(rects[0 * G + i], rects[1 * G + i], rects[2 * G + i], rects[3 * G + i]);
    // floor(eps * s * 0.5) for s = min(w) + min(h) in [0, kMaxCand): the count plane is idle until step d
    for (int sidx = tid; sidx < kMaxCand; sidx += kDetThreads) cnt[sidx] = (int)floor(fmin(d.eps * (double)sidx * 0.5, 2147483647.0));
    __syncthreads();
    // Work items of 64 x 64 tests (lane = one j of a 64-wide block, loop over a 64-wide block of i <= j) dealt round-robin to
    // the waves: the triangle is balanced over the 16 waves.  The differences are integers, so |d| <= delta (a double) is
    // |d| <= floor(delta): one double product and one conversion per pair, then integer compares.
    {
        const int lane = tid & 63, wave = tid >> 6;
        const int nb = (M + 63) >> 6;
        int item = 0;
        for (int b = 0; b < nb; ++b) {
            const int j = (b << 6) + lane;
            const int4 rj = cand[j < M ? j : 0];
            const int rjx = rj.x + rj.z, rjy = rj.y + rj.w;
            for (int c = 0; c <= b; ++c, ++item) {
                // items round-robin over the slices, a slice's items round-robin over its 16 waves
                if (sliced) {
                    if (item % S != slice || ((item / S) & (kDetWaves - 1)) != wave) continue;
                } else if ((item & (kDetWaves - 1)) != wave) {
                    continue;
                }
                const int i0 = c << 6;
                const int i1 = min(min(i0 + 64, M), j < M ? j : 0);      // this lane's tests of the item: i0 <= i < min(i1, j)
                const int iw = min(i0 + 64, M);                           // the wave walks the whole block
#pragma unroll 4
                for (int i = i0; i < iw; ++i) {
                    const int4 ri = cand[i];
                    const int sm = min(ri.z, rj.z) + min(ri.w, rj.w);
                    int t;
                    if ((unsigned)sm < (unsigned)kMaxCand) {
                        t = cnt[sm];
                    } else {            // sizes outside the table (or negative: delta < 0, nothing is similar)
                        const double delta = d.eps * (double)sm * 0.5;
                        t = delta < 0.0 ? -1 : (int)floor(fmin(delta, 2147483647.0));
                    }
                    // one predicate, no short-circuit branches: max of the four distances against t
                    const int far = max(max(abs(ri.x - rj.x), abs(ri.y - rj.y)), max(abs(ri.x + ri.z - rjx), abs(ri.y + ri.w - rjy)));
                    if ((far <= t) & (i < i1)) uf_union(parent, i, j);
                }
            }
        }
    }
    __syncthreads();
    if (sliced) {
        // Each slice publishes its forest (root of every candidate); the workgroup that arrives LAST merges the others'
        // into its own - M * (S - 1) unions - and finishes the problem.  Nobody waits for anybody: no spinning.
        for (int i = tid; i < M; i += kDetThreads) forests[(size_t)slice * G + i] = uf_find(parent, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __threadfence();                                   // release: this workgroup's stores before its arrival
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned tag = d.seq & 0xFFFFFFu;
            unsigned old = __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), want;
            while (true) {                                     // (launch tag | arrivals): a stale word of an earlier launch counts as zero
                want = (old >> 8) == tag ? old + 1 : ((tag << 8) | 1u);
                const unsigned prev = atomicCAS(arrive, old, want);
                if (prev == old) break;
                old = prev;
            }
            s_last = (int)(want & 0xFFu) == S;
            __threadfence();                                   // acquire: the other slices' forests
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (!s_last) return;
        for (int s2 = 0; s2 < S; ++s2) {
            if (s2 == slice) continue;
            const int* f = forests + (size_t)s2 * G;
            for (int i = tid; i < M; i += kDetThreads) {
                const int r = __builtin_nontemporal_load(f + i);
                if (r != i) uf_union(parent, i, r);
            }
        }
        __syncthreads();
    }
    for (int i = tid; i < M; i += kDetThreads) sum[0][i] = sum[1][i] = sum[2][i] = sum[3][i] = 0;
    for (int i = tid; i < kMaxCand; i += kDetThreads) cnt[i] = 0;
    __syncthreads();

    // ---- d: per-class integer sums -------------------------------------------------------------
    for (int i = tid; i < M; i += kDetThreads) {
        const int r = uf_find(parent, i);
        atomicAdd(&sum[0][r], rects[0 * G + i]);
        atomicAdd(&sum[1][r], rects[1 * G + i]);
        atomicAdd(&sum[2][r], rects[2 * G + i]);
        atomicAdd(&sum[3][r], rects[3 * G + i]);
        atomicAdd(&cnt[r], 1);
    }
    __syncthreads();
    // roots in ascending order == classes in partition()'s numbering
    int nc = 0;
    for (int base = 0; base < M; base += kDetThreads) {
        const int i = base + tid;
        const bool is_root = i < M && parent[i] == i;
        int tot;
        const int at = nc + block_excl_scan(is_root, wsum, &tot);
        if (is_root) {
            list[at] = i;
            // mean = saturate_cast<int>(sum * (1.f / n)): float multiply, round-half-even
            const float s = __fdiv_rn(1.f, (float)cnt[i]);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float v = __fmul_rn((float)sum[k][i], s);
                float r = rintf(v);
                r = fminf(fmaxf(r, -2147483648.f), 2147483520.f);
                sum[k][i] = (int)r;
            }
        }
        nc += tot;
    }
    __syncthreads();

    // Only classes with n > groupThreshold can contain another one: they are compacted first (parent[] is free now and
    // becomes their root list; their containment margins are rounded once), so the O(nc^2) scan of the reference
    // becomes O(nc * nk).  "Contained in SOME other class" does not depend on the scan order.
    int nk = 0;
    for (int base = 0; base < nc; base += kDetThreads) {
        const int ci = base + tid;
        const int r = ci < nc ? list[ci] : 0;
        const bool big = ci < nc && cnt[r] > P.group_thresh;
        int tot;
        const int at = nk + block_excl_scan(big, wsum, &tot);
        __syncthreads();            // every root test of this pass is done before parent[] is overwritten
        if (big && at < kMaxBig) {
            parent[at] = r;
            bdx[at] = round_coord((double)sum[2][r] * d.eps, FCN_RECT_ROUND_NEAREST_EVEN);
            bdy[at] = round_coord((double)sum[3][r] * d.eps, FCN_RECT_ROUND_NEAREST_EVEN);
        }
        nk += tot;
    }
    __syncthreads();
    // filters + ordered emission
    int outn = 0;
    for (int base = 0; base < nc; base += kDetThreads) {
        const int ci = base + tid;
        bool keep = false;
        int r1 = 0;
        if (ci < nc) {
            r1 = list[ci];
            const int n1 = cnt[r1];
            if (n1 > P.group_thresh) {
                const int x1 = sum[0][r1], y1 = sum[1][r1], w1 = sum[2][r1], h1 = sum[3][r1];
                keep = true;
                for (int k = 0; k < nk; ++k) {
                    const int r2 = parent[k];
                    if (r2 == r1) continue;
                    const int n2 = cnt[r2];
                    const int x2 = sum[0][r2], y2 = sum[1][r2], w2 = sum[2][r2], h2 = sum[3][r2];
                    const int dx = bdx[k], dy = bdy[k];
                    if (x1 >= x2 - dx && y1 >= y2 - dy && x1 + w1 <= x2 + w2 + dx && y1 + h1 <= y2 + h2 + dy &&
                        (n2 > max(3, n1) || n1 < 3)) {
                        keep = false;
                        break;
                    }
                }
                // vote_boxes: keep if rect[3] - rect[1] >= 20
                if (keep && h1 - y1 < P.min_height) keep = false;
            }
        }
        int tot;
        const int at = outn + block_excl_scan(keep, wsum, &tot);
        if (keep && at < P.max_out) {
            o_rects[at * 4 + 0] = sum[0][r1];
            o_rects[at * 4 + 1] = sum[1][r1];
            o_rects[at * 4 + 2] = sum[2][r1];
            o_rects[at * 4 + 3] = sum[3][r1];
            o_w[at] = cnt[r1];
        }
        outn += tot;
    }
    if (tid == 0) d.out_count[prob] = outn;
}

// ---------------------------------------------------------------------------------------------
// target generation: one lane per (image, class, cell)
// ---------------------------------------------------------------------------------------------
// NHWC = false: the Python layer's NCHW tops; NHWC = true: the engine's blob buffers (channel strides fg_cs / blk_cs)
template <bool NHWC>
__global__ __launch_bounds__(256) void gen_targets_kernel(const int32_t* __restrict__ rects, const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ offs, int batch, int C, int gy, int gx, int stride,
                                                          double iou_thresh, float* __restrict__ fg, float* __restrict__ bbox,
                                                          float* __restrict__ size, float* __restrict__ obj, float* __restrict__ cvgb,
                                                          int fg_cs, int blk_cs) {
    const int G = gy * gx;
    const long long total = (long long)batch * C * G;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int cell = (int)(t % G);
        const int cls = (int)((t / G) % C);
        const int img = (int)(t / ((long long)G * C));
        const int j = cell / gx, i = cell - j * gx;
        const int cx = i * stride, cy = j * stride;
        int hit = -1;
        for (int r = offs[img]; r < offs[img + 1]; ++r) {
            if (labels[r] != cls) continue;
            const int rx = rects[4 * r + 0], ry = rects[4 * r + 1], rw = rects[4 * r + 2], rh = rects[4 * r + 3];
            // JaccardCoeff.iou(cell, rect) — argumentation_engine.py:26-55
            const int ix = max(cx, rx), iy = max(cy, ry);
            const int iw = min(cx + stride, rx + rw) - ix;
            const int ih = min(cy + stride, ry + rh) - iy;
            if (iw < 0 || ih < 0) continue;
            const int ux = min(cx, rx), uy = min(cy, ry);
            const int uw = max(cx + stride, rx + rw) - ux;
            const int uh = max(cy + stride, ry + rh) - uy;
            const float aub = (float)((double)uw * (double)uh);   // area of the bounding box of the union
            const float anb = (float)((double)iw * (double)ih);
            const float area_ratio = __fdiv_rn((float)((double)stride * (double)stride), (float)((double)rw * (double)rh));
            float score = __fdiv_rn(anb, aub);
            score = __fdiv_rn(score, area_ratio);
            if ((double)score > iou_thresh) hit = r;  // later rects overwrite earlier ones
        }
        if (NHWC) {
            float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, s0 = 0.f, s1 = 0.f, ob = 0.f, cv = 0.f;
            if (hit >= 0) {
                const int rx = rects[4 * hit + 0], ry = rects[4 * hit + 1], rw = rects[4 * hit + 2], rh = rects[4 * hit + 3];
                b0 = (float)(rx - cx); b1 = (float)(ry - cy); b2 = (float)(rx + rw - cx); b3 = (float)(ry + rh - cy);
                s0 = (float)(1.0 / (double)rw); s1 = (float)(1.0 / (double)rh);
                ob = __fdiv_rn((float)((double)stride * (double)stride), (float)((double)rw * (double)rh));
                cv = 1.f;
            }
            const size_t o = ((size_t)img * G + cell) * blk_cs + 4 * cls;
            bbox[o] = b0; bbox[o + 1] = b1; bbox[o + 2] = b2; bbox[o + 3] = b3;
            size[o] = s0; size[o + 1] = s1; size[o + 2] = s0; size[o + 3] = s1;
            obj[o] = ob; obj[o + 1] = ob; obj[o + 2] = ob; obj[o + 3] = ob;
            cvgb[o] = cv; cvgb[o + 1] = cv; cvgb[o + 2] = cv; cvgb[o + 3] = cv;
            fg[((size_t)img * G + cell) * fg_cs + cls] = cv;
            continue;
        }
        const size_t o4 = ((size_t)img * 4 * C + 4 * cls) * G + cell;
        float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f, s0 = 0.f, s1 = 0.f, ob = 0.f, cv = 0.f;
        if (hit >= 0) {
            const int rx = rects[4 * hit + 0], ry = rects[4 * hit + 1], rw = rects[4 * hit + 2], rh = rects[4 * hit + 3];
            b0 = (float)(rx - cx);
            b1 = (float)(ry - cy);
            b2 = (float)(rx + rw - cx);
            b3 = (float)(ry + rh - cy);
            s0 = (float)(1.0 / (double)rw);
            s1 = (float)(1.0 / (double)rh);
            ob = __fdiv_rn((float)((double)stride * (double)stride), (float)((double)rw * (double)rh));
            cv = 1.f;
        }
        bbox[o4] = b0; bbox[o4 + G] = b1; bbox[o4 + 2 * (size_t)G] = b2; bbox[o4 + 3 * (size_t)G] = b3;
        size[o4] = s0; size[o4 + G] = s1; size[o4 + 2 * (size_t)G] = s0; size[o4 + 3 * (size_t)G] = s1;
        obj[o4] = ob; obj[o4 + G] = ob; obj[o4 + 2 * (size_t)G] = ob; obj[o4 + 3 * (size_t)G] = ob;
        cvgb[o4] = cv; cvgb[o4 + G] = cv; cvgb[o4 + 2 * (size_t)G] = cv; cvgb[o4 + 3 * (size_t)G] = cv;
        fg[((size_t)img * C + cls) * G + cell] = cv;
    }
}

}  // namespace

extern "C" {

size_t fcn_detect_workspace_bytes(const fcn_detect_params* h_p, int batch) {
    if (!h_p || batch <= 0) return 0;
    const size_t problems = (size_t)batch * h_p->num_classes;
    return (problems * (5 + detect_slices((long long)problems)) * (size_t)h_p->gy * h_p->gx + problems) * sizeof(int32_t);
}

int fcn_detect_decode_group(const float* cvg, const float* bbox, int batch, size_t cvg_image_stride, size_t box_image_stride,
                            const fcn_detect_params* h_p, void* d_workspace, int32_t* out_rects, int32_t* out_weights,
                            int32_t* out_count, fcn_stream_t s) {
    FCN_REQUIRE(cvg && bbox && h_p && d_workspace && out_rects && out_weights && out_count && batch > 0, FCN_E_ARG, "detect: null/empty");
    const fcn_detect_params& P = *h_p;
    FCN_REQUIRE(P.num_classes > 0 && P.gy > 0 && P.gx > 0 && P.max_out > 0, FCN_E_ARG, "detect: bad grid/classes/max_out");
    FCN_REQUIRE((long long)P.gy * P.gx < (1 << 24), FCN_E_UNSUPPORTED, "detect: grid %dx%d too large", P.gy, P.gx);
    FCN_REQUIRE(P.cvg_coffset >= 0 && P.cvg_cstride >= P.cvg_coffset + P.num_classes, FCN_E_ARG, "detect: coverage slice out of range");
    FCN_REQUIRE(P.box_coffset >= 0 && P.box_cstride >= P.box_coffset + 4 * P.num_classes, FCN_E_ARG, "detect: bbox slice out of range");
    FCN_REQUIRE(P.round_mode == FCN_RECT_ROUND_NEAREST_EVEN || P.round_mode == FCN_RECT_ROUND_TRUNCATE, FCN_E_ARG, "detect: bad round_mode");
    FCN_REQUIRE((long long)batch * P.num_classes < (1 << 30), FCN_E_UNSUPPORTED, "detect: too many problems");
    DetP d;
    d.p = P;
    d.eps = P.eps;
    d.cvg = cvg;
    d.bbox = bbox;
    d.cvg_image_stride = cvg_image_stride;
    d.box_image_stride = box_image_stride;
    d.ws = reinterpret_cast<int*>(d_workspace);
    d.out_rects = out_rects;
    d.out_weights = out_weights;
    d.out_count = out_count;
    static std::atomic<unsigned> launch_seq{1};
    d.slices = detect_slices((long long)batch * P.num_classes);
    do {
        d.seq = launch_seq.fetch_add(1, std::memory_order_relaxed);
    } while ((d.seq & 0xFFFFFFu) == 0);      // tag 0 is what a zero-filled workspace holds
    hipLaunchKernelGGL(detect_kernel, dim3(batch * P.num_classes * d.slices), dim3(kDetThreads), 0, as_stream(s), d);
    FCN_LAUNCH_CHECK("detect_decode_group");
    return 0;
}

int fcn_gen_targets(const int32_t* rects, const int32_t* labels, const int32_t* rect_offsets, int batch, int num_classes, int gy, int gx,
                    int stride, double iou_thresh, float* foreground, float* bbox, float* size, float* obj, float* cvg_block,
                    fcn_stream_t s) {
    FCN_REQUIRE(rects && labels && rect_offsets && foreground && bbox && size && obj && cvg_block, FCN_E_ARG, "gen_targets: null");
    FCN_REQUIRE(batch > 0 && num_classes > 0 && gy > 0 && gx > 0 && stride > 0, FCN_E_ARG, "gen_targets: bad extents");
    const long long total = (long long)batch * num_classes * gy * gx;
    hipLaunchKernelGGL(gen_targets_kernel<false>, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(s), rects, labels, rect_offsets, batch,
                       num_classes, gy, gx, stride, iou_thresh, foreground, bbox, size, obj, cvg_block, 0, 0);
    FCN_LAUNCH_CHECK("gen_targets");
    return 0;
}

int fcn_gen_targets_nhwc(const int32_t* rects, const int32_t* labels, const int32_t* rect_offsets, int batch, int num_classes, int gy, int gx,
                         int stride, double iou_thresh, float* foreground, int fg_cstride, float* bbox, float* size, float* obj,
                         float* cvg_block, int blk_cstride, fcn_stream_t s) {
    FCN_REQUIRE(rects && labels && rect_offsets && foreground && bbox && size && obj && cvg_block, FCN_E_ARG, "gen_targets_nhwc: null");
    FCN_REQUIRE(batch > 0 && num_classes > 0 && gy > 0 && gx > 0 && stride > 0 && fg_cstride >= num_classes && blk_cstride >= 4 * num_classes,
                FCN_E_ARG, "gen_targets_nhwc: bad extents");
    const long long total = (long long)batch * num_classes * gy * gx;
    hipLaunchKernelGGL(gen_targets_kernel<true>, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(s), rects, labels, rect_offsets, batch,
                       num_classes, gy, gx, stride, iou_thresh, foreground, bbox, size, obj, cvg_block, fg_cstride, blk_cstride);
    FCN_LAUNCH_CHECK("gen_targets_nhwc");
    return 0;
}

}  // extern "C"
