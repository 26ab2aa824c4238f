// Convolution forward for gfx950: NHWC implicit GEMM on the f32-input matrix cores.
//
// Stands in for Caffe's ConvolutionLayer::Forward (+ the in-place ReLU, the Sigmoid head and the
// Power(shift) input transform that follow or precede it)
// as executed by net.forward() in the reference (scripts/fcn_object_detector.py:87) over
// models/deploy.prototxt:8-2176.
//
//   out[m][n] = bias[n] + sum_k A[m][k] * Wt[n][k]      m = (img, oy, ox)   n = output channel
//   k = (r*kw + q)*Cin + c ,  A[m][k] = x[img][oy*s-p+r][ox*s-p+q][c]  (0 outside the image)
//
// Data layout: activations NHWC (channel-contiguous, so an A-row segment of 4 consecutive k is one
// 16-byte load), weights OHWI = [Cout][kh][kw][Cin] (a Wt row is K contiguous floats).  Both tiles
// are staged global -> registers -> LDS as [row][32 k + 4 pad] floats; the pad makes the
// ds_read_b128 fragment reads bank-conflict free (row stride 36 dwords: 16 rows hit 16 distinct
// 4-dword slots of the 64-bank row).  Each wave accumulates 32x32 output tiles with
// v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the chip's f32 peak).  A k-step of 8 uses
// one b128 read per operand: lanes 0-31 hold k+0..3, lanes 32-63 hold k+4..7, and MFMA j consumes
// element j of both fragments, so A and B see the same k permutation.
//
// Small-M layers (28x28 grid, M = 784) would leave most of the 256 CUs idle with large tiles, so
// the tile shape is a template parameter chosen per launch, down to one 32x32 tile per workgroup
// with the 4 waves splitting K (WAVES_K) and reducing through LDS.  Several independent problems
// (the branches of an inception module) can share ONE launch (fcn_conv2d_fwd_group_f32).
#include <mutex>

#include "common.h"

using namespace fcn;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace fcn {

// one 16-byte zero page per device, allocated on first use (never inside a graph capture: the engine
// issues an eager warm-up launch before it captures)
const float* zero_page_for_current_device(int* rc) {
    static std::mutex mu;
    static const float* pages[64] = {nullptr};
    int dev = 0;
    *rc = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { *rc = set_err(FCN_E_STATE, "conv: no current device"); return nullptr; }
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        void* ptr = nullptr;
        hipError_t e = hipMalloc(&ptr, 256);
        if (e == hipSuccess) e = hipMemset(ptr, 0, 256);
        if (e != hipSuccess) { *rc = set_err(-(int)e, "conv: zero page allocation failed: %s", hipGetErrorString(e)); return nullptr; }
        pages[dev] = reinterpret_cast<const float*>(ptr);
    }
    return pages[dev];
}

}  // namespace fcn

namespace {

struct ConvP {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* y2;
    int N, H, W, Cin, x_cstride;
    int Cout, kh, kw, pad, stride, OH, OW;
    int y_cstride, y_coffset, y2_cstride, y2_coffset;
    int flags;
    float in_shift;
    int M, K, tiles_m, tiles_n, tile_end;  // tile_end: exclusive prefix end of this problem's tiles in a group launch
    int kw_magic, bk_taps, bk_rem;
    const float* zero_page;               // 16 zero bytes in HBM: what out-of-image / out-of-tile lanes load
};

constexpr int BK = 32;
constexpr int LDS_ROW = BK + 4;  // floats per staged row

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int STAGES>
struct Cfg {
    static constexpr int BM = 32 * WTM * WAVES_M;
    static constexpr int BN = 32 * WTN * WAVES_N;
    static constexpr int NT = 64 * WAVES_M * WAVES_N * WAVES_K;
    static constexpr int A_IT = BM * 8 / NT;
    static constexpr int B_IT = BN * 8 / NT;
    static constexpr int STAGE_FLOATS = 2 * (BM + BN) * LDS_ROW;
    static constexpr int RED_FLOATS = WAVES_M * WAVES_N * (WAVES_K - 1) * WTM * WTN * 16 * 64;
    static constexpr int LDS_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
    static_assert(BM * 8 % NT == 0 && BN * 8 % NT == 0, "tile rows must divide over the threads");
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the workgroup");
    static_assert(WAVES_K == 1 || WAVES_K == 2 || WAVES_K == 4, "K split over 1, 2 or 4 waves");
    static_assert(STAGES >= 2 && STAGES <= 8 && STAGES % 2 == 0, "2, 4, 6 or 8 k-chunks in flight (even: LDS buffer parity is static)");
    static_assert(STAGES * A_IT <= 32, "shift_mask holds one bit per staged A segment");
};

// explicit global-address-space load: a pointer picked from a struct in memory (group launch) or selected
// against the zero page would otherwise be "generic" and compile to flat_load, which cannot be waited on
// with a counted vmcnt
typedef float v4f __attribute__((ext_vector_type(4)));
typedef const v4f __attribute__((address_space(1))) * gv4f_ptr;
__device__ __forceinline__ v4f ld4(const float* p) { return *(gv4f_ptr)(p); }

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int STAGES>
__device__ __forceinline__ void conv_body(const ConvP& p, int tile, float* smem) {
    using C = Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES>;
    constexpr int BM = C::BM, BN = C::BN, NT = C::NT, A_IT = C::A_IT, B_IT = C::B_IT;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wk = wid % WAVES_K;
    const int wn = (wid / WAVES_K) % WAVES_N;
    const int wm = wid / (WAVES_K * WAVES_N);

    const int tile_m = tile / p.tiles_n;
    const int tile_n = tile - tile_m * p.tiles_n;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    float* sA = smem;                      // [2][BM][LDS_ROW]
    float* sB = smem + 2 * BM * LDS_ROW;   // [2][BN][LDS_ROW]

    // ---- per-thread loader state -------------------------------------------------------------
    const int seg = tid & 7;        // which 16-byte segment of the 32-float k-chunk
    const int row0 = tid >> 3;      // first staged row of this thread; further rows every NT/8
    // k position of this thread's segment, kept as (tap, channel) and advanced by BK per chunk without branches
    int kc = seg * 4;
    int kt = kc / p.Cin;
    kc -= kt * p.Cin;
    const int taps = p.kh * p.kw;
    int a_iy0[A_IT], a_ix0[A_IT];
    const float* a_base[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + row0 + it * (NT / 8);
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int ox = mm % p.OW;
        const int t = mm / p.OW;
        const int oy = t % p.OH;
        const int img = t / p.OH;
        a_iy0[it] = oy * p.stride - p.pad;
        a_ix0[it] = ox * p.stride - p.pad;
        a_base[it] = p.x + (size_t)img * p.H * p.W * p.x_cstride;
    }
    const float* b_ptr[B_IT];
    bool b_ok[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + row0 + it * (NT / 8);
        b_ok[it] = n < p.Cout;
        b_ptr[it] = p.w + (size_t)(b_ok[it] ? n : 0) * p.K + seg * 4;
    }

    // STAGES k-chunks are kept in flight in registers (chunk c lives in register set c % STAGES): at
    // batch 1 most launches put one workgroup on a CU, so a single prefetched chunk leaves the
    // pipeline waiting out the whole L2/MALL latency every 32 k.  Every load is unconditional —
    // lanes outside the image / tile read a 16-byte zero page instead — so the loop has no
    // divergent control flow around its loads and hipcc can wait with a counted vmcnt(N).
    v4f a_reg[STAGES][A_IT], b_reg[STAGES][B_IT];
    unsigned shift_mask = 0;   // bit (set * A_IT + it): that staged segment is inside the image (gets in_shift)
    const bool has_shift = p.in_shift != 0.f;
    const float* zero_page = p.zero_page;

    auto load_chunk = [&](const int set, const int kbase) {
        const bool k_ok = kt < taps;
        const int kr = (kt * p.kw_magic) >> 16;     // kt / kw for kt < 8192 (magic = ceil(65536 / kw))
        const int kq = kt - kr * p.kw;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const int iy = a_iy0[it] + kr;
            const int ix = a_ix0[it] + kq;
            const bool ok = a_ok[it] && k_ok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* src = ok ? a_base[it] + ((size_t)iy * p.W + ix) * p.x_cstride + kc : zero_page;
            a_reg[set][it] = ld4(src);
            const unsigned bit = 1u << (set * A_IT + it);
            shift_mask = ok ? (shift_mask | bit) : (shift_mask & ~bit);
        }
        const bool kb_ok = kbase + seg * 4 < p.K;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) b_reg[set][it] = ld4((b_ok[it] && kb_ok) ? b_ptr[it] + kbase : zero_page);
        // advance by one chunk: BK = bk_taps * Cin + bk_rem
        kc += p.bk_rem;
        kt += p.bk_taps;
        const bool wrap = kc >= p.Cin;
        kc -= wrap ? p.Cin : 0;
        kt += wrap ? 1 : 0;
    };

    auto store_chunk = [&](const int set, const int buf) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            v4f v = a_reg[set][it];
            if (has_shift) v += (shift_mask >> (set * A_IT + it)) & 1u ? p.in_shift : 0.f;
            *reinterpret_cast<v4f*>(&sA[(buf * BM + row0 + it * (NT / 8)) * LDS_ROW + seg * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            *reinterpret_cast<v4f*>(&sB[(buf * BN + row0 + it * (NT / 8)) * LDS_ROW + seg * 4]) = b_reg[set][it];
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.K + BK - 1) / BK;
    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);

    auto compute = [&](const int buf) {
        const float* cA = sA + (buf * BM + wm * WTM * 32 + frag_row) * LDS_ROW + frag_k;
        const float* cB = sB + (buf * BN + wn * WTN * 32 + frag_row) * LDS_ROW + frag_k;
#pragma unroll
        for (int st = 0; st < (BK / 8) / WAVES_K; ++st) {
            const int ks = wk + st * WAVES_K;
            v4f af[WTM], bf[WTN];
#pragma unroll
            for (int i = 0; i < WTM; ++i) af[i] = *reinterpret_cast<const v4f*>(cA + i * 32 * LDS_ROW + ks * 8);
#pragma unroll
            for (int j = 0; j < WTN; ++j) bf[j] = *reinterpret_cast<const v4f*>(cB + j * 32 * LDS_ROW + ks * 8);
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
                }
        }
    };

    // prologue: chunks 0 .. STAGES-1 in flight, chunk 0 staged
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nchunks) load_chunk(s, s * BK);
    store_chunk(0, 0);
    __syncthreads();

    // steady state: every iteration refills the register set it has just drained (no conditionals)
    int ch = 0;
    const int n_main = nchunks > STAGES ? (nchunks - STAGES) / STAGES : 0;
    for (int g = 0; g < n_main; ++g) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s, ++ch) {
            load_chunk(s, (ch + STAGES) * BK);   // set s held chunk ch, already staged in LDS
            compute(s & 1);                       // STAGES is even: ch & 1 == s & 1
            store_chunk((s + 1) % STAGES, (s + 1) & 1);
            __syncthreads();
        }
    }
    // tail: the last STAGES .. 2*STAGES-1 chunks (ch is a multiple of STAGES here)
#pragma unroll
    for (int t = 0; t < 2 * STAGES; ++t) {
        const int c = ch + t;
        if (c < nchunks) {
            if (c + STAGES < nchunks) load_chunk(t % STAGES, (c + STAGES) * BK);
            compute(t & 1);
            if (c + 1 < nchunks) store_chunk((t + 1) % STAGES, (t + 1) & 1);
            __syncthreads();
        }
    }

    // ---- K-split reduction across the wk waves of one (wm, wn) -------------------------------
    if (WAVES_K > 1) {
        float* red = smem;  // staging is dead after the loop's last barrier
        if (wk > 0) {
            float* dst = red + ((size_t)((wm * WAVES_N + wn) * (WAVES_K - 1) + (wk - 1)) * WTM * WTN * 16) * 64 + lane;
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[((i * WTN + j) * 16 + r) * 64] = acc[i][j][r];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int s = 0; s < WAVES_K - 1; ++s) {
                const float* src = red + ((size_t)((wm * WAVES_N + wn) * (WAVES_K - 1) + s) * WTM * WTN * 16) * 64 + lane;
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += src[((i * WTN + j) * 16 + r) * 64];
            }
        }
    }
    if (wk != 0) return;

    // ---- epilogue: bias, ReLU / sigmoid, NHWC store at a channel offset ----------------------
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const bool do_relu = (p.flags & FCN_CONV_RELU) != 0;
    const bool do_sig2 = (p.flags & FCN_CONV_SIGMOID2) != 0 && p.y2 != nullptr;
#pragma unroll
    for (int j = 0; j < WTN; ++j) {
        const int n = n0 + (wn * WTN + j) * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float bv = p.bias ? *(const float __attribute__((address_space(1)))*)(p.bias + n) : 0.f;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            const int mrow = m0 + (wm * WTM + i) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mrow + (r & 3) + 8 * (r >> 2);
                if (m < p.M) {
                    float v = acc[i][j][r] + bv;
                    if (do_relu) v = fmaxf(v, 0.f);
                    *(float __attribute__((address_space(1)))*)(p.y + (size_t)m * p.y_cstride + p.y_coffset + n) = v;
                    if (do_sig2)
                        *(float __attribute__((address_space(1)))*)(p.y2 + (size_t)m * p.y2_cstride + p.y2_coffset + n) = 1.f / (1.f + expf(-v));
                }
            }
        }
    }
}

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int STAGES>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K) void conv_fwd_one(ConvP p) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES>::LDS_FLOATS];
    conv_body<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES>(p, blockIdx.x, smem);
}

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int STAGES>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K) void conv_fwd_group(const ConvP* __restrict__ probs, int nprob) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES>::LDS_FLOATS];
    int tile = blockIdx.x;
    int pi = 0, begin = 0;
    while (pi + 1 < nprob && tile >= probs[pi].tile_end) { begin = probs[pi].tile_end; ++pi; }
    const ConvP p = probs[pi];
    conv_body<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES>(p, tile - begin, smem);
}

// ---- host side -------------------------------------------------------------------------------

struct TileCfg { int bm, bn, waves; };
constexpr int kNumCfg = 6;
constexpr TileCfg kCfgs[kNumCfg] = {
    {128, 128, 4},  // 0: 2x2 tiles, 2x2 waves
    {128, 64, 4},   // 1: 2x1 tiles, 2x2 waves
    {64, 64, 4},    // 2: 1x1 tiles, 2x2 waves
    {128, 32, 4},   // 3: 1x1 tiles, 4x1 waves
    {64, 32, 4},    // 4: 1x1 tiles, 2x1 waves, K split 2
    {32, 32, 4},    // 5: 1x1 tiles, 1x1 waves, K split 4
};

int validate(const fcn_conv_desc& d) {
    FCN_REQUIRE(d.x && d.w && d.y, FCN_E_ARG, "conv: null x/w/y");
    FCN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.kh > 0 && d.kw > 0 && d.stride > 0 && d.pad >= 0,
                FCN_E_ARG, "conv: non-positive extent");
    FCN_REQUIRE(d.Cin % 4 == 0 && d.x_cstride % 4 == 0 && d.x_cstride >= d.Cin, FCN_E_ALIGN,
                "conv: Cin (%d) and x_cstride (%d) must be multiples of 4 (pad the input channels)", d.Cin, d.x_cstride);
    FCN_REQUIRE(((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.w & 15) == 0, FCN_E_ALIGN, "conv: x/w must be 16-byte aligned");
    FCN_REQUIRE(d.OH == (d.H + 2 * d.pad - d.kh) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.kw) / d.stride + 1,
                FCN_E_ARG, "conv: OH/OW (%d,%d) do not match floor((H+2p-k)/s)+1", d.OH, d.OW);
    FCN_REQUIRE(d.OH > 0 && d.OW > 0, FCN_E_ARG, "conv: empty output");
    FCN_REQUIRE(d.y_cstride >= d.y_coffset + d.Cout && d.y_coffset >= 0, FCN_E_ARG, "conv: output slice exceeds y_cstride");
    FCN_REQUIRE(d.kh * d.kw < 8192, FCN_E_UNSUPPORTED, "conv: kernel window too large");
    if (d.flags & FCN_CONV_SIGMOID2)
        FCN_REQUIRE(d.y2 && d.y2_cstride >= d.y2_coffset + d.Cout, FCN_E_ARG, "conv: FCN_CONV_SIGMOID2 needs y2");
    FCN_REQUIRE((long long)d.N * d.OH * d.OW < (1ll << 31) && (long long)d.kh * d.kw * d.Cin < (1ll << 31), FCN_E_UNSUPPORTED,
                "conv: problem too large for int32 indexing");
    return 0;
}

void fill(ConvP& p, const fcn_conv_desc& d) {
    p.x = d.x; p.w = d.w; p.bias = d.bias; p.y = d.y; p.y2 = d.y2;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.x_cstride = d.x_cstride;
    p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw; p.pad = d.pad; p.stride = d.stride; p.OH = d.OH; p.OW = d.OW;
    p.y_cstride = d.y_cstride; p.y_coffset = d.y_coffset; p.y2_cstride = d.y2_cstride; p.y2_coffset = d.y2_coffset;
    p.flags = d.flags; p.in_shift = d.in_shift;
    p.M = d.N * d.OH * d.OW;
    p.K = d.kh * d.kw * d.Cin;
    p.kw_magic = (65536 + d.kw - 1) / d.kw;
    p.bk_taps = BK / d.Cin;
    p.bk_rem = BK % d.Cin;
    p.zero_page = nullptr;
}

// Pick the tile shape that minimises (waves of workgroups over 256 CUs) x (MFMA work per workgroup).
int choose_cfg(const ConvP* ps, int n) {
    const char* force = getenv("FCN_CONV_CFG");
    if (force && force[0] >= '0' && force[0] < '0' + kNumCfg) return force[0] - '0';
    int best = 0;
    double best_cost = 1e300;
    for (int c = 0; c < kNumCfg; ++c) {
        long long tiles = 0;
        double work = 0;  // per-workgroup MFMA time, weighted
        for (int i = 0; i < n; ++i) {
            long long t = (long long)cdiv(ps[i].M, kCfgs[c].bm) * cdiv(ps[i].Cout, kCfgs[c].bn);
            tiles += t;
            const int kpad = cdiv(ps[i].K, BK) * BK;
            work += (double)t * kCfgs[c].bm * kCfgs[c].bn * kpad;
        }
        const double per_wg = work / (double)tiles;      // average flops per workgroup
        const double rounds = (double)((tiles + 255) / 256);
        // smaller tiles re-read operands more often and pay more barrier/LDS overhead per flop
        const double overhead = 1.0 + 24.0 / kCfgs[c].bm + 24.0 / kCfgs[c].bn;
        const double cost = rounds * per_wg * overhead;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

// tile configurations: X(index, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, STAGES)
#define FCN_CONV_CONFIGS(X) \
    X(0, 2, 2, 2, 2, 1, 2)  \
    X(1, 2, 1, 2, 2, 1, 4)  \
    X(2, 1, 1, 2, 2, 1, 4)  \
    X(3, 1, 1, 4, 1, 1, 4)  \
    X(4, 1, 1, 2, 1, 2, 6)  \
    X(5, 1, 1, 1, 1, 4, 8)

int plan_tiles_cfg(int cfg, ConvP* ps, int n) {
    const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        ps[i].tiles_m = cdiv(ps[i].M, bm);
        ps[i].tiles_n = cdiv(ps[i].Cout, bn);
        total += ps[i].tiles_m * ps[i].tiles_n;
        ps[i].tile_end = total;
    }
    return total;
}

void launch_one_cfg(int cfg, const ConvP& p, int total, hipStream_t st) {
    switch (cfg) {
#define X(I, A, B, C_, D, E, F)                                                                                          \
    case I:                                                                                                              \
        static_assert(Cfg<A, B, C_, D, E, F>::BM == kCfgs[I].bm && Cfg<A, B, C_, D, E, F>::BN == kCfgs[I].bn, "kCfgs out of sync"); \
        hipLaunchKernelGGL((conv_fwd_one<A, B, C_, D, E, F>), dim3(total), dim3(Cfg<A, B, C_, D, E, F>::NT), 0, st, p);   \
        break;
        FCN_CONV_CONFIGS(X)
#undef X
    }
}

void launch_group_cfg(int cfg, const ConvP* d_ps, int n, int total, hipStream_t st) {
    switch (cfg) {
#define X(I, A, B, C_, D, E, F)                                                                                                \
    case I:                                                                                                                    \
        hipLaunchKernelGGL((conv_fwd_group<A, B, C_, D, E, F>), dim3(total), dim3(Cfg<A, B, C_, D, E, F>::NT), 0, st, d_ps, n); \
        break;
        FCN_CONV_CONFIGS(X)
#undef X
    }
}

}  // namespace

extern "C" {

int fcn_conv2d_fwd_f32(const fcn_conv_desc* h_desc, fcn_stream_t s) {
    FCN_REQUIRE(h_desc, FCN_E_ARG, "fcn_conv2d_fwd_f32: null desc");
    int rc = validate(*h_desc);
    if (rc) return rc;
    ConvP p;
    fill(p, *h_desc);
    p.zero_page = zero_page_for_current_device(&rc);
    if (rc) return rc;
    const int cfg = choose_cfg(&p, 1);
    const int total = plan_tiles_cfg(cfg, &p, 1);
    hipStream_t st = as_stream(s);
    launch_one_cfg(cfg, p, total, st);
    FCN_LAUNCH_CHECK("conv_fwd_one");
    return 0;
}

size_t fcn_conv2d_group_workspace_bytes(int n) { return sizeof(ConvP) * (size_t)(n > 0 ? n : 0); }

int fcn_conv2d_group_prepare(const fcn_conv_desc* h_descs, int n, void* d_workspace, fcn_conv_group* h_out) {
    FCN_REQUIRE(h_descs && h_out && d_workspace && n > 0 && n <= 16, FCN_E_ARG, "fcn_conv2d_group_prepare: need 1..16 problems, workspace, out");
    ConvP ps[16];
    int zrc = 0;
    const float* zp = zero_page_for_current_device(&zrc);
    if (zrc) return zrc;
    for (int i = 0; i < n; ++i) {
        int rc = validate(h_descs[i]);
        if (rc) return rc;
        fill(ps[i], h_descs[i]);
        ps[i].zero_page = zp;
    }
    const int cfg = choose_cfg(ps, n);
    const int total = plan_tiles_cfg(cfg, ps, n);
    FCN_HIP(hipMemcpy(d_workspace, ps, sizeof(ConvP) * n, hipMemcpyHostToDevice));
    h_out->d_probs = d_workspace;
    h_out->n = n;
    h_out->cfg = cfg;
    h_out->total_tiles = total;
    return 0;
}

int fcn_conv2d_fwd_group_f32(const fcn_conv_group* g, fcn_stream_t s) {
    FCN_REQUIRE(g && g->d_probs && g->n > 0 && g->total_tiles > 0, FCN_E_ARG, "fcn_conv2d_fwd_group_f32: unprepared group");
    const ConvP* d_ps = reinterpret_cast<const ConvP*>(g->d_probs);
    hipStream_t st = as_stream(s);
    FCN_REQUIRE(g->cfg >= 0 && g->cfg < kNumCfg, FCN_E_ARG, "fcn_conv2d_fwd_group_f32: bad cfg %d", g->cfg);
    launch_group_cfg(g->cfg, d_ps, g->n, g->total_tiles, st);
    FCN_LAUNCH_CHECK("conv_fwd_group");
    return 0;
}

}  // extern "C"
