// Convolution forward for gfx950: NHWC implicit GEMM on the f32-input matrix cores.
//
// Stands in for Caffe's ConvolutionLayer::Forward (+ the in-place ReLU, the Sigmoid head and the
// Power(shift) input transform that follow or precede it) as executed by net.forward() in the
// reference (scripts/fcn_object_detector.py:87) over models/deploy.prototxt:8-2176.
//
//   out[m][n] = bias[n] + sum_k A[m][k] * Wt[n][k]      m = (img, oy, ox)   n = output channel
//   k = (r*kw + q)*Cin + c ,  A[m][k] = x[img][oy*s-p+r][ox*s-p+q][c]  (0 outside the image)
//
// Data layout: activations NHWC (channel-contiguous, so an A-row segment of 4 consecutive k is one
// 16-byte load), weights OHWI = [Cout][kh][kw][Cin] (a Wt row is K contiguous floats).  Both tiles
// are staged global -> registers -> LDS as [row][BK k + 4 pad] floats; the pad makes the
// ds_read_b128 fragment reads bank-conflict free (row stride = 4 mod 64 dwords: 16 rows hit 16
// distinct 4-dword slots of the 64-bank row).  Each wave accumulates 32x32 output tiles with
// v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the chip's f32 peak).  A k-step of 8 uses
// one b128 read per operand: lanes 0-31 hold k+0..3, lanes 32-63 hold k+4..7, and MFMA j consumes
// element j of both fragments, so A and B see the same k permutation.
//
// Pipeline (per workgroup, chunk = BK consecutive k):
//   * STAGES chunks are in flight global -> registers (chunk c in register set c % STAGES); every
//     load is unconditional (masked lanes read a zero page) so hipcc waits with a counted vmcnt(N);
//   * three LDS buffers: iteration c reads the fragments of chunk c+1 into registers (consumed by the
//     MFMAs of iteration c+1), runs the MFMAs of chunk c from registers, and writes chunk c+2 — the
//     LDS round trip and the one barrier per chunk hide behind the MFMA stream;
//   * small-M layers (28x28 grid, M = 784) cannot fill 256 CUs with large tiles, so the tile shape is
//     a template parameter chosen per launch, down to one 32x32 tile per workgroup whose 4 waves
//     split each chunk's k range (WAVES_K) and reduce through LDS at the end.
// Several independent problems (the branches of an inception module) share ONE launch
// (fcn_conv2d_fwd_group_f32).
#include <mutex>

#include "common.h"

using namespace fcn;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace fcn {

// one 16-byte zero page per device (what out-of-image / out-of-tile lanes load), allocated by fcn_init
const float* zero_page_for_current_device(int* rc) {
    static std::mutex mu;
    static const float* pages[64] = {nullptr};
    int dev = 0;
    *rc = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        *rc = set_err(FCN_E_STATE, "conv: no current device");
        return nullptr;
    }
    std::lock_guard<std::mutex> lock(mu);
    if (!pages[dev]) {
        void* ptr = nullptr;
        hipError_t e = hipMalloc(&ptr, 256);
        if (e == hipSuccess) e = hipMemset(ptr, 0, 256);
        if (e != hipSuccess) {
            *rc = set_err(-(int)e, "conv: zero page allocation failed: %s", hipGetErrorString(e));
            return nullptr;
        }
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
    const float* zero_page;               // 16 zero bytes in HBM: what out-of-image / out-of-tile lanes load
};

constexpr int kTapSlots = 64;    // filter taps (kh*kw <= 63) + sentinels

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int STAGES, bool PF>
struct Cfg {
    static constexpr int BM = 32 * WTM * WAVES_M;
    static constexpr int BN = 32 * WTN * WAVES_N;
    static constexpr int NT = 64 * WAVES_M * WAVES_N * WAVES_K;
    static constexpr int SEGS = BK / 4;                 // 16-byte segments per staged row
    static constexpr int ROWS_PER_PASS = NT / SEGS;     // rows the workgroup stages per load instruction
    static constexpr int A_IT = BM / ROWS_PER_PASS;
    static constexpr int B_IT = BN / ROWS_PER_PASS;
    static constexpr int LDS_ROW = BK + 4;              // floats per staged row
    static constexpr int KS = BK / 8 / WAVES_K;         // k-steps of 8 each wave runs per chunk
    static constexpr int NBUF = 3;
    static constexpr int STAGE_FLOATS = NBUF * (BM + BN) * LDS_ROW;
    static constexpr int RED_FLOATS = WAVES_M * WAVES_N * (WAVES_K - 1) * WTM * WTN * 16 * 64;
    static constexpr int STAGE_OR_RED_FLOATS = STAGE_FLOATS > RED_FLOATS ? STAGE_FLOATS : RED_FLOATS;
    static constexpr int LDS_FLOATS = STAGE_OR_RED_FLOATS + 2 * kTapSlots;   // + the tap table (int2 per slot)
    static_assert(BK % 8 == 0 && (BK / 8) % WAVES_K == 0, "each wave needs whole k-steps of a chunk");
    static_assert(NT % SEGS == 0 && BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile rows must divide over the threads");
    static_assert(A_IT >= 1 && B_IT >= 1, "tile too small for the workgroup");
    static_assert(WAVES_K == 1 || WAVES_K == 2 || WAVES_K == 4, "K split over 1, 2 or 4 waves");
    static_assert(STAGES >= 4 && STAGES <= 8 && STAGES % 2 == 0, "4, 6 or 8 k-chunks in flight (even: fragment parity is static)");
    static_assert(STAGES * A_IT <= 32, "shift_mask holds one bit per staged A segment");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
};

// explicit global-address-space accesses: a pointer picked from a struct in memory (group launch) or
// selected against the zero page would otherwise be "generic" and compile to flat_load, which cannot be
// waited on with a counted vmcnt
typedef const v4f __attribute__((address_space(1))) * gv4f_ptr;
typedef float __attribute__((address_space(1))) * gf_ptr;
__device__ __forceinline__ v4f ld4(const float* p) { return *(gv4f_ptr)(p); }

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int STAGES, bool PF>
__device__ __forceinline__ void conv_body(const ConvP& p, int tile, float* smem) {
    using C = Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, PF>;
    constexpr int BM = C::BM, BN = C::BN, NT = C::NT, A_IT = C::A_IT, B_IT = C::B_IT, SEGS = C::SEGS, RPP = C::ROWS_PER_PASS;
    constexpr int LDS_ROW = C::LDS_ROW, KS = C::KS, NBUF = C::NBUF;
    constexpr int BUF_FLOATS = (BM + BN) * LDS_ROW;

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

    // ---- tap table: (kr, kq) and the element offset of every filter tap, plus "outside" sentinels ----
    // One ds_read_b64 per chunk replaces the divisions / multiplies of the k -> (r, q, c) decode.
    int2* s_tap = reinterpret_cast<int2*>(smem + C::STAGE_OR_RED_FLOATS);   // [kTapSlots]
    const int taps = p.kh * p.kw;
    for (int t = tid; t < kTapSlots; t += NT) {
        int2 e;
        if (t < taps) {
            const int kr = t / p.kw, kq = t - kr * p.kw;
            e.x = (kr << 16) | kq;
            e.y = (kr * p.W + kq) * p.x_cstride;
        } else {
            e.x = 0x4000 << 16;   // row far below the image: every bounds test fails
            e.y = 0;
        }
        s_tap[t] = e;
    }

    // ---- per-thread loader state -------------------------------------------------------------
    const int seg = tid % SEGS;     // which 16-byte segment of the BK-float k-chunk
    const int row0 = tid / SEGS;    // first staged row of this thread; further rows every RPP
    // k position of this thread's segment, kept as (tap, channel) and advanced by BK per chunk without branches
    int kc = seg * 4;
    int kt = kc / p.Cin;
    kc -= kt * p.Cin;
    const int bk_taps = BK / p.Cin, bk_rem = BK - bk_taps * p.Cin;
    int a_iy0[A_IT], a_ix0[A_IT], a_off[A_IT];   // window origin and its element offset (32-bit: validated on the host)
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + row0 + it * RPP;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.OW;
        const int t = mm / p.OW;
        const int oy = t % p.OH;
        const int img = t / p.OH;
        a_iy0[it] = ok ? oy * p.stride - p.pad : -(1 << 20);   // rows past M never pass the bounds test
        a_ix0[it] = ox * p.stride - p.pad;
        a_off[it] = ((img * p.H + a_iy0[it]) * p.W + a_ix0[it]) * p.x_cstride;
    }
    int b_off[B_IT];   // element offset of this thread's segment in weight row n; negative = row past Cout
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int n = n0 + row0 + it * RPP;
        b_off[it] = n < p.Cout ? n * p.K + seg * 4 : -1;
    }
    int kb = seg * 4;   // this segment's k index in the current chunk (weights are zero past K)

    v4f a_reg[STAGES][A_IT], b_reg[STAGES][B_IT];
    unsigned shift_mask = 0;   // bit (set * A_IT + it): that staged segment is inside the image (gets in_shift)
    const bool has_shift = p.in_shift != 0.f;
    const float* zero_page = p.zero_page;
    __syncthreads();           // tap table visible

    const bool dbg_noload = (p.flags & 0x200) != 0, dbg_nomfma = (p.flags & 0x100) != 0, dbg_nolds = (p.flags & 0x400) != 0;
    auto load_chunk = [&](const int set) {
        if (dbg_noload) return;
        const int2 tp = s_tap[min(kt, kTapSlots - 1)];
        const int kr = tp.x >> 16, kq = tp.x & 0xffff;
        const int koff = tp.y + kc;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = (unsigned)(a_iy0[it] + kr) < (unsigned)p.H && (unsigned)(a_ix0[it] + kq) < (unsigned)p.W;
            a_reg[set][it] = ld4(ok ? p.x + (a_off[it] + koff) : zero_page);
            if (has_shift) {
                const unsigned bit = 1u << (set * A_IT + it);
                shift_mask = ok ? (shift_mask | bit) : (shift_mask & ~bit);
            }
        }
        const bool kb_ok = kb < p.K;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            b_reg[set][it] = ld4((kb_ok && b_off[it] >= 0) ? p.w + b_off[it] : zero_page);
            b_off[it] += b_off[it] >= 0 ? BK : 0;
        }
        kb += BK;
        // advance by one chunk: BK = bk_taps * Cin + bk_rem
        kc += bk_rem;
        kt += bk_taps;
        const bool wrap = kc >= p.Cin;
        kc -= wrap ? p.Cin : 0;
        kt += wrap ? 1 : 0;
    };

    auto store_chunk = [&](const int set, const int buf) {
        if (dbg_nolds) return;
        float* sA = smem + buf * BUF_FLOATS;
        float* sB = sA + BM * LDS_ROW;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            v4f v = a_reg[set][it];
            if (has_shift) v += (shift_mask >> (set * A_IT + it)) & 1u ? p.in_shift : 0.f;
            *reinterpret_cast<v4f*>(&sA[(row0 + it * RPP) * LDS_ROW + seg * 4]) = v;
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) *reinterpret_cast<v4f*>(&sB[(row0 + it * RPP) * LDS_ROW + seg * 4]) = b_reg[set][it];
    };

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = (p.K + BK - 1) / BK;
    // this lane's fragment origin inside a staged tile: row (lane & 31), k offset 4 * (lane >> 5), wave's k-steps
    const int frag_a = (wm * WTM * 32 + (lane & 31)) * LDS_ROW + 4 * (lane >> 5) + wk * KS * 8;
    const int frag_b = (BM + wn * WTN * 32 + (lane & 31)) * LDS_ROW + 4 * (lane >> 5) + wk * KS * 8;

    // PF: fragments of the chunk being multiplied and of the next one live in registers, so the LDS round trip
    // of chunk c+1 hides behind the MFMAs of chunk c.  Large tiles (already >= 2048 MFMA cycles per barrier)
    // read their fragments just in time instead and keep the registers for accumulators.
    v4f af[PF ? 2 : 1][PF ? KS : 1][WTM], bf[PF ? 2 : 1][PF ? KS : 1][WTN];
    auto read_frags = [&](const int par, const int buf) {
        if (!PF || dbg_nolds) return;
        const float* base = smem + buf * BUF_FLOATS;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) af[par][st][i] = *reinterpret_cast<const v4f*>(base + frag_a + i * 32 * LDS_ROW + st * 8);
#pragma unroll
            for (int j = 0; j < WTN; ++j) bf[par][st][j] = *reinterpret_cast<const v4f*>(base + frag_b + j * 32 * LDS_ROW + st * 8);
        }
    };
    auto mfma_chunk = [&](const int par, const int buf_cur) {
        if (dbg_nomfma) return;
        const float* base = smem + buf_cur * BUF_FLOATS;
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            if (!PF) {
#pragma unroll
                for (int i = 0; i < WTM; ++i) af[0][0][i] = *reinterpret_cast<const v4f*>(base + frag_a + i * 32 * LDS_ROW + st * 8);
#pragma unroll
                for (int j = 0; j < WTN; ++j) bf[0][0][j] = *reinterpret_cast<const v4f*>(base + frag_b + j * 32 * LDS_ROW + st * 8);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WTM; ++i)
#pragma unroll
                    for (int j = 0; j < WTN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[PF ? par : 0][PF ? st : 0][i][e], bf[PF ? par : 0][PF ? st : 0][j][e],
                                                                          acc[i][j], 0, 0, 0);
        }
    };

    // ---- prologue: chunks 0 .. STAGES-1 in flight; chunks 0 and 1 staged; fragments of chunk 0 read ----
#pragma unroll
    for (int s = 0; s < STAGES; ++s)
        if (s < nchunks) load_chunk(s);
    store_chunk(0, 0);
    if (1 < nchunks) store_chunk(1, 1);
    __syncthreads();
    read_frags(0, 0);

    // LDS buffer of chunk c is c % 3, tracked incrementally (uniform scalars)
    int buf_cur = 0;           // buffer holding chunk c
    int buf_next = 1 % NBUF;   // buffer holding chunk c + 1
    int buf_fill = 2 % NBUF;   // buffer that receives chunk c + 2
    auto rotate = [&]() {
        buf_cur = buf_next;
        buf_next = buf_fill;
        buf_fill = buf_fill + 1 == NBUF ? 0 : buf_fill + 1;
    };

    // ---- steady state: no conditionals; every iteration refills the register set it drained two chunks ago ----
    int ch = 0;
    const int n_main = nchunks > STAGES ? (nchunks - STAGES) / STAGES : 0;
    for (int g = 0; g < n_main; ++g) {
#pragma unroll
        for (int s = 0; s < STAGES; ++s, ++ch) {
            load_chunk(s);                               // chunk ch + STAGES into the set chunk ch left
            read_frags((s + 1) & 1, buf_next);           // fragments of chunk ch + 1 (used next iteration)
            mfma_chunk(s & 1, buf_cur);                  // chunk ch
            store_chunk((s + 2) % STAGES, buf_fill);     // chunk ch + 2 -> LDS
            rotate();
            __syncthreads();
        }
    }
    // ---- tail: the last STAGES .. 2*STAGES-1 chunks (ch is a multiple of STAGES here) ----
#pragma unroll
    for (int t = 0; t < 2 * STAGES; ++t) {
        const int c = ch + t;
        if (c < nchunks) {
            if (c + STAGES < nchunks) load_chunk(t % STAGES);
            if (c + 1 < nchunks) read_frags((t + 1) & 1, buf_next);
            mfma_chunk(t & 1, buf_cur);
            if (c + 2 < nchunks) store_chunk((t + 2) % STAGES, buf_fill);
            rotate();
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
                    *(gf_ptr)(p.y + (size_t)m * p.y_cstride + p.y_coffset + n) = v;
                    if (do_sig2) *(gf_ptr)(p.y2 + (size_t)m * p.y2_cstride + p.y2_coffset + n) = 1.f / (1.f + expf(-v));
                }
            }
        }
    }
}

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int STAGES, bool PF>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K) void conv_fwd_group(const ConvP* __restrict__ probs, int nprob) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, PF>::LDS_FLOATS];
    int tile = blockIdx.x;
    int pi = 0, begin = 0;
    while (pi + 1 < nprob && tile >= probs[pi].tile_end) { begin = probs[pi].tile_end; ++pi; }
    const ConvP p = probs[pi];
    conv_body<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, PF>(p, tile - begin, smem);
}

template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int STAGES, bool PF>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N * WAVES_K) void conv_fwd_one(ConvP p) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, PF>::LDS_FLOATS];
    conv_body<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, PF>(p, blockIdx.x, smem);
}

// ---- host side -------------------------------------------------------------------------------

// tile configurations: X(index, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, STAGES, fragment prefetch)
#define FCN_CONV_CONFIGS(X)             \
    X(0, 2, 2, 2, 2, 1, 32, 4, false)   \
    X(1, 2, 1, 2, 2, 1, 32, 4, false)   \
    X(2, 1, 1, 2, 2, 1, 32, 4, true)    \
    X(3, 1, 1, 4, 1, 1, 32, 4, true)    \
    X(4, 1, 1, 2, 1, 2, 64, 4, true)    \
    X(5, 1, 1, 1, 1, 4, 128, 4, true)   \
    X(6, 1, 1, 2, 2, 1, 64, 4, false)   \
    X(7, 1, 1, 1, 1, 4, 64, 4, true)

struct TileCfg { int bm, bn, bk, mfma_per_barrier; };
constexpr int kNumCfg = 8;
constexpr TileCfg kCfgs[kNumCfg] = {
#define X(I, A, B, C_, D, E, F, G, H) {Cfg<A, B, C_, D, E, F, G, H>::BM, Cfg<A, B, C_, D, E, F, G, H>::BN, F, A * B * 4 * Cfg<A, B, C_, D, E, F, G, H>::KS},
    FCN_CONV_CONFIGS(X)
#undef X
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
    FCN_REQUIRE(d.kh * d.kw < kTapSlots, FCN_E_UNSUPPORTED, "conv: kernel window %dx%d exceeds %d taps", d.kh, d.kw, kTapSlots - 1);
    FCN_REQUIRE((long long)d.N * d.H * d.W * d.x_cstride < (1ll << 31) && (long long)d.Cout * d.kh * d.kw * d.Cin < (1ll << 31),
                FCN_E_UNSUPPORTED, "conv: tensor too large for 32-bit element offsets");
    if (d.flags & FCN_CONV_SIGMOID2)
        FCN_REQUIRE(d.y2 && d.y2_cstride >= d.y2_coffset + d.Cout, FCN_E_ARG, "conv: FCN_CONV_SIGMOID2 needs y2");
    FCN_REQUIRE((long long)d.N * d.OH * d.OW < (1ll << 31), FCN_E_UNSUPPORTED, "conv: problem too large for int32 indexing");
    return 0;
}

void fill(ConvP& p, const fcn_conv_desc& d, const float* zero_page) {
    p.x = d.x; p.w = d.w; p.bias = d.bias; p.y = d.y; p.y2 = d.y2;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.x_cstride = d.x_cstride;
    p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw; p.pad = d.pad; p.stride = d.stride; p.OH = d.OH; p.OW = d.OW;
    p.y_cstride = d.y_cstride; p.y_coffset = d.y_coffset; p.y2_cstride = d.y2_cstride; p.y2_coffset = d.y2_coffset;
    p.flags = d.flags; p.in_shift = d.in_shift;
    p.M = d.N * d.OH * d.OW;
    p.K = d.kh * d.kw * d.Cin;
    p.tiles_m = p.tiles_n = p.tile_end = 0;
    p.zero_page = zero_page;
}

// Cost model fitted to tools/conv_sweep.py on MI355X: a workgroup pays a fixed pipeline fill plus, per
// k-chunk, its MFMA time and one barrier / LDS round trip; workgroups run in rounds of 256 (one per CU).
int choose_cfg(const ConvP* ps, int n) {
    const char* force = getenv("FCN_CONV_CFG");
    if (force && force[0] >= '0' && force[0] < '0' + kNumCfg) return force[0] - '0';
    int best = 0;
    double best_cost = 1e300;
    for (int c = 0; c < kNumCfg; ++c) {
        long long tiles = 0;
        double longest = 0;   // cycles of the slowest workgroup
        double total = 0;     // cycles summed over workgroups
        for (int i = 0; i < n; ++i) {
            const long long t = (long long)cdiv(ps[i].M, kCfgs[c].bm) * cdiv(ps[i].Cout, kCfgs[c].bn);
            const int chunks = cdiv(ps[i].K, kCfgs[c].bk);
            const double per_chunk = 64.0 * kCfgs[c].mfma_per_barrier + 350.0;   // MFMA issue + sync overhead
            const double wg = 4000.0 + chunks * per_chunk;
            tiles += t;
            total += t * wg;
            if (wg > longest) longest = wg;
        }
        // one workgroup per CU per round; with fewer tiles than CUs the slowest tile sets the time
        const double rounds = (double)((tiles + 255) / 256);
        const double avg = total / (double)tiles;
        const double cost = rounds <= 1.0 ? longest : rounds * avg + (longest - avg);
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

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
#define X(I, A, B, C_, D, E, F, G, H)                                                                                           \
    case I:                                                                                                                     \
        hipLaunchKernelGGL((conv_fwd_one<A, B, C_, D, E, F, G, H>), dim3(total), dim3(Cfg<A, B, C_, D, E, F, G, H>::NT), 0, st, p); \
        break;
        FCN_CONV_CONFIGS(X)
#undef X
    }
}

void launch_group_cfg(int cfg, const ConvP* d_ps, int n, int total, hipStream_t st) {
    switch (cfg) {
#define X(I, A, B, C_, D, E, F, G, H)                                                                                                    \
    case I:                                                                                                                              \
        hipLaunchKernelGGL((conv_fwd_group<A, B, C_, D, E, F, G, H>), dim3(total), dim3(Cfg<A, B, C_, D, E, F, G, H>::NT), 0, st, d_ps, n); \
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
    const float* zp = zero_page_for_current_device(&rc);
    if (rc) return rc;
    ConvP p;
    fill(p, *h_desc, zp);
    const int cfg = choose_cfg(&p, 1);
    const int total = plan_tiles_cfg(cfg, &p, 1);
    launch_one_cfg(cfg, p, total, as_stream(s));
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
        fill(ps[i], h_descs[i], zp);
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
    FCN_REQUIRE(g->cfg >= 0 && g->cfg < kNumCfg, FCN_E_ARG, "fcn_conv2d_fwd_group_f32: bad cfg %d", g->cfg);
    launch_group_cfg(g->cfg, reinterpret_cast<const ConvP*>(g->d_probs), g->n, g->total_tiles, as_stream(s));
    FCN_LAUNCH_CHECK("conv_fwd_group");
    return 0;
}

}  // extern "C"
