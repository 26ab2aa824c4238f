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
// 16-byte load), weights OHWI = [Cout][kh][kw][Cin] (a Wt row is K contiguous floats).  Each wave
// accumulates 32x32 output tiles with v_mfma_f32_32x32x2_f32 (exact f32, 64 FLOP/clk/SIMD = the
// chip's f32 peak).  A k-step of 8 uses one ds_read_b128 per operand: lanes 0-31 hold k+0..3, lanes
// 32-63 hold k+4..7, and MFMA j consumes element j of both fragments, so A and B see the same k
// permutation.
//
// Pipeline (per workgroup, chunk = BK consecutive k):
//   * both tiles are staged HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction,
//     no VGPR round trip, no ds_write: LDS stores run at ~1/3 of the LDS read rate and were the limiter
//     of the register-staged version) into a ring of NBUF slots, NBUF-1 chunks in flight;
//   * every lane always loads — lanes outside the image / tile / K read a 16-byte zero page — so halo and
//     tail arrive as zeros and each chunk is a fixed number of instructions: one counted
//     s_waitcnt vmcnt(N) + one raw s_barrier per chunk, never vmcnt(0) inside the loop;
//   * LDS rows are unpadded (LDS-DMA writes lane-linearly); bank conflicts are avoided by XOR-swizzling
//     the 16-byte slots of a row with the row index, applied to the source address on the way in and
//     to the slot index on the way out;
//   * small tiles keep the fragments of the next chunk in registers (PF) so the LDS read latency hides
//     behind the MFMAs of the current chunk;
//   * small-M layers (28x28 grid, M = 784) cannot fill 256 CUs with large tiles, so the tile shape is
//     a template parameter chosen per launch, down to one 32x32 tile per workgroup whose 4 waves
//     split each chunk's k range (WAVES_K) and reduce through LDS at the end.
// Several independent problems (the branches of an inception module) share ONE launch
// (fcn_conv2d_fwd_group_f32).
#include <cstddef>
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <type_traits>

#include "common.h"
#include "conv_common.h"

using namespace fcn;

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

// work items (16 bytes of one output pixel) a pooling thread takes: one while the pooling is small - its workgroups are the launch's first
// and a convolution tile shares their CUs, so they should be gone quickly (the 28 x 28 inception poolings: A levels 6.7 -> 6.1 us) - two
// for the large ones, where twice the workgroups cost more than they save (56 x 56: 7.2 -> 7.7 us with one)
constexpr int kPoolItemsPerThread = 2;
constexpr int kPoolSmallItems = 128 * 1024;

// MAX pooling riding in a convolution launch (Caffe semantics as in pointwise.hip: window clipped to the image, strict
// '>' so the first maximum in raster order wins).  One work item = 4 channels of one output pixel.
template <typename T, int NT>
__device__ __forceinline__ void pool_body(const PoolP& q, int wg) {
    constexpr int EPS = 16 / (int)sizeof(T);      // channels per 16-byte work item
    typedef T vec_t __attribute__((ext_vector_type(EPS)));
    const int cg = q.C / EPS;
    const T* x = reinterpret_cast<const T*>(q.x);
    T* y = reinterpret_cast<T*>(q.y);
    const int ipt = q.items <= kPoolSmallItems ? 1 : kPoolItemsPerThread;      // (uniform; the host sizes wg_end the same way)
#pragma unroll
    for (int it = 0; it < kPoolItemsPerThread; ++it) {
        if (it >= ipt) return;
        const int t = (wg * ipt + it) * NT + (int)threadIdx.x;
        if (t >= q.items) return;
        const int pix = t / cg;
        const int g = t - pix * cg;
        const int row = pix / q.OW;
        const int ox = pix - row * q.OW;
        const int n = row / q.OH;
        const int oy = row - n * q.OH;
        int hs = oy * q.stride - q.pad, ws = ox * q.stride - q.pad;
        const int he = min(hs + q.k, q.H), we = min(ws + q.k, q.W);
        hs = max(hs, 0);
        ws = max(ws, 0);
        const T* xb = x + (size_t)n * q.H * q.W * q.x_cstride + g * EPS;
        vec_t m;
        int mi[EPS];
#pragma unroll
        for (int e = 0; e < EPS; ++e) {
            m[e] = sizeof(T) == 2 ? (T)-65504.f : (T)-3.402823466e+38f;      // lowest finite value of the element type
            mi[e] = -1;
        }
        if (q.k == 3) {
            // 3x3 windows (every pooling of the reference nets that rides in a convolution launch): all nine loads in flight before the
            // first compare.  The loop below waits for each load in turn - 18 dependent round trips per thread made the 92 pooling
            // workgroups of an inception level (2 us of work) the LONGEST workgroups of a 5 us launch (round 4).  Taps outside the image
            // load the nearest pixel inside (a valid address) and are skipped by the compares, so the maximum and the first-maximum
            // index are exactly the loop's.
            const int y0 = oy * q.stride - q.pad, x0 = ox * q.stride - q.pad;
            vec_t v[9];
            int id[9];
            bool in[9];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const int iy = min(max(y0 + r, 0), q.H - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ix = min(max(x0 + c, 0), q.W - 1);
                    in[3 * r + c] = (int)((unsigned)(y0 + r) < (unsigned)q.H) & (int)((unsigned)(x0 + c) < (unsigned)q.W);
                    id[3 * r + c] = iy * q.W + ix;
                    v[3 * r + c] = *(const vec_t*)(xb + (size_t)id[3 * r + c] * q.x_cstride);
                }
            }
#pragma unroll
            for (int t9 = 0; t9 < 9; ++t9)
#pragma unroll
                for (int e = 0; e < EPS; ++e)
                    if ((int)in[t9] & (int)(v[t9][e] > m[e])) { m[e] = v[t9][e]; mi[e] = id[t9]; }
        } else
        for (int iy = hs; iy < he; ++iy)
            for (int ix = ws; ix < we; ++ix) {
                const vec_t v = *(const vec_t*)(xb + ((size_t)iy * q.W + ix) * q.x_cstride);
                const int id = iy * q.W + ix;
#pragma unroll
                for (int e = 0; e < EPS; ++e)
                    if (v[e] > m[e]) { m[e] = v[e]; mi[e] = id; }
            }
        *(vec_t*)(y + (size_t)pix * q.y_cstride + q.y_coffset + g * EPS) = m;
        if (q.idx) {
#pragma unroll
            for (int e = 0; e < EPS; e += 4)
                *(int4*)(q.idx + (size_t)pix * q.C + g * EPS + e) = make_int4(mi[e], mi[e + 1], mi[e + 2], mi[e + 3]);
        }
    }
}

// Ring slots >= 16 select SPLIT ROLES (the value minus 16 is the ring depth): the workgroup has twice the waves, the first
// half multiplies (fragment reads + MFMAs, no vector-memory work), the second half loads (address arithmetic + LDS-DMA, no
// MFMAs).  With one wave per SIMD the loader's ~45 VALU instructions per chunk do not hide behind the wave's own MFMAs (the
// launch costs MFMA time PLUS loader time: 4a_A 8.4 us with, 6.8 us without its MFMAs); with a loader wave beside a
// multiplier wave on every SIMD the vector pipe and the matrix pipe really run side by side.
template <int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int NBUF_, bool PF>
struct Cfg {
    static constexpr bool ROLES = NBUF_ >= 16;
    // Ring slots >= 32 (round 4): split roles with TWICE the loading waves - the first NW of them stage the A rows, the second NW
    // the B rows, one LDS-DMA piece per wave and chunk where the plain split gives every loading wave an A and a B piece.
    static constexpr int LOADX = NBUF_ >= 32 ? 2 : 1;
    static constexpr int NBUF = NBUF_ % 16;
    static constexpr int BM = 32 * WTM * WAVES_M;
    static constexpr int BN = 32 * WTN * WAVES_N;
    static constexpr int NW = WAVES_M * WAVES_N * WAVES_K;   // multiplying waves per workgroup (loading waves: NW, or 2 NW with LOADX = 2)
    static constexpr int NT = 64 * NW * (ROLES ? 1 + LOADX : 1);
    static constexpr int SEGS = BK / 4;                 // 16-byte slots per staged row
    static constexpr int RPI = 256 / BK;                // rows one LDS-DMA wave-instruction (1 KiB) fills
    static constexpr int STEP = RPI * NW;               // rows between two consecutive instructions of one wave
    static constexpr int IA = BM / STEP;                // A-row instructions per wave per chunk
    static constexpr int IB = BN / STEP;                // B-row instructions per wave per chunk
    static constexpr int INST = LOADX == 2 ? IA : IA + IB;   // LDS-DMA instructions one loading wave issues per chunk
    static_assert(LOADX == 1 || IA == IB, "doubled loaders: the A waves and the B waves issue the same number of pieces");
    static constexpr int D = NBUF - 1;                  // chunks in flight
    static constexpr int KS = BK / 8 / WAVES_K;         // k-steps of 8 each wave runs per chunk
    static constexpr int BUF_FLOATS = (BM + BN) * BK;   // one ring slot: A rows then B rows, unpadded
    static constexpr int RING_FLOATS = NBUF * BUF_FLOATS;
    static constexpr int EPI_FLOATS = NW * WTM * WTN * 1024;   // every wave parks its 32x32 accumulator tiles for the epilogue
    static constexpr int LDS_FLOATS = RING_FLOATS > EPI_FLOATS ? RING_FLOATS : EPI_FLOATS;
    static_assert(BK == 32 || BK == 64, "row swizzle is defined for 8 or 16 slots per row");
    static_assert((BK / 8) % WAVES_K == 0, "each wave needs whole k-steps of a chunk");
    static_assert(STEP % 16 == 0, "a lane's rows must agree mod 16 so that its swizzle (and k position) is the same for all of them");
    static_assert(BM % STEP == 0 && BN % STEP == 0, "A / B rows must split into whole wave-instructions");
    static_assert(WAVES_K == 1 || WAVES_K == 2 || WAVES_K == 4 || WAVES_K == 8, "K split over 1, 2, 4 or 8 waves");
    static_assert(WTM <= 2 && WTN <= 2, "fragment reads are written out for at most 2x2 MFMA tiles per wave");
    static_assert(NBUF >= 3 && INST * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(!PF || NBUF >= 4, "fragment prefetch needs chunk c+1 landed while c+2.. are in flight");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
};

// Diagnostic build only (make stamps -> libfcnhip_stamps.so, tools/conv_timeline.py): wave 0 of every workgroup records
// the constant 100 MHz clock and the shader clock at fixed points of conv_body into a buffer of its own.  No stamp exists
// in the product build; no output value depends on one.
#ifdef FCN_CONV_STAMPS
constexpr int kStampWords = 32;
__device__ unsigned long long* g_conv_stamps = nullptr;
__device__ int g_conv_stamps_cap = 0;
#define FCN_STAMP(i)                                                                                          \
    do {                                                                                                      \
        if (g_conv_stamps && (int)blockIdx.x < g_conv_stamps_cap && threadIdx.x == 0) {                       \
            g_conv_stamps[(size_t)blockIdx.x * kStampWords + 2 * (i)] = __builtin_amdgcn_s_memrealtime();     \
            g_conv_stamps[(size_t)blockIdx.x * kStampWords + 2 * (i) + 1] = __builtin_amdgcn_s_memtime();     \
        }                                                                                                     \
    } while (0)
#else
#define FCN_STAMP(i) do { } while (0)
#endif

// T = float: v_mfma_f32_32x32x2_f32, 4 elements per 16-byte segment.  T = _Float16 (inference with f16 activations and
// weights, f32 accumulation, BASELINE configs[4]): v_mfma_f32_32x32x16_f16, 8 elements per segment - the staging, the LDS
// image, the swizzle and the pipeline are byte-for-byte the same (BK counts 4-byte words), a chunk just covers twice the
// k range and one MFMA consumes what four f32 MFMAs do.
typedef const TailP __attribute__((address_space(4))) * tail_kptr;      // a group's tail in the kernel-argument segment

template <typename T, int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int NBUF_, bool PF, bool TAIL = false>
__device__ __forceinline__ void conv_body(const ConvP& p, int tile, float* smem, tail_kptr tp = nullptr) {
    using C = Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF_, PF>;
    static_assert(!TAIL || (C::ROLES && C::BN == 32 && sizeof(T) == 4 && C::NW * 64 >= kTailRows * 8),
                  "tails ride in the float32 split-role shapes with 32-channel tiles (the multiplying waves stage the narrow filters)");
    constexpr int NBUF = C::NBUF;
    constexpr bool ROLES = C::ROLES;
    constexpr int BM = C::BM, SEGS = C::SEGS, RPI = C::RPI, STEP = C::STEP, IA = C::IA, IB = C::IB, INST = C::INST;
    constexpr int D = C::D, KS = C::KS, BUF_FLOATS = C::BUF_FLOATS;
    constexpr bool F16 = sizeof(T) == 2;
    constexpr int EPS = 16 / (int)sizeof(T);          // elements per 16-byte segment
    constexpr int BKE = BK * 4 / (int)sizeof(T);      // k indices one chunk covers
    constexpr int MPS = F16 ? 1 : 4;                  // MFMAs that consume one ds_read_b128 fragment pair
    const T* px = reinterpret_cast<const T*>(p.x);
    const T* pw = reinterpret_cast<const T*>(p.w);

    FCN_STAMP(0);      // kernel entry (after the kernarg loads of the group prologue)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base goes through M0)
    // split roles: waves 0 .. NW-1 multiply, waves NW .. 2NW-1 load; otherwise every wave does both
    const bool is_loader = !ROLES || wid_all >= C::NW, is_mult = !ROLES || wid_all < C::NW;
    const bool loads_b = C::LOADX == 2 && wid_all >= 2 * C::NW;                   // doubled loaders: this wave stages B rows (else A rows)
    const int wid = ROLES && wid_all >= C::NW ? (wid_all - C::NW) % C::NW : wid_all;      // index within the role
    const int wk = wid % WAVES_K;
    const int wn = (wid / WAVES_K) % WAVES_N;
    const int wm = wid / (WAVES_K * WAVES_N);

    // (an integer division is ~30 instructions of straight-line code that every workgroup executes once from a cold
    // instruction cache: the launch prologue uses host-computed multipliers instead)
    const int tile_m = fast_div(tile, p.tiles_n_magic);
    const int tile_n = tile - tile_m * p.tiles_n;
#ifdef FCN_EXP_SAMETILE      // (elimination build: every workgroup stages - and stores - tile (0, 0): all loads hit in L2)
    const int m0 = 0, n0 = 0;
#else
    const int m0 = tile_m * BM;
    const int n0 = tile_n * C::BN;
#endif

    // ---- loader state: this lane stages slot (lane % SEGS) of rows STEP*i + RPI*wid + lane / SEGS ------------
    const int lrow = RPI * wid + lane / SEGS;                  // row of instruction 0
    const int lseg = (lane % SEGS) ^ swz<SEGS>(lrow);          // k-segment this lane fetches (same for all its rows)
    int a_iy0[IA], a_ix0[IA], a_off[IA];   // window origin and its element offset (32-bit: validated on the host)
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int m = m0 + STEP * i + lrow;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int t = fast_div(mm, p.ow_magic);
        const int ox = mm - t * p.OW;
        const int img = fast_div(t, p.oh_magic);
        const int oy = t - img * p.OH;
        a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);   // rows past M never pass the bounds test
        a_ix0[i] = ox * p.stride - p.pad;
        a_off[i] = ((img * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.x_cstride;
    }
    const int taps = p.kh * p.kw;
    const int lds_wave_base = RPI * wid * BK;   // float offset of instruction 0's 1 KiB piece inside a ring slot
    const bool lean = p.lean_chunks > 0;        // uniform: which of the two loaders below feeds this problem
    const int nchunks = lean ? p.lean_chunks : (p.K + BKE - 1) / BKE;

    f32x16 acc[WTM][WTN];
#pragma unroll
    for (int i = 0; i < WTM; ++i)
#pragma unroll
        for (int j = 0; j < WTN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Fragment reads are inline asm: hipcc cannot tell an LDS-DMA in flight from the ds_read of a slot that landed
    // long ago and would drain vmcnt to 0 in front of every compiler-generated LDS read of this array.
    // Addressing: row (lane & 31) of a 32-row tile, k-segment 2*step + (lane >> 5), de-swizzled per lane.
    const int fswz = swz<SEGS>(lane & 31);
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    unsigned fa[KS], fb[KS];           // byte address of this lane's fragment inside ring slot 0, per k-step
#pragma unroll
    for (int st = 0; st < KS; ++st) {
        const int slot4 = ((2 * (wk * KS + st) + (lane >> 5)) ^ fswz) * 4;
        fa[st] = lds0 + 4u * ((wm * WTM * 32 + (lane & 31)) * BK + slot4);
        fb[st] = lds0 + 4u * ((BM + wn * WTN * 32 + (lane & 31)) * BK + slot4);
    }

    // PF: the fragments of the chunk being multiplied and of the next one live in registers, so the LDS read
    // latency of chunk c+1 hides behind the MFMAs of chunk c.  Large tiles (>= 2048 MFMA cycles per barrier) read
    // their fragments just in time instead and keep the registers for accumulators.
    v4f af[PF ? 2 : 1][PF ? KS : 1][WTM], bf[PF ? 2 : 1][PF ? KS : 1][WTN];
    auto ds_read = [](v4f& dst, unsigned addr, auto off) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value));
    };
    auto frags_landed = [&](const int par, const int st0, const int st1) {   // wait for the reads, pinning the registers
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int st = st0; st < st1; ++st) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) asm volatile("" : "+v"(af[par][st][i]));
#pragma unroll
            for (int j = 0; j < WTN; ++j) asm volatile("" : "+v"(bf[par][st][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto read_step = [&](const int par, const int fst, const int st, const unsigned slot_bytes) {
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            if (i == 0) ds_read(af[par][fst][0], fa[st] + slot_bytes, std::integral_constant<int, 0>{});
            if (i == 1) ds_read(af[par][fst][WTM > 1 ? 1 : 0], fa[st] + slot_bytes, std::integral_constant<int, 32 * BK * 4>{});
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            if (j == 0) ds_read(bf[par][fst][0], fb[st] + slot_bytes, std::integral_constant<int, 0>{});
            if (j == 1) ds_read(bf[par][fst][WTN > 1 ? 1 : 0], fb[st] + slot_bytes, std::integral_constant<int, 32 * BK * 4>{});
        }
    };
    // the same with the ring slot as a compile-time constant: slot and tile displacement ride in the instruction's 16-bit
    // offset field, no address arithmetic at all (the f32 MFMA shares the vector ALU with every v_add a wave issues)
    auto read_step_c = [&](const int par, const int fst, const int st, auto slot_c) {
        constexpr int SB = decltype(slot_c)::value * BUF_FLOATS * 4;
#pragma unroll
        for (int i = 0; i < WTM; ++i) {
            if (i == 0) ds_read(af[par][fst][0], fa[st], std::integral_constant<int, SB>{});
            if (i == 1) ds_read(af[par][fst][WTM > 1 ? 1 : 0], fa[st], std::integral_constant<int, SB + 32 * BK * 4>{});
        }
#pragma unroll
        for (int j = 0; j < WTN; ++j) {
            if (j == 0) ds_read(bf[par][fst][0], fb[st], std::integral_constant<int, SB>{});
            if (j == 1) ds_read(bf[par][fst][WTN > 1 ? 1 : 0], fb[st], std::integral_constant<int, SB + 32 * BK * 4>{});
        }
    };
    auto read_frags = [&](const int par, const int buf) {
        const unsigned slot_bytes = (unsigned)buf * (BUF_FLOATS * 4);
#pragma unroll
        for (int st = 0; st < KS; ++st) read_step(par, st, st, slot_bytes);
    };
    // one MFMA group = the WTM x WTN instructions that consume element e of a fragment pair (f32: e = 0..3, 2 k each;
    // f16: the whole 16-byte fragment = 8 halves per lane, 16 k, in one v_mfma_f32_32x32x16_f16)
    auto mfma_group = [&](const int par, const int fst, const int e) {
#ifdef FCN_CONV_NOMFMA      // experiment build (make nomfma): the loader and the barriers alone
        return;
#endif
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                if constexpr (F16)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, af[par][fst][i]),
                                                                       __builtin_bit_cast(v8h, bf[par][fst][j]), acc[i][j], 0, 0, 0);
                else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[par][fst][i][e], bf[par][fst][j][e], acc[i][j], 0, 0, 0);
            }
    };
    auto mfma_step = [&](const int par, const int fst) {
#pragma unroll
        for (int e = 0; e < MPS; ++e) mfma_group(par, fst, e);
    };
    auto mfma_chunk = [&](const int par, const int buf_cur) {
        if (PF) {
#pragma unroll
            for (int st = 0; st < KS; ++st) mfma_step(par, st);
        } else {
            const unsigned slot_bytes = (unsigned)buf_cur * (BUF_FLOATS * 4);
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                read_step(0, 0, st, slot_bytes);
                frags_landed(0, 0, 1);
                mfma_step(0, 0);
            }
        }
    };

    // the bias is needed only by the epilogue (4 consecutive channels per thread): fetch it now so that its latency is
    // not exposed at the end
    float bias_v[4];
    {
        const int nb = n0 + 4 * (tid % (C::BN / 4));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            bias_v[e] = p.bias ? *(const float __attribute__((address_space(1)))*)(p.bias + (nb + e < p.Cout ? nb + e : p.Cout - 1)) : 0.f;
    }

    if constexpr (TAIL) {
        // the narrow problems' filters over this tile's 32 channels -> LDS rows behind the ring ([row][32 floats]): the multiplying waves
        // have nothing to do until the first chunk lands, so the load's latency costs nothing, and the epilogue reads them at LDS speed
        if ((p.flags & FCN_CONV_TAILF) && is_mult && tid < tp->h.rows * 8 && !(tp->h.dbg & 1)) {
            const int o = tid >> 3, seg = tid & 7;
            const float* g = tp->h.gw[0];
#pragma unroll
            for (int j = 1; j < kTailRows / 4; ++j)
                if ((o >> 2) == j) g = tp->h.gw[j];
            const v4f w4 = *(const v4f*)(g + (size_t)(o & 3) * tp->h.K + p.y_coffset + n0 + 4 * seg);
            *(v4f*)(smem + C::LDS_FLOATS + o * 32 + 4 * seg) = w4;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (nothing of this is left in flight when the multiplying waves' counted waits begin)
        }
    }

    FCN_STAMP(7);      // addresses set up
    // ---- prologue: chunks 0 .. D-1 in flight (chunks past K are all-zero, so the counts below never change) ----
    // Tried and dropped in round 2, all measured on one box against this version (gpurun_out/r2/sweep_ab.log, sweep_xr.log;
    // profiles/experiments/r02_*.json): skipping the issue of the chunks past K (uniform branches around the pieces of the
    // interleaved schedule: +15-20 % on the K >= 864 launches); rotating each tile's K loop and an XCD-aware tile order so
    // that tiles sharing an operand hit in L2 instead of missing together (no gain; the bookkeeping of the rotation alone,
    // a dozen VALU instructions per chunk, cost 4-6 %: with one wave per SIMD the loader's VALU work does NOT hide behind the
    // MFMAs - 4a_A runs 8.4 us with and 6.8 us without its MFMAs, sweep_nomfma.log).
    // The ring pipeline, parameterised by the loader that feeds it (its four pieces): prologue issue, one counted wait + one
    // barrier per chunk, MFMAs with the next chunk's fragment reads and the refill's pieces in their shadow.
    auto pipeline = [&](auto& issue_pre, auto& issue_a, auto& issue_b, auto& issue_post) {
    // Issue the LDS-DMA of one chunk (INST x 1 KiB wave-instructions) into ring slot `buf`.  Every lane always loads - lanes
    // outside the image / tile / K get zeros - so the number of outstanding instructions per chunk is a constant the vmcnt
    // waits can count on.
    auto issue_chunk = [&](const int buf) {
        issue_pre(buf);
        if constexpr (C::LOADX == 2) {      // (wave-uniform: a scalar branch)
            if (loads_b) {
#pragma unroll
                for (int i = 0; i < IB; ++i) issue_b(i);
            } else {
#pragma unroll
                for (int i = 0; i < IA; ++i) issue_a(i);
            }
        } else {
#pragma unroll
            for (int i = 0; i < IA; ++i) issue_a(i);
#pragma unroll
            for (int i = 0; i < IB; ++i) issue_b(i);
        }
        issue_post();
    };
    int buf_issue = 0;                 // ring slot of the next chunk to issue
    auto next = [](int b) { return b + 1 == NBUF ? 0 : b + 1; };
    int buf_cur = 0;                   // ring slot of chunk c
    if constexpr (ROLES) {
        // ---- split roles: the same ring, the same one barrier per chunk, but the two halves of the workgroup run different
        // loops.  Every barrier below is executed by BOTH loops (1 + nchunks + 1).
        static_assert(PF, "split roles use the prefetching multiplier loop");
        if (is_loader) {
            // (Tried: issuing only the chunks that exist - a scalar branch around issue_chunk and tail waits that count down -
            // so that the epilogue does not wait for D all-zero transfers.  8 % SLOWER end to end (3335 -> 3057 frames/s, same
            // box): behind a branch the chunk's address arithmetic can no longer be scheduled above the barrier, every chunk
            // is issued a few dozen cycles later, and the loop's pace is set by issue-to-landed latency over D - 1 chunks.)
#pragma unroll 1
            for (int c = 0; c < D; ++c) {
                issue_chunk(buf_issue);
                buf_issue = next(buf_issue);
            }
            wait_vmcnt<INST*(D - 1)>();        // chunk 0 landed (this wave's pieces) ...
            __builtin_amdgcn_s_barrier();      // ... and everybody else's
#pragma unroll 1
            for (int c = 0; c < nchunks; ++c) {
                wait_vmcnt<INST*(D - 2)>();    // chunk c + 1 landed
                __builtin_amdgcn_s_barrier();  // the multipliers are done with chunk c - 1: its slot takes chunk c + D
                asm volatile("" ::: "memory");
#ifndef FCN_EXP_NOLOAD      // (elimination build, make exp EXP=-DFCN_EXP_NOLOAD EXPSRC=conv_fwd: the loop without its staging)
                issue_chunk(buf_issue);
#endif
                buf_issue = next(buf_issue);
            }
            wait_vmcnt<0>();                   // the all-zero chunks issued past K must land before the ring is reused
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        } else {
            // (tried in round 4: s_setprio 3 for the multiplying waves, so that they win the SIMD's issue arbitration - no change on any launch)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            FCN_STAMP(2);      // first chunk usable
            read_frags(0, 0);
            // unrolled so that both the fragment parity and the ring slot are compile-time constants (when the ring fits the
            // 16-bit offset field of ds_read; otherwise by 2 and the slot offset is added at run time)
            constexpr bool IMM = (NBUF + 1) * BUF_FLOATS * 4 < 65536;
            constexpr int U = !IMM ? 2 : (NBUF % 2 == 0 ? NBUF : 2 * NBUF);
            auto iteration = [&](auto u_c) {
                constexpr int u = decltype(u_c)::value;
                __builtin_amdgcn_s_barrier();      // chunk c + 1 is in LDS
                asm volatile("" ::: "memory");
                frags_landed(u & 1, 0, KS);
                const unsigned nslot = (unsigned)next(buf_cur) * (BUF_FLOATS * 4);
                constexpr int NG = KS * MPS;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    mfma_group(u & 1, g / MPS, g % MPS);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g < KS) {      // the fragments of chunk c + 1, in the MFMAs' shadow
                        if constexpr (IMM) read_step_c((u + 1) & 1, g, g, std::integral_constant<int, (u + 1) % NBUF>{});
                        else read_step((u + 1) & 1, g, g, nslot);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                buf_cur = next(buf_cur);
            };
            for (int c0 = 0; c0 < nchunks; c0 += U) {      // (c0 is a multiple of the ring depth: chunk c0 + u sits in slot u % NBUF)
                if (c0 + 0 < nchunks) iteration(std::integral_constant<int, 0>{});
                if (c0 + 1 < nchunks) iteration(std::integral_constant<int, 1>{});
                if constexpr (U > 2) {
                    if (c0 + 2 < nchunks) iteration(std::integral_constant<int, 2>{});
                    if (c0 + 3 < nchunks) iteration(std::integral_constant<int, 3>{});
                }
                if constexpr (U > 4) {
                    if (c0 + 4 < nchunks) iteration(std::integral_constant<int, 4>{});
                    if (c0 + 5 < nchunks) iteration(std::integral_constant<int, 5>{});
                }
                if constexpr (U > 6) {
                    if (c0 + 6 < nchunks) iteration(std::integral_constant<int, 6>{});
                    if (c0 + 7 < nchunks) iteration(std::integral_constant<int, 7>{});
                    if (c0 + 8 < nchunks) iteration(std::integral_constant<int, 8>{});
                    if (c0 + 9 < nchunks) iteration(std::integral_constant<int, 9>{});
                }
                static_assert(U <= 10, "ring depths up to 5 (odd) or 10 (even)");
            }
            FCN_STAMP(3);      // main loop done
            frags_landed(0, 0, KS);      // (the reads of the chunk that does not exist: see the comment in the other branch)
            frags_landed(1, 0, KS);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
#pragma unroll 1
    for (int c = 0; c < D; ++c) {      // rolled: straight-line code costs instruction fetches, and this runs once
        issue_chunk(buf_issue);
        buf_issue = next(buf_issue);
    }
    FCN_STAMP(1);      // prologue issued
    if (PF) {
        wait_vmcnt<INST*(D - 1)>();   // chunk 0 landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();  // ... and everybody else's
        asm volatile("" ::: "memory");
        read_frags(0, 0);
    }

    // ---- main loop: one counted wait + one barrier per chunk -------------------------------------------------
    // Iteration c: [PF] chunk c+1 (else chunk c) has landed -> barrier -> refill the slot chunk c-1 vacated with
    // chunk c+D -> [PF] read the fragments of chunk c+1 -> MFMAs of chunk c.
    for (int c0 = 0; c0 < nchunks; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {      // unrolled by 2: the fragment parity is a compile-time constant
            if (c0 + u < nchunks) {
                wait_vmcnt<INST*(D - 1 - (PF ? 1 : 0))>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#ifdef FCN_CONV_STAMPS
                if (c0 + u == 0) FCN_STAMP(2);      // first chunk usable
#endif
                if (PF) {
                    // The wave has one instruction stream: address arithmetic, DMA issue and fragment reads placed in
                    // front of the MFMAs would leave the matrix core idle meanwhile (with one wave per SIMD nobody
                    // else feeds it).  The fragments of chunk c were read one iteration ago, so the MFMAs start right
                    // after the barrier and everything else is issued in their shadow, one piece per MFMA group:
                    // first the fragment reads of chunk c+1, then the DMA of chunk c+D.
                    frags_landed(u & 1, 0, KS);
                    const unsigned nslot = (unsigned)next(buf_cur) * (BUF_FLOATS * 4);
                    constexpr int NG = KS * MPS;                // MFMA groups of WTM*WTN instructions
                    constexpr int NP = KS + 1 + IA + IB + 1;    // pieces: reads, issue_pre, A loads, B loads, issue_post
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        mfma_group(u & 1, g / MPS, g % MPS);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int q = g * NP / NG; q < (g + 1) * NP / NG; ++q) {
                            if (q < KS) read_step((u + 1) & 1, q, q, nslot);
                            else if (q == KS) issue_pre(buf_issue);
                            else if (q < KS + 1 + IA) issue_a(q - KS - 1);
                            else if (q < KS + 1 + IA + IB) issue_b(q - KS - 1 - IA);
                            else issue_post();
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    buf_issue = next(buf_issue);
                } else {
                    issue_chunk(buf_issue);
                    buf_issue = next(buf_issue);
                    mfma_chunk(u & 1, buf_cur);
                }
                buf_cur = next(buf_cur);
#ifdef FCN_CONV_STAMPS
                if (c0 + u < 4) FCN_STAMP(8 + c0 + u);      // iterations 0..3 done (each stamp costs the loop ~0.3 us: read the first three rows only)
#endif
            }
        }
    }
    FCN_STAMP(3);      // main loop done
    wait_vmcnt<0>();                   // the all-zero chunks issued past K must land before the ring is reused
    // The last iteration prefetched the fragments of a chunk that does not exist, with inline-asm ds_reads the compiler
    // cannot see: nothing waits for them, their destination registers are dead, and the register allocator hands those
    // registers to the epilogue (the accumulator copies).  A read that returns late - LDS contended by a co-resident
    // kernel of another stream - then overwrites four consecutive accumulator copies AFTER v_accvgpr_read filled them:
    // a 32-column x 4-register block of stale partial sums (found in round 2 by diffing gradient blobs of repeated two-stream steps).  Retire them -
    // and keep the fragment registers ALIVE across the wait (frags_landed pins them and ends with a scheduling barrier):
    // a bare `s_waitcnt` asm orders memory operations only, and the scheduler did hoist the epilogue's v_accvgpr_read
    // into those registers above it (round 2: the two-stream determinism tests caught exactly that).
    if (PF) {
        frags_landed(0, 0, KS);
        frags_landed(1, 0, KS);
    } else {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);

    }

    };      // pipeline

    if (lean) {
        // ---- scalar-addressed loader -----------------------------------------------------------------------------------
        // v_mfma_f32_32x32x2_f32 runs on the vector ALUs (64 FLOP/clk/SIMD IS the f32 vector rate): the loader's address
        // arithmetic does not hide behind a wave's own MFMAs, it ADDS to them - with 32x32 K-split tiles (4 MFMAs = 256 cycles
        // per wave and chunk) the ~45 VALU instructions of the per-lane loader below cost as much again, which is why a launch
        // without its MFMAs is barely faster than with them (sweep_nomfma.log) and why two waves per SIMD do not help.  Here a
        // chunk never straddles a filter tap (1x1 filters, or Cin padded to whole chunks per tap: the pad lanes load zeros), so
        // the chunk's tap, channel origin and both operand offsets are WAVE-UNIFORM: they live in SGPRs and advance on the
        // scalar unit.  Per lane there remain one select per piece (+ the halo test of padded convolutions).  Lanes outside
        // the image / tile / K carry an out-of-range buffer offset: `buffer_load ... lds` writes zeros for them
        // (tools/probes/bufload_lds_probe.hip), so no zero page and no 64-bit address select either.
        // (Tried on top of this in round 2: treating a whole filter ROW of kw taps as one contiguous run when the input's pixels
        // are packed - conv1 as 7 x 28 floats, a 5x5 on 16 channels as 5 x 80 - so that small-Cin layers take this loader too.
        // The two extra VALU instructions per chunk it needs cost every launch 3-6 % and conv1 gained nothing:
        // gpurun_out/r2/sweep_merged_ab.log, same box, alternating builds.  Also tried: the chunk offsets in the instruction's
        // scalar-offset field instead of a per-lane add (neutral, sweep_lean3_ab.log); plus a per-row bit mask of the taps
        // inside the image and a scalar-selected tail predicate - 7 instead of 11 VALU instructions per chunk in the steady
        // state, but the mask's set-up loops and the ballot lengthen the prologue: 4a_A 6.2 -> 7.1 us, sweep_lean2_ab.log.)
        constexpr int ESZ = (int)sizeof(T);
        constexpr int OOB = (int)0x80000000u;      // >= num_records of every buffer (validate(): operands stay below 2 GiB)
        const int lane_c = lseg * EPS;             // channel of this lane's segment inside a chunk
        const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<T*>(px), 0, (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * ESZ), 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(pw), 0, p.Cout * p.K * ESZ, 0x00020000);
        int a_vo[IA], b_vo[IB];
#pragma unroll
        for (int i = 0; i < IA; ++i) a_vo[i] = (a_off[i] + lane_c) * ESZ;
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int n = n0 + STEP * i + lrow;
            b_vo[i] = n < p.Cout ? (n * p.K + lane_c) * ESZ : OOB;
        }
        float* dst = smem;
        // which of the two scalar-addressed loaders: plan_tiles_cfg marks the problems that take the scalar-light one (kw_magic, which only the
        // per-lane loader reads, is -1 then) - 1x1 filters and filters whose taps span several chunks; split-role shapes only
        bool light = false;
        if constexpr (ROLES) light = p.kw_magic == -1;
        if (light) {
          if constexpr (ROLES) {
            // ---- round 4: the same loader with its bookkeeping off the critical path.  Elimination builds of the split-role 32 x 32
            // shape (profiles/experiments/r04_elimination_cfg23.txt): 0.125 us per chunk with the staging compiled out (the four MFMAs
            // are 0.117), 0.159 us with the MFMAs compiled out, 0.179 us as built - and the same with every workgroup staging the SAME
            // tile (all loads L2 hits), with a deeper ring, with two workgroups per CU: the loading waves' own instruction stream sets
            // the pace.  Per chunk it was ~45 scalar and 9 vector instructions around two LDS-DMA pieces - the generic (tap, channel)
            // state machine - and the scalar unit is shared by all waves of a CU.  Here a filter tap is entered once: its halo tests and
            // per-lane offsets are computed THEN (vector ALU, once per tap), the chunks of the tap differ only in the SCALAR offset of
            // the buffer instruction, and the partial last chunk of a tap (Cin not a multiple of the chunk) swaps in pre-masked offsets.
            // Per chunk: ~16 scalar instructions, no vector instruction, two LDS-DMA pieces.
            // A vector instruction of a loading wave is the expensive kind here: v_mfma_f32_32x32x2_f32 occupies the vector ALU the
            // loading wave shares with a multiplying wave, so each one waits for a gap between MFMAs (a tap change of ~10 vector
            // instructions measured 0.26 us against 0.14 us for a plain chunk: tools/conv_timeline.py on a 3x3 with one chunk per
            // tap).  So the halo tests of ALL taps are made once, in the prologue, into one bit per tap and lane (taps <= 32:
            // plan_tiles_cfg), the buffer's base is moved back by the largest negative window offset so that the tap offset can ride
            // in the scalar offset too, and a tap change is and + compare + select per staged row.
            const int cpt = (p.Cin + BKE - 1) / BKE;                 // chunks per tap
            const int rem = p.Cin - (cpt - 1) * BKE;                 // channels the last chunk of a tap covers
            const bool partial = rem != BKE;
            const int bias = (p.pad * p.W + p.pad) * p.x_cstride * ESZ;      // bytes: the most negative (window origin - image origin)
            const __amdgpu_buffer_rsrc_t rxb = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<T*>(px) - bias / ESZ, 0, (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * ESZ) + bias, 0x00020000);
            unsigned okmask[IA];
            unsigned row_bits = 0;                                   // (scalar) bit r * kw for every filter row r
            for (int r = 0; r < p.kh; ++r) row_bits |= 1u << (r * p.kw);
            int a_vb[IA], a_cur[IA], b_last[IB], b_cur[IB];
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                a_vb[i] = a_vo[i] + bias;                            // >= 0 for every lane whose tap is inside the image
                if (taps == 1) {
                    okmask[i] = (unsigned)((int)((unsigned)a_iy0[i] < (unsigned)p.H) & (int)((unsigned)a_ix0[i] < (unsigned)p.W));
                } else {
                    // the filter rows inside the image are a RANGE [lo, hi) of r (likewise the columns), so both masks are bit fields in
                    // closed form - straight-line code, no loop over the taps: bit r * kw of `rows`, bit q of `cols`, and their product
                    // has bit r * kw + q (no carries: cols < 2^kw, the row bits are kw apart; taps <= 31)
                    const int lo_r = min(max(-a_iy0[i], 0), p.kh), hi_r = min(max(p.H - a_iy0[i], 0), p.kh);
                    const int lo_q = min(max(-a_ix0[i], 0), p.kw), hi_q = min(max(p.W - a_ix0[i], 0), p.kw);
                    const unsigned rows = row_bits & (((1u << (max(hi_r - lo_r, 0) * p.kw)) - 1u) << (lo_r * p.kw));
                    const unsigned cols = ((1u << max(hi_q - lo_q, 0)) - 1u) << lo_q;
                    okmask[i] = rows * cols;
                }
            }
            const bool lane_in_rem = lane_c < rem;
            const bool last_always = partial && cpt == 1;            // one chunk per tap and that chunk is partial: every chunk is a "last" one
            int b_tap[IB];                                           // what a tap's first chunk uses for the B rows
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                b_last[i] = lane_in_rem ? b_vo[i] : OOB;
                b_tap[i] = last_always ? b_last[i] : b_vo[i];
            }
            if (last_always) {
#pragma unroll
                for (int i = 0; i < IA; ++i) okmask[i] = lane_in_rem ? okmask[i] : 0u;
            }
            int s_j = 0, s_kq = -1, s_aoff = 0, s_boff = 0, s_tapb = -p.Cin * ESZ, s_left = taps;      // s_left: taps not entered yet
            unsigned s_nbit = 1u;                                    // bit of the NEXT tap to be entered, 0 when there is none
            const int s_dq = p.x_cstride * ESZ, s_dr = (p.W - p.kw + 1) * p.x_cstride * ESZ;      // next tap of the row / first tap of the next row
            int tap_off = -s_dq;
            int a_next[IA];                                          // the NEXT tap's per-lane offsets, computed one tap ahead
#pragma unroll
            for (int i = 0; i < IA; ++i) a_next[i] = (okmask[i] & s_nbit) != 0 ? a_vb[i] : OOB;
            // Straight-line (selects, no branches).  The and / compare / select of a tap are a dependent chain of vector instructions that
            // each wait for a gap between the MFMAs of the wave sharing the SIMD; computed at the tap change itself they delayed the
            // next chunk's loads (one chunk per tap: 0.24 us per chunk against 0.14).  So a tap's offsets are computed one tap AHEAD -
            // the chain has a whole tap to finish - and the change itself is a register move.  Past K the tap bit is 0, which
            // turns every A lane into an out-of-range offset by itself.
            auto tap_advance = [&]() {
                const bool live = s_left > 0;
                s_left -= live ? 1 : 0;
                ++s_kq;
                const bool wq = s_kq == p.kw;
                s_kq = wq ? 0 : s_kq;
                tap_off += wq ? s_dr : s_dq;                         // byte offset of this tap from the window origin
                s_aoff = tap_off;
                s_tapb += p.Cin * ESZ;
                s_boff = s_tapb;
                s_j = live ? cpt : 0x7fffffff;
                s_nbit = s_left > 0 ? s_nbit << 1 : 0u;
#pragma unroll
                for (int i = 0; i < IA; ++i) {
                    a_cur[i] = a_next[i];
                    a_next[i] = (okmask[i] & s_nbit) != 0 ? a_vb[i] : OOB;
                }
#pragma unroll
                for (int i = 0; i < IB; ++i) b_cur[i] = live ? b_tap[i] : OOB;
            };
            tap_advance();
            auto lpre = [&](const int buf) { dst = smem + buf * BUF_FLOATS + lds_wave_base; };
            // (the host pass of hipcc type-checks this builtin too and insists on a constant scalar offset there: device code only)
            auto la = [&](const int i) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rxb, (lds_ptr)(dst + STEP * i * BK), 16, a_cur[i], s_aoff, 0, 0);
#endif
            };
            auto lb = [&](const int i) {
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + (BM + STEP * i) * BK), 16, b_cur[i], s_boff, 0, 0);
#endif
            };
            auto lpost = [&]() {
                s_aoff += BKE * ESZ;
                s_boff += BKE * ESZ;
                --s_j;
                if (__builtin_expect(s_j <= 1, 0)) {      // (one rarely taken branch on the common path)
                    if (s_j == 0) {
                        tap_advance();
                    } else if (partial) {                   // the tap's last chunk covers `rem` channels only
#pragma unroll
                        for (int i = 0; i < IA; ++i) a_cur[i] = lane_in_rem ? a_cur[i] : OOB;
#pragma unroll
                        for (int i = 0; i < IB; ++i) b_cur[i] = b_last[i];
                    }
                }
            };
            pipeline(lpre, la, lb, lpost);
          }
        } else {
        int s_kr = 0, s_kq = 0, s_kc = 0, s_koff = 0, s_kb = 0, s_left = nchunks;      // wave-uniform K position of the next chunk
        bool cvalid = false;
        auto lpre = [&](const int buf) {
            const int thr = s_left > 0 ? p.Cin - s_kc : 0;      // channels of this tap the chunk still covers (0: past the end)
            cvalid = lane_c < thr;
            dst = smem + buf * BUF_FLOATS + lds_wave_base;
        };
        auto la = [&](const int i) {
            // (rows past M carry a hugely negative origin and fail the first test; for unpadded convolutions both tests hold
            // for every real row - selecting a cheaper test per problem cost more select instructions than the test itself)
            const bool ok = (int)cvalid & (int)((unsigned)(a_iy0[i] + s_kr) < (unsigned)p.H) & (int)((unsigned)(a_ix0[i] + s_kq) < (unsigned)p.W);
            int vo = ok ? a_vo[i] + s_koff : OOB;
            asm volatile("" : "+v"(vo));      // one select, one DMA
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(dst + STEP * i * BK), 16, vo, 0, 0, 0);
        };
        auto lb = [&](const int i) {
            int vo = cvalid ? b_vo[i] + s_kb : OOB;      // (OOB + a row offset stays out of range: both below 2^31)
            asm volatile("" : "+v"(vo));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + (BM + STEP * i) * BK), 16, vo, 0, 0, 0);
        };
        auto lpost = [&]() {      // scalar unit only
            --s_left;
            const bool nt = s_kc + BKE >= p.Cin;                  // the next chunk starts the next tap
            s_kb += (nt ? p.Cin - s_kc : BKE) * ESZ;              // weights are [tap][Cin]: the tap's end, or one chunk on
            const int kq1 = s_kq + (nt ? 1 : 0);
            const bool wq = kq1 == p.kw;
            s_kq = wq ? 0 : kq1;
            s_kr += wq ? 1 : 0;
            s_kc = nt ? 0 : s_kc + BKE;
            s_koff = nt ? (s_kr * p.W + s_kq) * p.x_cstride * ESZ : s_koff + BKE * ESZ;
        };
        pipeline(lpre, la, lb, lpost);
        }
    } else {
        // ---- per-lane loader: any geometry (chunks may straddle taps: Cin = 3 + 1 pad of conv1, 16, ...) -----------------
        // k position of this lane's segment, kept as (tap, channel) and advanced by BK per chunk without branches
        int kc = lseg * EPS;
        int kt = (int)(((unsigned)kc * p.cin_magic24) >> 24);      // kc / Cin (kc < 256)
        kc -= kt * p.Cin;
        const int bk_taps = (int)(((unsigned)BKE * p.cin_magic24) >> 24), bk_rem = BKE - bk_taps * p.Cin;      // BKE / Cin
        int b_off[IB];   // element offset of this lane's segment in weight row n; negative = row past Cout
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            const int n = n0 + STEP * i + lrow;
            b_off[i] = n < p.Cout ? n * p.K + lseg * EPS : -1;
        }
        int kb = lseg * EPS;   // this segment's k index (weights are zero past K)
        // the zero page pointer is laundered into VGPRs so that "in bounds ? source : zero page" stays a plain select
        // (one LDS-DMA instruction per row group) instead of two exec-masked instructions
        unsigned long long zp_bits = reinterpret_cast<unsigned long long>(p.zero_page);
        asm volatile("" : "+v"(zp_bits));
        const float* zero_page = reinterpret_cast<const float*>(zp_bits);
        int is_kr = 0, is_kq = 0, is_koff = 0;   // state shared by the pieces of one chunk's issue
        bool is_kok = false, is_kbok = false;
        float* is_dst = smem;
        auto issue_pre = [&](const int buf) {
            is_kr = (kt * p.kw_magic) >> 16;            // kt / kw (magic = ceil(65536 / kw), exact for kt < 8192)
            is_kq = kt - is_kr * p.kw;
            is_koff = (is_kr * p.W + is_kq) * p.x_cstride + kc;
            is_kok = kt < taps;
            is_kbok = kb < p.K;
            is_dst = smem + buf * BUF_FLOATS + lds_wave_base;
        };
        auto issue_a = [&](const int i) {
            // bitwise & on purpose: && becomes a divergent branch around the address arithmetic
            const bool ok = (int)is_kok & (int)((unsigned)(a_iy0[i] + is_kr) < (unsigned)p.H) & (int)((unsigned)(a_ix0[i] + is_kq) < (unsigned)p.W);
            unsigned long long src = ok ? reinterpret_cast<unsigned long long>(px + (a_off[i] + is_koff)) : reinterpret_cast<unsigned long long>(zero_page);
            asm volatile("" : "+v"(src));    // one select, one DMA (keeps hipcc from forking the load into two exec-masked copies)
            __builtin_amdgcn_global_load_lds((gvoid_cptr)src, (lds_ptr)(is_dst + STEP * i * BK), 16, 0, 0);
        };
        auto issue_b = [&](const int i) {
            unsigned long long src = ((int)is_kbok & (int)(b_off[i] >= 0)) ? reinterpret_cast<unsigned long long>(pw + b_off[i])
                                                                           : reinterpret_cast<unsigned long long>(zero_page);
            asm volatile("" : "+v"(src));
            __builtin_amdgcn_global_load_lds((gvoid_cptr)src, (lds_ptr)(is_dst + (BM + STEP * i) * BK), 16, 0, 0);
            b_off[i] += b_off[i] >= 0 ? BKE : 0;
        };
        auto issue_post = [&]() {
            kb += BKE;
            // advance by one chunk: BKE = bk_taps * Cin + bk_rem
            kc += bk_rem;
            kt += bk_taps;
            const bool wrap = kc >= p.Cin;
            kc -= wrap ? p.Cin : 0;
            kt += wrap ? 1 : 0;
        };
        pipeline(issue_pre, issue_a, issue_b, issue_post);
    }

    // ---- epilogue: every wave parks its accumulators in LDS, then ALL threads of the workgroup reduce the K-split
    // partials (fixed order wk = 0, 1, ..), add the bias, apply ReLU / accumulate / mask / sigmoid and store 4 consecutive
    // channels per lane (one 16-byte store).  The former epilogue - wave wk = 0 alone, one 4-byte store per accumulator
    // register behind a ladder of per-element branches - was 1700 instructions of straight-line code executed once:
    // 2.0 us of a 9 us launch at M = 784 (tools/conv_timeline.py), most of it instruction fetch.
    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    if (is_mult) {
        float* slab = smem + (size_t)(((wm * WAVES_N + wn) * WAVES_K + wk) * WTM * WTN) * 1024 + (lane >> 5) * 128 + (lane & 31);
#pragma unroll
        for (int i = 0; i < WTM; ++i)
#pragma unroll
            for (int j = 0; j < WTN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) slab[(i * WTN + j) * 1024 + ((r & 3) + 8 * (r >> 2)) * 32] = acc[i][j][r];
    }
    __syncthreads();
    FCN_STAMP(4);      // accumulators parked
    constexpr int C4 = C::BN / 4;                   // float4 columns of the output tile
    static_assert(C::NT % C4 == 0, "every thread keeps one float4 column");
    const int c4 = tid % C4;                        // the same for all items of this thread
    const int n = n0 + 4 * c4;
    if constexpr (TAIL) {
        if (p.flags & FCN_CONV_TAILF) {
            // A tail problem (host-checked: float32, bias + ReLU only, Cout and the output slice in whole 32-channel tiles, 16-byte aligned
            // output).  C4 == 8: thread (it = tid / 8, c4) holds channels n .. n+3 of pixel m0 + it.
            const int rows = tp->h.rows, tm = tp->h.M;
            const int it = tid >> 3, m = m0 + it;
            if (it < BM) {      // (wave-uniform: whole waves)
                const bool act = m < p.M;
                const int wm_t = it / (32 * WTM), i_t = (it / 32) % WTM;
                const int wn_t = (4 * c4) / (32 * WTN), jt = ((4 * c4) / 32) % WTN, lc = (4 * c4) & 31;
                const float* src = smem + (size_t)(wn_t * WAVES_K * WTM * WTN + jt) * 1024 + lc +
                                   (size_t)(wm_t * WAVES_N * WAVES_K * WTM * WTN + i_t * WTN) * 1024 + (it & 31) * 32;
                v4f v = *(const v4f*)src;
#pragma unroll
                for (int s = 1; s < WAVES_K; ++s) v += *(const v4f*)(src + (size_t)s * WTM * WTN * 1024);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += bias_v[e];
                if (p.flags & FCN_CONV_RELU)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                if (act) *(v4f*)(p.y + (size_t)m * p.y_cstride + p.y_coffset + n) = v;
                // partial sums of the narrow problems over these 4 channels, then over the 8 lanes that hold the pixel's 32 channels
                // (a butterfly: every lane runs the same tree, the sum does not depend on the lane that stores it)
                float* part = tp->h.scratch + ((size_t)((p.y_coffset + n) >> 5) * tm + (act ? m : 0)) * rows;
                const float* tw = smem + C::LDS_FLOATS + 4 * c4;
                // A ROLLED loop over the groups of four rows: straight-line code that runs once is paid in instruction fetch (~1.2 ns per
                // instruction from a cold instruction cache, the finding behind this epilogue's own shape) - unrolled six times this block
                // cost the launch 2 us.  The exchanges are DPP moves on the vector ALU - lane ^ 1, lane ^ 2, then the mirror image inside
                // the 8 lanes (every lane of a quad already holds the quad's sum); __shfl_xor would be an LDS round trip each.
                auto dpp_add = [](float x, auto ctrl) {
                    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, false));
                };
#pragma unroll 1
                for (int j = 0; 4 * j < ((tp->h.dbg & 2) ? 0 : rows); ++j) {
                    v4f t;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const v4f w4 = *(const v4f*)(tw + (4 * j + r) * 32);
                        t[r] = (v[0] * w4[0] + v[1] * w4[1]) + (v[2] * w4[2] + v[3] * w4[3]);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = t[r];
                        x = dpp_add(x, std::integral_constant<int, 0xB1>{});       // quad_perm [1, 0, 3, 2]
                        x = dpp_add(x, std::integral_constant<int, 0x4E>{});       // quad_perm [2, 3, 0, 1]
                        x = dpp_add(x, std::integral_constant<int, 0x141>{});      // row_half_mirror: lane i <-> 7 - i of every 8
                        t[r] = x;
                    }
                    // (device-coherent store, sc1: written through the XCD's L2 - the tile that adds the slots up runs on any XCD, and a
                    //  release fence instead would be an L2 write-back per workgroup: 500 of them made this launch 55 us instead of 18)
                    if (act && c4 == j && !(tp->h.dbg & 4)) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(part + 4 * j), "v"(t) : "memory");
                }
            }
            const int need = tp->h.need;
            if (need == 0) return;      // (uniform) an earlier launch of the same blob: partial sums only
            // arrival: the workgroup that completes a pixel block finishes the narrow problems for it.  Nobody waits for anybody.
            int* s_last = reinterpret_cast<int*>(smem + C::LDS_FLOATS + kTailRows * 32);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                // every partial sum of this workgroup has been acknowledged by memory (write-through stores, vmcnt(0) in front of the barrier)
                // before the arrival (a device-scope atomic at the memory side); the sums are read back with device-coherent loads: no fence
                const unsigned old = atomicAdd(tp->h.arrive + tile_m, 1u);
                *s_last = old + 1u == (unsigned)need ? 1 : 0;
            }
            __syncthreads();
            if (!*s_last) return;
            // The last tile of the block: sum[pixel][row] = (slots 0 .. H-1, in order) + (slots H .. 2H-1, in order), H = half the slots - one
            // thread per (pixel, four rows, half), all of a thread's loads in flight at once (a device-coherent load is a trip to memory:
            // a loop over the slots in batches was four or five trips long), the two halves meet in LDS.  A fixed order: the same sum in every run.
            const int groups = rows >> 2, nslots = tp->h.K >> 5, items = BM * groups;
            constexpr int H = 16;
            const size_t slot_stride = (size_t)tm * rows;
            float* halves = smem;      // (the parked accumulators have been read: 2 x items x 4 floats fit in their place)
            static_assert(2 * BM * (kTailRows / 4) * 4 <= C::EPI_FLOATS, "the half sums borrow the accumulator image");
            for (int w = tid; w < 2 * items; w += C::NT) {
                const int hf = w >= items ? 1 : 0, wi = w - hf * items;
                const int px = wi / groups, g = wi - px * groups, mm = m0 + px;
                const float* src = tp->h.scratch + (size_t)(mm < p.M ? mm : 0) * rows + 4 * g + (size_t)(hf * H) * slot_stride;
                v4f q[H];
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    q[u] = v4f{0.f, 0.f, 0.f, 0.f};
                    if (hf * H + u < nslots)      // device-coherent load: never a stale line of this XCD's L2
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+v"(q[u]) : "v"(src + (size_t)u * slot_stride) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int u = 0; u < H; ++u) asm volatile("" : "+v"(q[u]));
                __builtin_amdgcn_sched_barrier(0);
                v4f sum = q[0];
#pragma unroll
                for (int u = 1; u < H; ++u) sum += q[u];
                *(v4f*)(halves + (size_t)w * 4) = sum;
            }
            __syncthreads();
            TailSlice sls[kTailMaxSlices];      // (all slices in one trip to the kernel arguments)
#pragma unroll
            for (int si = 0; si < kTailMaxSlices; ++si) {
                sls[si].bias = tp->s[si].bias; sls[si].y = tp->s[si].y; sls[si].y2 = tp->s[si].y2;
                sls[si].o0 = tp->s[si].o0; sls[si].nout = tp->s[si].nout; sls[si].flags = tp->s[si].flags;
                sls[si].y_cstride = tp->s[si].y_cstride; sls[si].y_coffset = tp->s[si].y_coffset;
                sls[si].y2_cstride = tp->s[si].y2_cstride; sls[si].y2_coffset = tp->s[si].y2_coffset;
            }
            for (int w = tid; w < items; w += C::NT) {
                const int px = w / groups, g = w - px * groups, mm = m0 + px;
                if (mm >= p.M) continue;
                const v4f sum = *(const v4f*)(halves + (size_t)w * 4) + *(const v4f*)(halves + (size_t)(items + w) * 4);
                const int o = 4 * g;
#pragma unroll
                for (int si = 0; si < kTailMaxSlices; ++si) {
                    const TailSlice& sl = sls[si];
                    if (si >= tp->h.nslices || o < sl.o0 || o >= sl.o0 + sl.nout) continue;
                    v4f r = sum;
                    if (sl.bias) r += *(const v4f*)(sl.bias + (o - sl.o0));
                    if (sl.flags & FCN_CONV_RELU)
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
                    *(v4f*)(sl.y + (size_t)mm * sl.y_cstride + sl.y_coffset + (o - sl.o0)) = r;
                    if ((sl.flags & FCN_CONV_SIGMOID2) && sl.y2) {
                        v4f sg;
#pragma unroll
                        for (int e = 0; e < 4; ++e) sg[e] = 1.f / (1.f + expf(-r[e]));
                        *(v4f*)(sl.y2 + (size_t)mm * sl.y2_cstride + sl.y2_coffset + (o - sl.o0)) = sg;
                    }
                }
            }
            if (tid == 0) tp->h.arrive[tile_m] = 0u;      // (every tile of the block has arrived: the word is ready for the next launch)
            return;
        }
    }
    if (n >= p.Cout) return;
    const bool do_relu = (p.flags & FCN_CONV_RELU) != 0;
    const bool do_sig2 = (p.flags & FCN_CONV_SIGMOID2) != 0 && p.y2 != nullptr;
    const bool do_mask = (p.flags & FCN_CONV_MASK) != 0 && p.y2 != nullptr;      // ReLU backward of the layer below (y2 = its activation)
    const bool do_accum = (p.flags & FCN_CONV_ACCUM) != 0;
    const bool out_f32 = (p.flags & FCN_CONV_OUT_F32) != 0;      // f16 inputs, f32 output blob (the detection heads)
    const bool out_f16 = (p.flags & FCN_CONV_OUT_F16) != 0;      // f32 inputs, f16 output blob (the first layer of an f16 net)
    const bool half_out = (F16 && !out_f32) || (!F16 && out_f16);
    // 16-byte (8-byte for halves) accesses need the whole group inside the output slice and aligned
    const bool vec = n + 3 < p.Cout && ((p.y_cstride | p.y_coffset) & 3) == 0 && ((p.y2_cstride | p.y2_coffset) & 3) == 0 &&
                     ((unsigned)(size_t)p.y & 15) == 0 && ((unsigned)(size_t)p.y2 & 15) == 0;
    const int wn_t = (4 * c4) / (32 * WTN), jt = ((4 * c4) / 32) % WTN, lc = (4 * c4) & 31;
    const float* lds_col = smem + (size_t)(wn_t * WAVES_K * WTM * WTN + jt) * 1024 + lc;
#pragma unroll 1
    for (int it = tid / C4; it < BM; it += C::NT / C4) {
        const int m = m0 + it;
        if (m >= p.M) break;
        const int wm_t = it / (32 * WTM), i_t = (it / 32) % WTM;
        const float* src = lds_col + (size_t)(wm_t * WAVES_N * WAVES_K * WTM * WTN + i_t * WTN) * 1024 + (it & 31) * 32;
        v4f v = *(const v4f*)src;
#pragma unroll
        for (int s = 1; s < WAVES_K; ++s) v += *(const v4f*)(src + (size_t)s * WTM * WTN * 1024);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] += bias_v[e];
        const size_t o = (size_t)m * p.y_cstride + p.y_coffset + n;
        const size_t o2 = (size_t)m * p.y2_cstride + p.y2_coffset + n;
        if (half_out) {      // f16 activations: rounded once, after bias and ReLU
            typedef f16_t v4h __attribute__((ext_vector_type(4)));
            f16_t* dst = reinterpret_cast<f16_t*>(p.y) + o;
            if (vec) {
                if (do_accum) { const v4h old = *(const v4h*)dst; for (int e = 0; e < 4; ++e) v[e] += (float)old[e]; }
                if (do_relu) for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                v4h h;
                for (int e = 0; e < 4; ++e) h[e] = (f16_t)v[e];
                *(v4h*)dst = h;
            } else {
                for (int e = 0; e < 4; ++e)
                    if (n + e < p.Cout) {
                        float t = v[e];
                        if (do_accum) t += (float)dst[e];
                        if (do_relu) t = fmaxf(t, 0.f);
                        dst[e] = (f16_t)t;
                    }
            }
        } else if (vec) {
            float* dst = p.y + o;
            if (do_accum) v += *(const v4f*)dst;
            if (do_relu) for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (do_mask) { const v4f a = *(const v4f*)(p.y2 + o2); for (int e = 0; e < 4; ++e) v[e] = a[e] > 0.f ? v[e] : 0.f; }
            *(v4f*)dst = v;
            if (do_sig2) {
                v4f sg;
                for (int e = 0; e < 4; ++e) sg[e] = 1.f / (1.f + expf(-v[e]));
                *(v4f*)(p.y2 + o2) = sg;
            }
        } else {
            for (int e = 0; e < 4; ++e)
                if (n + e < p.Cout) {
                    float t = v[e];
                    if (do_accum) t += p.y[o + e];
                    if (do_relu) t = fmaxf(t, 0.f);
                    if (do_mask) t = p.y2[o2 + e] > 0.f ? t : 0.f;
                    p.y[o + e] = t;
                    if (do_sig2) p.y2[o2 + e] = 1.f / (1.f + expf(-t));
                }
        }
    }
#ifdef FCN_CONV_STAMPS
    FCN_STAMP(5);      // stores issued
    wait_vmcnt<0>();
    FCN_STAMP(6);      // stores acknowledged
    if (g_conv_stamps && (int)blockIdx.x < g_conv_stamps_cap && threadIdx.x == 0) {
        unsigned xcc = 0, hwid = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_conv_stamps[(size_t)blockIdx.x * kStampWords + 30] = xcc;
        g_conv_stamps[(size_t)blockIdx.x * kStampWords + 31] = hwid;
    }
#endif
}

__device__ __forceinline__ const GroupArgs& group_of(const GroupArgs& a) { return a; }
__device__ __forceinline__ const GroupArgs& group_of(const GroupArgsTail& a) { return a.g; }
#ifdef FCN_EXP_TAIL_NOLDS      // (elimination build, timing only with FCN_TAIL_DEBUG=7: the tail variant without its extra LDS)
constexpr int kTailLdsFloats = 0;
#else
constexpr int kTailLdsFloats = kTailRows * 32 + 4;      // the narrow problems' filters over a tile's 32 channels + the arrival flag
#endif

template <typename T, int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int NBUF, bool PF, bool TAIL = false>
__global__ __launch_bounds__((Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::NT)) void conv_fwd_group(const int nprob, const int te0, const int te1, const int te2,
                                                                                  const int te3, const int te4, const int te5, const int te6,
                                                                                  const int te7, const int pool_wgs, const int snake,
                                                                                  const std::conditional_t<TAIL, GroupArgsTail, GroupArgs> a_) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::LDS_FLOATS + (TAIL ? kTailLdsFloats : 0)];
    const GroupArgs& a = group_of(a_);
    // The launch's fixed cost is what counts at M = 784 (9 us launches, 1.5 us of MFMA work).  The problem table (nprob and
    // the exclusive tile prefix of every problem) travels as the kernel's first SCALAR arguments (eleven with the pooling count and the round dealing): the build preloads
    // them into SGPRs at wave launch (-amdgpu-kernarg-preload-count, Makefile), so a workgroup knows its problem without
    // a memory round trip, and the problem itself is ONE round trip to the kernarg segment - written as inline asm because
    // the compiler sinks each field's load to its first use and pays four or five dependent round trips instead.
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    typedef const GroupArgs __attribute__((address_space(4))) * karg_ptr;
    // the GroupArgs copy sits behind the eleven ints in the kernarg segment, at its natural alignment
    static_assert(alignof(GroupArgsTail) == alignof(GroupArgs) && offsetof(GroupArgsTail, g) == 0, "the group leads its tail");
    constexpr size_t kArgsOffset = (11 * sizeof(int) + alignof(GroupArgs) - 1) / alignof(GroupArgs) * alignof(GroupArgs);
    karg_ptr ka = (karg_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kArgsOffset);
    const int head[1 + kMaxGroup] = {nprob, te0, te1, te2, te3, te4, te5, te6, te7};
    static_assert(kMaxGroup == 8, "the tile prefix travels as eight scalar kernel arguments");
    // The poolings fused into this launch take the FIRST pool_wgs workgroups: they are short (a microsecond or two) and start
    // at once, beside the convolution tiles' set-up and first memory latency - queued behind the tiles, as in round 1, they
    // were the launch's tail (inception_4a's 1x1 group: 6.1 us alone, 7.3 us with the module's pool behind it).
    // (pool_wgs < 0, $FCN_POOL_LAST=1: the poolings take the LAST workgroups instead - an experiment repeated in round 4 with the faster
    //  pooling body; pool_wgs is then minus their number and te7 the number of tiles in front of them)
    const int pool_first = pool_wgs > 0 ? pool_wgs : 0;
    if ((int)blockIdx.x < pool_first || (pool_wgs < 0 && (int)blockIdx.x >= te7)) {
        const int w = pool_wgs < 0 ? (int)blockIdx.x - te7 : (int)blockIdx.x;
        if (w < a.pool[0].wg_end) pool_body<T, Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::NT>(a.pool[0], w);
        else pool_body<T, Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::NT>(a.pool[1], w - a.pool[0].wg_end);
        return;
    }
    // Workgroup p and workgroup p + #CUs land on the same CU (HW_ID stamps, tools/conv_timeline.py TIMELINE_PLACEMENT=1), and a
    // group's problems are sorted by chunks per tile, longest first: dealt in index order, the CUs that drew long tiles in one
    // round draw long tiles in every round and the launch ends on them (inception_3b's 3x3 group: 141 chunk-units on the
    // fullest CU against an average of 117).  So the rounds are dealt like cards in a snake: the last (partial) round forward,
    // the one before it backward, and so on; round 0 (where the poolings sit) always forward.
    // snake = rounds << 8 | log2(#CUs), 0 = off, negative = the two-round rotation below; scalar unit only.
    int pos = blockIdx.x;
    if (snake > 0) {
        const int sh = snake & 255, rounds = snake >> 8;
        const int r = pos >> sh, j = pos & ((1 << sh) - 1);
        if (r > 0 && r < rounds - 1 && ((rounds - 1 - r) & 1)) pos = (r << sh) + ((1 << sh) - 1 - j);
    }
    int tile = pos - pool_first;
    if (snake == (int)0x80000000) {
        // (experiment, FCN_CONV_XCD=1) workgroup p runs on XCD p % 8: give every XCD a CONTIGUOUS run of the tile list, so that the tiles
        // sharing A rows (consecutive indices: tile_m = index / tiles_n) meet in one L2 instead of eight
        const int x = pos & 7, sl = pos >> 3, q = te7 >> 3, r = te7 & 7;
        tile = x * q + (x < r ? x : r) + sl;
    } else if (snake < 0) {
        // Two rounds (round 4): -snake = E workgroups more than CUs, so CUs 0 .. E-1 take two tiles each - positions j and #CUs + j -
        // and in index order (longest problem first) those are E of the LONGEST tiles plus the E shortest: the launch ends on
        // them (inception_4a's 3x3 level: 27 + 13 chunks on 19 CUs, 27 or less on the others).  Rotated by E, positions 0 .. E-1 take
        // the E shortest tiles and positions #CUs .. take the next-shortest: the doubled CUs hold two short tiles.
        tile += snake;
        tile += tile < 0 ? te7 : 0;      // (te7 = the group's tile count: tile_end of problems past nprob repeats the last one)
    }
    int pi = 0, begin = 0;
#pragma unroll
    for (int i = 0; i < kMaxGroup - 1; ++i) {      // tile_end is increasing: count the problems that end at or before this tile
        const int end_i = head[1 + i];
        const bool past = i + 1 < nprob && tile >= end_i;
        pi += past ? 1 : 0;
        begin = past ? end_i : begin;
    }
    u32x16 ra, rb;
    u32x8 rc;
    pi = __builtin_amdgcn_readfirstlane(pi);      // (uniform already; keeps the address arithmetic below on the scalar unit)
    begin = __builtin_amdgcn_readfirstlane(begin);
    const ConvP __attribute__((address_space(4)))* pbase = &ka->p[pi];
    asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx8 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                 : "=&s"(ra), "=&s"(rb), "=&s"(rc)
                 : "s"(pbase)
                 : "memory");
    ConvP prob;
    __builtin_memcpy((char*)&prob, &ra, 64);
    __builtin_memcpy((char*)&prob + 64, &rb, 64);
    __builtin_memcpy((char*)&prob + 128, &rc, 32);
    if constexpr (TAIL) {
        // (the tail's fields are fetched from the kernel arguments where they are used.  Fetching its header here, in the problem's own
        //  trip, and carrying it in scalar registers through the kernel was measured SLOWER: 13.8 -> 16.1 / 24.1 -> 27.8 us for the two
        //  launches of inception_5b - the scalar-addressed loader's loop wants those registers)
        tail_kptr tp = (tail_kptr)((const char __attribute__((address_space(4)))*)ka + offsetof(GroupArgsTail, tail));
        conv_body<T, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF, true>(prob, tile - begin, smem, tp);
    } else {
        conv_body<T, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>(prob, tile - begin, smem);
    }
}

template <typename T, int WTM, int WTN, int WAVES_M, int WAVES_N, int WAVES_K, int BK, int NBUF, bool PF>
__global__ __launch_bounds__((Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::NT)) void conv_fwd_one(ConvP p) {
    __shared__ __attribute__((aligned(16))) float smem[Cfg<WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>::LDS_FLOATS];
    conv_body<T, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, NBUF, PF>(p, blockIdx.x, smem);
}

// ---------------------------------------------------------------------------------------------------------------------
// First-layer kernel: 7x7, stride 2, pad 3 on 4-channel pixels (conv1/7x7_s2 of models/deploy.prototxt: 3 image channels +
// 1 pad), up to 64 output channels.  The implicit-GEMM kernel above stages im2col rows; with 16 bytes per pixel every staged
// 16-byte segment is a different filter tap, so its loader decodes a tap per lane and chunk and every workgroup's 32 x 32
// tile lives for seven chunks only (22.8 us at batch 1, a quarter of the matrix-core rate).  Here a workgroup owns an
// 8 x 32 patch of OUTPUT pixels and all output channels:
//   * the 21 x 70 input pixels under the patch and the whole filter bank go to LDS once, through registers (ordinary 16-byte
//     loads return at L1 rate; `buffer_load ... lds` moves one lane per clock and took 2.7 us for the same bytes), and are
//     PACKED on the way: 3 floats per pixel / tap, the pad channel is dropped;
//   * one filter row of the packed image is then 21 consecutive floats for the patch (7 taps x 3 channels, the next output
//     pixel 6 floats further) and for the bank, so K runs over 7 rows x 22 = 154 (147 real + 1 zero weight per row) instead
//     of 7 x 8 x 4 = 224 with 16-byte pixels: an MFMA's two k-values are floats 2j and 2j + 1 of the row;
//   * lane (pixel ox, half h) reads patch[2 oy + r][6 ox + 2 j + h] and bank[n][r][2 j + h] with ds_read_b32 whose address is
//     a per-lane base plus an instruction immediate - no address arithmetic inside the 77 steps x 4 MFMAs, and both reads
//     are bank-conflict free (strides of 6 and 154 floats: 32 distinct even banks, the h = 1 lanes on the odd ones);
//   * each of the 4 waves keeps 2 output rows x 64 channels (four 32 x 32 accumulators) and leaves through LDS so that the
//     stores are 16 bytes per lane, 256 contiguous bytes per pixel.
constexpr int kD7Th = 8, kD7Tw = 32;                         // output rows x columns of a workgroup
constexpr int kD7Ph = 2 * kD7Th + 5, kD7Pw = 2 * kD7Tw + 6;  // input rows x columns under them (+1 column: the zero-weight tap)
constexpr int kD7RowF = kD7Pw * 3;                           // floats per packed patch row
constexpr int kD7PatchF = kD7Ph * kD7RowF;                   // 4410
constexpr int kD7WRowF = 22, kD7WPitchF = 7 * kD7WRowF;      // bank: [64 channels][7 rows][21 + 1 floats]
constexpr int kD7WOff = (kD7PatchF + 3) / 4 * 4;             // float offset of the bank
constexpr int kD7EpiPitch = 68;                              // floats per pixel in the epilogue image (64 channels + bank spread)
constexpr int kD7StageF = kD7WOff + 64 * kD7WPitchF;
constexpr int kD7EpiF = 4 * 2 * 32 * kD7EpiPitch;
constexpr int kD7LdsBytes = (kD7StageF > kD7EpiF ? kD7StageF : kD7EpiF) * 4;
constexpr int kD7Threads = 256;
static_assert(kD7LdsBytes <= 80 * 1024, "first-layer kernel: two workgroups per CU");

__global__ __launch_bounds__(kD7Threads) void conv_first7_kernel(const ConvP p) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float smem[kD7LdsBytes / 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    FCN_STAMP(0);
    const int tiles_x = (p.OW + kD7Tw - 1) / kD7Tw, tiles_y = (p.OH + kD7Th - 1) / kD7Th;
    int t = blockIdx.x;
    const int tx = t % tiles_x;
    t /= tiles_x;
    const int ty = t % tiles_y, n = t / tiles_y;
    const int oy0 = ty * kD7Th, ox0 = tx * kD7Tw;
    const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    // ---- stage: the patch (6 pixels per thread) and the bank (13 taps per thread), all loads in flight before the first write
    constexpr int PI = (kD7Ph * kD7Pw + kD7Threads - 1) / kD7Threads, WI = (64 * 49 + kD7Threads - 1) / kD7Threads;
    v4f px[PI], wv[WI];
    const float* xn = p.x + (size_t)n * p.H * p.W * 4;
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int s = tid + kD7Threads * i;
        const int pr = s / kD7Pw, pc = s - pr * kD7Pw;
        const int iy = iy0 + pr, ix = ix0 + pc;
        const bool ok = s < kD7Ph * kD7Pw && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        px[i] = ok ? *reinterpret_cast<const v4f*>(xn + ((size_t)iy * p.W + ix) * 4) : zero4;
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int tp = tid + kD7Threads * i;
        wv[i] = tp < p.Cout * 49 ? *reinterpret_cast<const v4f*>(p.w + (size_t)tp * 4) : zero4;
    }
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int s = tid + kD7Threads * i;
        if (s < kD7Ph * kD7Pw) {
            float* d = smem + s * 3;      // (row pitch = 70 pixels x 3: the packed image is contiguous)
            d[0] = px[i][0]; d[1] = px[i][1]; d[2] = px[i][2];
        }
    }
#pragma unroll
    for (int i = 0; i < WI; ++i) {
        const int tp = tid + kD7Threads * i;
        if (tp < 64 * 49) {
            const int co = tp / 49, rq = tp - co * 49;
            const int r = rq / 7, q = rq - r * 7;
            float* d = smem + kD7WOff + co * kD7WPitchF + r * kD7WRowF + q * 3;
            d[0] = wv[i][0]; d[1] = wv[i][1]; d[2] = wv[i][2];
        }
    }
    for (int i = tid; i < 64 * 7; i += kD7Threads) smem[kD7WOff + i * kD7WRowF + 21] = 0.f;      // the zero weight that pairs with float 20
    // bias of the 4 channels this lane stores in the epilogue
    const int c4 = (lane & 15) * 4;
    const v4f bias_v = (p.bias && c4 < p.Cout) ? *reinterpret_cast<const v4f*>(p.bias + c4) : zero4;
    FCN_STAMP(1);
    // ---- per-lane fragment bases (floats): A = patch[2 (2 wave + mt) + r][6 ox + 2 j + h], B = bank[nl + 32 nt][r][2 j + h]
    const int nl = lane & 31, h = lane >> 5;
    const float* a_base = smem + (4 * wave) * kD7RowF + 6 * nl + h;
    const float* b_base = smem + kD7WOff + nl * kD7WPitchF + h;
    typedef float v16f __attribute__((ext_vector_type(16)));
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    __syncthreads();
    FCN_STAMP(2);
    auto frag = [&](const int step, float (&a)[2], float (&b)[2]) {      // step = 11 r + j
        const int r = step / 11, j = step - 11 * r;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) a[mt] = a_base[(2 * mt + r) * kD7RowF + 2 * j];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) b[nt] = b_base[nt * 32 * kD7WPitchF + r * kD7WRowF + 2 * j];
    };
    float fa[3][2], fb[3][2];      // fragments of steps s, s + 1, s + 2: two steps (8 MFMAs = 512 cycles) of LDS latency cover
    frag(0, fa[0], fb[0]);
    frag(1, fa[1], fb[1]);
#pragma unroll
    for (int step = 0; step < 77; ++step) {
        if (step + 2 < 77) frag(step + 2, fa[(step + 2) % 3], fb[(step + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise sinks the reads to their first use and waits there)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[step % 3][mt], fb[step % 3][nt], acc[mt][nt], 0, 0, 0);
    }
    FCN_STAMP(3);
    // ---- epilogue through LDS: accumulator register v of lane (nl, h) is pixel ox = (v & 3) + 8 (v >> 2) + 4 h, channel nl (+32)
    __syncthreads();      // every wave is done with the patch and the bank
    float* epi = smem + wave * (2 * 32 * kD7EpiPitch);
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int v = 0; v < 16; ++v) epi[(mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * kD7EpiPitch + nl + 32 * nt] = acc[mt][nt][v];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave reads back its own image only)
    __builtin_amdgcn_sched_barrier(0);
    const bool do_relu = (p.flags & FCN_CONV_RELU) != 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int oy = oy0 + 2 * wave + mt;
        float* yrow = p.y + ((size_t)(n * p.OH + oy) * p.OW + ox0) * p.y_cstride + p.y_coffset + c4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ox = (lane >> 4) + 4 * i;
            v4f v = *reinterpret_cast<const v4f*>(epi + (mt * 32 + ox) * kD7EpiPitch + c4) + bias_v;
            if (do_relu)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            if (oy < p.OH && ox0 + ox < p.OW && c4 < p.Cout) *reinterpret_cast<v4f*>(yrow + (size_t)ox * p.y_cstride) = v;
        }
    }
#ifdef FCN_CONV_STAMPS
    FCN_STAMP(5);
    wait_vmcnt<0>();
    FCN_STAMP(6);
#endif
#endif
}

// The same layer in the half-float engine (BASELINE configs[4], batch 32): the image holds 8 halves per pixel (3 channels,
// two constant-1 channels that carry the folded Power shift, 3 zeros - Engine._packed_weight), so a pixel IS one lane's
// operand of v_mfma_f32_32x32x16_f16 (8 k-values per half-wave): lane (pixel ox, half h) feeds tap q + 4 h of filter row r
// with ONE ds_read_b128 at a per-lane base + immediate, 28 steps x 4 MFMAs per wave and tile.  The implicit-GEMM kernel
// ran this layer at 239 us per 32 frames (a tap decode per lane and chunk, a seventh of the forward).  Workgroups are
// PERSISTENT and the filters are STATIONARY IN REGISTERS: a lane always multiplies by the same 56 filter segments (its two
// output channels x 7 rows x 4 tap pairs = 224 VGPRs, one wave per SIMD anyway), loaded once; with the bank in LDS the 32-cycle
// half-float MFMA waited on fragment reads (four 16-byte reads per four MFMAs: the LDS array, not the matrix core, set the
// pace - 114 us).  The workgroup then walks its share of the 8 x 32 pixel tiles; the next tile's patch is fetched into
// registers before the MFMA loop and written to the other patch buffer behind it, the accumulators get bias and ReLU in f32,
// are rounded to halves once, and leave through LDS as 16-byte stores.  N-tile 0 holds
// the even output channels and N-tile 1 the odd ones, so a lane's two results of a pixel are neighbours and go out as one dword.
constexpr int kH7PatchSlots = (kD7Ph * kD7Pw + 63) / 64 * 64;      // 16-byte slots per patch buffer
constexpr int kH7EpiOff = 2 * kH7PatchSlots * 16;                // bytes: two patch buffers, then the epilogue image
constexpr int kH7EpiPitch = 144;                                 // bytes per pixel in the epilogue image: 64 halves + 16 (bank spread)
constexpr int kH7LdsBytes = kH7EpiOff + 4 * 64 * kH7EpiPitch;
static_assert(kH7LdsBytes <= 160 * 1024, "half-float first-layer kernel: LDS image");

__global__ __launch_bounds__(kD7Threads) void conv_first7_f16_kernel(const ConvP p, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) char smem[kH7LdsBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.OW + kD7Tw - 1) / kD7Tw, tiles_y = (p.OH + kD7Th - 1) / kD7Th;
    const f16_t* xh = reinterpret_cast<const f16_t*>(p.x);
    const f16_t* wh = reinterpret_cast<const f16_t*>(p.w);
    f16_t* yh = reinterpret_cast<f16_t*>(p.y);
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    constexpr int PI = (kD7Ph * kD7Pw + kD7Threads - 1) / kD7Threads;
    // where a tile's patch comes from: the 16-byte pixels of this thread's slots (zeros outside the image / past the last tile)
    auto fetch_patch = [&](const int tile, v4f (&px)[PI]) {
        int t = tile;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y, n = t / tiles_y;
        const int iy0 = 2 * ty * kD7Th - 3, ix0 = 2 * tx * kD7Tw - 3;
        const f16_t* xn = xh + (size_t)n * p.H * p.W * 8;
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int s = tid + kD7Threads * i;
            const int pr = s / kD7Pw, pc = s - pr * kD7Pw;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = tile < ntiles && s < kD7Ph * kD7Pw && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            px[i] = ok ? *reinterpret_cast<const v4f*>(xn + ((size_t)iy * p.W + ix) * 8) : zero4;
        }
    };
    auto store_patch = [&](const int buf, const v4f (&px)[PI]) {
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int s = tid + kD7Threads * i;
            if (s < kD7Ph * kD7Pw) *reinterpret_cast<v4f*>(smem + (buf * kH7PatchSlots + s) * 16) = px[i];
        }
    };
    // ---- once per workgroup: this lane's filter segments (channel 2 nl + nt, row r, tap q + 4 h; tap 7 does not exist: zeros)
    //      and the first patch
    const int nl = lane & 31, h = lane >> 5;
    v4f px[PI];
    int tile = blockIdx.x;
    fetch_patch(tile, px);
    v4f breg[28][2];
#pragma unroll
    for (int step = 0; step < 28; ++step)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int r = step >> 2, q = (step & 3) + 4 * h, co = 2 * nl + nt;
            breg[step][nt] = (q < 7 && co < p.Cout) ? *reinterpret_cast<const v4f*>(wh + ((size_t)co * 49 + r * 7 + q) * 8) : zero4;
        }
    store_patch(0, px);
    // ---- per-lane constants: fragment base (bytes), bias of this lane's two channels (2 nl, 2 nl + 1)
    const char* a_lane = smem + ((4 * wave) * kD7Pw + 2 * nl + 4 * h) * 16;
    const float bias0 = (p.bias && 2 * nl < p.Cout) ? p.bias[2 * nl] : 0.f;
    const float bias1 = (p.bias && 2 * nl + 1 < p.Cout) ? p.bias[2 * nl + 1] : 0.f;
    const bool do_relu = (p.flags & FCN_CONV_RELU) != 0;
    char* epi = smem + kH7EpiOff + wave * (64 * kH7EpiPitch);
    typedef float v16f __attribute__((ext_vector_type(16)));
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile += gridDim.x) {
        fetch_patch(tile + (int)gridDim.x, px);      // the next tile's pixels travel while this tile multiplies
        const char* a_base = a_lane + buf * kH7PatchSlots * 16;
        v16f acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        auto frag = [&](const int step, v4f (&a)[2]) {
            const int r = step >> 2, q = step & 3;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const v4f*>(a_base + ((2 * mt + r) * kD7Pw + q) * 16);
        };
        v4f fa[3][2];
        frag(0, fa[0]);
        frag(1, fa[1]);
#pragma unroll
        for (int step = 0; step < 28; ++step) {
            if (step + 2 < 28) frag(step + 2, fa[(step + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fa[step % 3][mt]), __builtin_bit_cast(v8h, breg[step][nt]),
                                                                         acc[mt][nt], 0, 0, 0);
        }
        // ---- epilogue: register v of lane (nl, h) is pixel ox = (v & 3) + 8 (v >> 2) + 4 h, channels 2 nl (N-tile 0) and 2 nl + 1 (N-tile 1)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                float e0 = acc[mt][0][v] + bias0, e1 = acc[mt][1][v] + bias1;
                if (do_relu) {
                    e0 = fmaxf(e0, 0.f);
                    e1 = fmaxf(e1, 0.f);
                }
                typedef f16_t v2h __attribute__((ext_vector_type(2)));
                const v2h pk = {(f16_t)e0, (f16_t)e1};
                *reinterpret_cast<v2h*>(epi + (mt * 32 + (v & 3) + 8 * (v >> 2) + 4 * h) * kH7EpiPitch + nl * 4) = pk;
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave reads back its own image only)
        __builtin_amdgcn_sched_barrier(0);
        {
            int t = tile;
            const int tx = t % tiles_x;
            t /= tiles_x;
            const int ty = t % tiles_y, n = t / tiles_y;
            const int c8 = (lane & 7) * 8;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int oy = ty * kD7Th + 2 * wave + mt;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ox = (lane >> 3) + 8 * i;
                    const v4f val = *reinterpret_cast<const v4f*>(epi + (mt * 32 + ox) * kH7EpiPitch + c8 * 2);
                    if (oy < p.OH && tx * kD7Tw + ox < p.OW && c8 < p.Cout)
                        *reinterpret_cast<v4f*>(yh + ((size_t)(n * p.OH + oy) * p.OW + tx * kD7Tw + ox) * p.y_cstride + p.y_coffset + c8) = val;
                }
            }
        }
        store_patch(buf ^ 1, px);      // (behind the epilogue: the fetch has had the MFMA loop and the epilogue to arrive)
        // the next patch is in LDS and everybody is done with this one: a raw barrier behind the wave's own LDS writes -
        // __syncthreads() would also wait for the tile's global stores (a write latency per tile)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf ^= 1;
    }
#endif
}

// The half-float first layer on an image whose channels 3 and 4 are the constant 1 (FCN_CONV_IMAGE_ONES: the engine's folded
// Power shift, Engine._packed_weight) - round 3.  The kernel above multiplies 8-half pixels of which three carry data: 112 MFMAs per
// wave and tile, 224 filter registers per lane, ONE wave per SIMD (405 VGPRs) - nothing overlaps a tile's epilogue (2.9 of its 4.6 us).
// With the flag the two constant channels need no multiplication at all: their contribution to an output is the sum of the filter's
// shift terms over the taps that lie inside the image - a per-channel constant for every pixel whose window is inside (added to the
// bias, in f32) and a short sum over a 49 x 64 table in LDS for the pixels within 3 columns / rows of the border.  What is left is
// b, g, r: a pixel shrinks to 4 halves (8 bytes, the fourth zero) in the LDS patch, one 16-byte fragment read covers TWO neighbouring
// taps (the stride of 2 keeps it aligned), a filter row takes 2 MFMA steps instead of 4: 56 MFMAs per wave and tile, 112 filter
// registers - and with <= 256 VGPRs TWO workgroups share a CU, so one multiplies while the other stores.  Same tile walk, same
// epilogue through the LDS image (whole 128-byte lines per store instruction: DESIGN.md 4.7 (b)).
constexpr int kX4PatchBytes = kD7Ph * kD7Pw * 8;                              // 21 x 70 pixels of 8 bytes; rows of 560 bytes (16-byte multiples)
constexpr int kX4PatchInstr = 48;                                             // LDS-DMA instructions per patch (256 bytes each; 46 carry pixels)
constexpr int kX4PatchPitch = kX4PatchInstr * 256;                            // bytes between the patch buffers
constexpr int kX4NBuf = 3;                                                    // patch buffers: the tile being multiplied and two on their way
constexpr int kX4EpiOff = kX4NBuf * kX4PatchPitch;                            // then the epilogue image (one output row per wave at a time), then the table
constexpr int kX4EpiPitch = 144;
constexpr int kX4TabOff = kX4EpiOff + 4 * 32 * kX4EpiPitch;
constexpr int kX4LdsBytes = kX4TabOff + 64 * 64 * 4;      // prefix sums P[r][q][channel], r, q = 0 .. 7, of the shift terms
static_assert(kX4PatchBytes % 16 == 0 && (kD7Pw * 8) % 16 == 0 && kX4PatchBytes <= 46 * 256, "fragment reads are 16-byte aligned; 46 instructions cover a patch");
static_assert(2 * kX4LdsBytes <= 160 * 1024 && 49 * 64 * 4 <= kX4EpiOff, "two workgroups per CU; the raw shift table borrows the patch buffers");

#ifdef FCN_EXP_X4_FREE      // (experiment: no register cap - one workgroup per CU)
#define FCN_X4_ATTR
#else
#define FCN_X4_ATTR __attribute__((amdgpu_waves_per_eu(2, 2)))
#endif
#ifndef FCN_X4_LINEAR
#define FCN_X4_LINEAR 1      // 1: tiles b, b + G, ..; 0 (experiment): runs of consecutive tiles per XCD - measured no gain (97 vs 100 us):
#endif                       // the kernel is bound by its own vector instructions, not by memory (elimination builds: 90 us without loads AND stores)
__global__ __launch_bounds__(kD7Threads) FCN_X4_ATTR void conv_first7_f16x4_kernel(const ConvP p, const int ntiles) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) char smem[kX4LdsBytes];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = (p.OW + kD7Tw - 1) / kD7Tw, tiles_y = (p.OH + kD7Th - 1) / kD7Th;
    const f16_t* wh = reinterpret_cast<const f16_t*>(p.w);
    f16_t* yh = reinterpret_cast<f16_t*>(p.y);
    // (filter segments travel as INTEGER words: two halves seen as a float are a denormal once the upper one is masked away, and
    //  float moves may flush it)
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2u zero2 = {0u, 0u};
    // A tile's patch goes from HBM to LDS by LDS-DMA, TWO tiles ahead (with the patch in registers a workgroup had one tile's bytes
    // in flight and waited a memory latency per tile - 5.7 us per tile where the MFMAs take 0.85): a lane moves one dword - half of a
    // pixel's (b, g, r, 1) - so two neighbouring lanes fill a pixel's 8 bytes; instruction k of the patch carries slots 32 k .. 32 k + 31.
    // The pixel's constant channel arrives as it is (1 inside the image): the filters' fourth half is zero.  Pixels outside the
    // image / past the last tile take an out-of-range offset: zeros.
    constexpr int OOB = (int)0x80000000u;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((long long)p.N * p.H * p.W * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
        yh, 0, (int)((((long long)p.N * p.OH * p.OW - 1) * p.y_cstride + p.y_coffset + p.Cout) * 2), 0x00020000);
    auto issue_patch = [&](const int tile, const int buf) __attribute__((always_inline)) {
        int t = tile;
        const int tx = t % tiles_x;
        t /= tiles_x;
        const int ty = t % tiles_y, n = t / tiles_y;
        const int iy0 = 2 * ty * kD7Th - 3, ix0 = 2 * tx * kD7Tw - 3;
        const int img = n * p.H;
#pragma unroll
        for (int i = 0; i < kX4PatchInstr / 4; ++i) {
            const int k = wave + 4 * i;
            const int sl = k * 32 + (lane >> 1);
            const int pr = sl / kD7Pw, pc = sl - pr * kD7Pw;
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = tile < ntiles && sl < kD7Ph * kD7Pw && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            int vo = ok ? ((img + iy) * p.W + ix) * 16 + (lane & 1) * 4 : OOB;
            asm volatile("" : "+v"(vo));
#ifndef FCN_X4_NOLOAD      // (elimination builds: timing only)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(smem + buf * kX4PatchPitch + k * 256), 4, vo, 0, 0, 0);
#endif
        }
    };
    // ---- once per workgroup: the shift table T[tap][channel] = w[channel][tap][3] + w[channel][tap][4] (the two halves of the
    //      folded Power shift), this lane's filter segments (channel 2 nl + nt, filter row r, taps 4 hs + 2 h and + 1; tap 7: zeros)
    //      and its 2-D prefix sums P[r][q] = sum of T over filter rows < r and taps < q: the taps of a window that lie inside the image
    //      are a rectangle [rlo, rhi) x [qlo, qhi), so a border pixel's shift is four table reads (a loop over up to 49 taps per pixel
    //      made the tiles of the first image rows the slowest of the launch)
    float* traw = reinterpret_cast<float*>(smem);                  // (the patch buffers are free until the first patch is asked for)
    float* tab = reinterpret_cast<float*>(smem + kX4TabOff);
    for (int e = tid; e < 49 * 64; e += kD7Threads) {
        const int t = e >> 6, co = e & 63;
        traw[e] = co < p.Cout ? (float)wh[((size_t)co * 49 + t) * 8 + 3] + (float)wh[((size_t)co * 49 + t) * 8 + 4] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < 8 * 64; e += kD7Threads) {      // one (r, channel) per item: the running sums along q, rows < r in row order
        const int r = e >> 6, co = e & 63;
        float run = 0.f;
        tab[(r * 8) * 64 + co] = 0.f;
        for (int q = 0; q < 7; ++q) {
            for (int rr = 0; rr < r; ++rr) run += traw[(rr * 7 + q) * 64 + co];
            tab[(r * 8 + q + 1) * 64 + co] = run;
        }
    }
    __syncthreads();      // everybody is done with the raw table: the patch buffers may fill
    const int nl = lane & 31, h = lane >> 5;
    // Tile walk: b, b + G, ...  (FCN_X4_LINEAR = 0: blocks b and b + 8 share an XCD, so the blocks of one XCD take RUNS of consecutive
    // tiles - neighbouring tiles share 5 of their patch's 21 rows and 6 of its 70 columns; step k of block (x = b & 7, j = b >> 3) is tile
    // (8 k + x) J + j, J = blocks per XCD.)
    const int G = (int)gridDim.x, xcd = (int)blockIdx.x & 7, J = G >> 3;
    const bool runs = (G & 7) == 0 && !(FCN_X4_LINEAR);
    auto tile_of = [&](const int kk) __attribute__((always_inline)) { return runs ? (8 * kk + xcd) * J + ((int)blockIdx.x >> 3) : (int)blockIdx.x + kk * G; };
    int step_k = 0;
    int tile = tile_of(0);
    issue_patch(tile, 0);
    issue_patch(tile_of(1), 1);
    v4u breg[14][2];
#pragma unroll
    for (int step = 0; step < 14; ++step)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int r = step >> 1, q0 = 4 * (step & 1) + 2 * h, co = 2 * nl + nt;
            v2u lo = zero2, hi = zero2;
            if (co < p.Cout) {
                lo = *reinterpret_cast<const v2u*>(wh + ((size_t)co * 49 + r * 7 + q0) * 8);
                if (q0 + 1 < 7) hi = *reinterpret_cast<const v2u*>(wh + ((size_t)co * 49 + r * 7 + q0 + 1) * 8);
            }
            breg[step][nt] = v4u{lo[0], lo[1] & 0xffffu, hi[0], hi[1] & 0xffffu};      // (the fourth half - the constant channel's filter - is not multiplied)
        }
    // ---- per-lane constants: fragment base (bytes), bias + the whole shift of this lane's two channels (windows inside the image)
    const char* a_lane = smem + ((4 * wave) * kD7Pw + 2 * nl + 2 * h) * 8;
    const float bias0 = (p.bias && 2 * nl < p.Cout) ? p.bias[2 * nl] : 0.f;
    const float bias1 = (p.bias && 2 * nl + 1 < p.Cout) ? p.bias[2 * nl + 1] : 0.f;
    const v2f tall = *reinterpret_cast<const v2f*>(tab + (7 * 8 + 7) * 64 + 2 * nl);
    const float in0 = bias0 + tall[0], in1 = bias1 + tall[1];
    const bool do_relu = (p.flags & FCN_CONV_RELU) != 0;
    char* epi = smem + kX4EpiOff + wave * (32 * kX4EpiPitch);
    typedef float v16f __attribute__((ext_vector_type(16)));
    // the filter loads above are older than nothing that matters: wait for the FIRST patch (this wave's 12 instructions of the second
    // one may stay in flight), then publish it
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kX4PatchInstr / 4) : "memory");
    __syncthreads();
    int buf = 0;
    for (; tile < ntiles; tile = tile_of(++step_k)) {
        {      // the patch two tiles ahead goes into the buffer the previous tile used (everybody left it at the last barrier)
            const int b2 = buf + 2 >= kX4NBuf ? buf + 2 - kX4NBuf : buf + 2;
            issue_patch(tile_of(step_k + 2), b2);
        }
        const char* a_base = a_lane + buf * kX4PatchPitch;
        v16f acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        auto frag = [&](const int step, v4u (&a)[2]) {
            const int r = step >> 1, hs = step & 1;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) a[mt] = *reinterpret_cast<const v4u*>(a_base + ((2 * mt + r) * kD7Pw + 4 * hs) * 8);
        };
        v4u fa[3][2];
        frag(0, fa[0]);
        frag(1, fa[1]);
#pragma unroll
        for (int step = 0; step < 14; ++step) {
            if (step + 2 < 14) frag(step + 2, fa[(step + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#ifdef FCN_X4_NOMFMA
                    acc[mt][nt][0] += __builtin_bit_cast(float, fa[step % 3][mt][0] ^ breg[step][nt][0]);
#else
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, fa[step % 3][mt]), __builtin_bit_cast(v8h, breg[step][nt]),
                                                                         acc[mt][nt], 0, 0, 0);
#endif
        }
        int tq = tile;
        const int tx = tq % tiles_x;
        tq /= tiles_x;
        const int ty = tq % tiles_y, n = tq / tiles_y;
        // windows of this tile that reach outside the image: rows 2 oy - 3 .. + 3, columns 2 ox - 3 .. + 3
        const bool border = ty == 0 || tx == 0 || 2 * (ty * kD7Th + kD7Th - 1) + 3 >= p.H || 2 * (tx * kD7Tw + kD7Tw - 1) + 3 >= p.W;
        // ---- epilogue, one output row of the wave at a time: register v of lane (nl, h) is pixel ox = (v & 3) + 8 (v >> 2) + 4 h,
        //      channels 2 nl (N-tile 0) and 2 nl + 1 (N-tile 1); through the LDS image whole 128-byte lines leave per store instruction
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int oy = ty * kD7Th + 2 * wave + mt;
            const int rlo = max(0, 3 - 2 * oy), rhi = min(7, p.H + 3 - 2 * oy);
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int oxl = (v & 3) + 8 * (v >> 2) + 4 * h;
                float b0 = in0, b1 = in1;
                if (border) {
                    const int ox = tx * kD7Tw + oxl;
                    const int qlo = max(0, 3 - 2 * ox), qhi = min(7, p.W + 3 - 2 * ox);
                    if (rlo > 0 || rhi < 7 || qlo > 0 || qhi < 7) {      // the shift terms of the taps inside the image only
                        const int rh = max(rhi, rlo), qh = max(qhi, qlo);      // (pixels past the image: an empty rectangle)
                        const v2f a = *reinterpret_cast<const v2f*>(tab + (rh * 8 + qh) * 64 + 2 * nl), bq = *reinterpret_cast<const v2f*>(tab + (rlo * 8 + qh) * 64 + 2 * nl),
                                  cq = *reinterpret_cast<const v2f*>(tab + (rh * 8 + qlo) * 64 + 2 * nl), dq = *reinterpret_cast<const v2f*>(tab + (rlo * 8 + qlo) * 64 + 2 * nl);
                        b0 = bias0 + ((a[0] - bq[0]) - (cq[0] - dq[0]));
                        b1 = bias1 + ((a[1] - bq[1]) - (cq[1] - dq[1]));
                    }
                }
                typedef f16_t v2h __attribute__((ext_vector_type(2)));
                const v2f e = v2f{acc[mt][0][v], acc[mt][1][v]} + v2f{b0, b1};      // (one packed add, one packed conversion, one packed maximum:
                v2h pk = {(f16_t)e[0], (f16_t)e[1]};                                  //  the kernel is bound by its vector instructions)
                if (do_relu) pk = __builtin_elementwise_max(pk, v2h{(f16_t)0.f, (f16_t)0.f});      // (max after rounding = rounding after max)
                *reinterpret_cast<v2h*>(epi + oxl * kX4EpiPitch + nl * 4) = pk;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the wave reads back its own image only; LDS operations of a wave run in order)
            __builtin_amdgcn_sched_barrier(0);
            const int c8 = (lane & 7) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ox = (lane >> 3) + 8 * i;
                const v4u val = *reinterpret_cast<const v4u*>(epi + ox * kX4EpiPitch + c8 * 2);
                // (always ONE store instruction per i - lanes without an output take an out-of-range offset - so that the wait below can
                //  count this iteration's stores whatever the tile)
                const bool okp = oy < p.OH && tx * kD7Tw + ox < p.OW && c8 < p.Cout;
                int vo = okp ? (((n * p.OH + oy) * p.OW + tx * kD7Tw + ox) * p.y_cstride + p.y_coffset + c8) * 2 : OOB;
                asm volatile("" : "+v"(vo));
#ifdef FCN_X4_NOSTORE
                asm volatile("" ::"v"(val), "v"(vo));
#else
                __builtin_amdgcn_raw_buffer_store_b128(val, ry, vo, 0, 0);
#endif
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the image is rewritten for the next row)
            __builtin_amdgcn_sched_barrier(0);
        }
        // The next tile's patch - asked for one iteration ago, older than this iteration's 12 DMA instructions and 8 stores, and the
        // vector-memory counter retires in order - has landed; everybody is done with this tile's patch: publish / release with one
        // raw barrier (__syncthreads() would also wait for the tile's global stores: a write latency per tile).
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kX4PatchInstr / 4 + 8) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        buf = buf + 1 == kX4NBuf ? 0 : buf + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (patches asked for past the last tile: all-zero loads, but they write this workgroup's LDS)
#endif
}

bool first7_f16_ok(const ConvP& p) {
    return p.kh == 7 && p.kw == 7 && p.stride == 2 && p.pad == 3 && p.Cin == 8 && p.x_cstride == 8 && p.Cout > 32 && p.Cout <= 64 &&
           p.Cout % 8 == 0 && ((p.y_cstride | p.y_coffset) & 7) == 0 && ((uintptr_t)p.y & 15) == 0 &&
           (p.flags & ~(FCN_CONV_RELU | FCN_CONV_IMAGE_ONES)) == FCN_CONV_F16;
}

// Does the first-layer kernel take this problem?
bool first7_ok(const ConvP& p) {
    return p.kh == 7 && p.kw == 7 && p.stride == 2 && p.pad == 3 && p.Cin == 4 && p.x_cstride == 4 && p.Cout > 32 && p.Cout <= 64 &&
           p.Cout % 4 == 0 && ((p.y_cstride | p.y_coffset) & 3) == 0 && ((uintptr_t)p.y & 15) == 0 && (!p.bias || ((uintptr_t)p.bias & 15) == 0) &&
           (p.flags & ~FCN_CONV_RELU) == 0;
}

// ---------------------------------------------------------------------------------------------------------------------
// Narrow 1x1 convolutions (the detection heads: cvg/classifier and bbox/regressor, 4 + 16 outputs over K = 1024 at 784
// pixels - 32 MFLOP).  As 32 x 32 MFMA tiles they are 50 workgroups that walk 32 chunks of K one after the other behind a
// full prologue and epilogue: 8-9 us for 0.2 us of arithmetic, whatever the tile shape (tools/conv_sweep.py heads).  Here K
// is spread over the 64 LANES of a wave instead: a workgroup owns 4 pixels, each of its waves up to 8 output channels of
// one problem, a lane multiplies the same 4-float slice of every 256-float stretch of K (16-byte loads, 1 KiB contiguous
// per wave-instruction) into 4 x 8 partial sums, and the 64 partials of every sum are added in lane order through LDS
// (fixed order: bit-reproducible).  Bias, ReLU and the sigmoid second output as in the tile kernel's epilogue.
constexpr int kDotPx = 4, kDotOut = 8, kDotMaxSlices = 4;
constexpr int kDotRedPx = 4;          // pixels per reduction round (the LDS image holds kDotRedPx x kDotOut sums per wave)
constexpr int kDotRedPitch = 68;      // floats per sum in the reduction image: 64 lane partials + 4 (16-byte rows, banks rotate by 4)
struct DotSlice {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    float* y2;
    int K, x_cstride, y_cstride, y_coffset, y2_cstride, y2_coffset, nout, flags;
};
struct DotArgs {
    int M, nslices;
    DotSlice s[kDotMaxSlices];
};

__global__ __launch_bounds__(64 * kDotMaxSlices) void conv_dot1x1_kernel(const DotArgs a) {
    __shared__ __attribute__((aligned(16))) float red[kDotMaxSlices][kDotRedPx * kDotOut * kDotRedPitch];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    if (wave >= a.nslices) return;      // (no workgroup barrier below: every wave reduces through its own LDS rows)
    DotSlice sl = a.s[0];
#pragma unroll
    for (int i = 1; i < kDotMaxSlices; ++i)
        if (wave == i) sl = a.s[i];
    const int m0 = blockIdx.x * kDotPx;
    const v4f zero4 = {0.f, 0.f, 0.f, 0.f};
    v4f acc[kDotPx][kDotOut];      // (packed FMAs: two multiply-adds per instruction)
#pragma unroll
    for (int p = 0; p < kDotPx; ++p)
#pragma unroll
        for (int o = 0; o < kDotOut; ++o) acc[p][o] = zero4;
    // Four 256-float stretches of K per trip, ALL their loads issued before the first multiply: the launch is a few memory
    // latencies long, and a loop that waits for every stretch pays one per stretch.  (Measured alternatives: scalar partial
    // sums - 8 instead of 2 vector instructions per term, 6.7 us; eight pixels per workgroup - half the filter re-reads but
    // twice the serial work per wave, 9.5 us.)
    constexpr int U = 4;
    for (int k0 = lane * 4; k0 < sl.K; k0 += 256 * U) {
        v4f xa[U][kDotPx], wb[U][kDotOut];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 256 * u;
            const bool in = k < sl.K;
#pragma unroll
            for (int p = 0; p < kDotPx; ++p)
                xa[u][p] = in && m0 + p < a.M ? *reinterpret_cast<const v4f*>(sl.x + (size_t)(m0 + p) * sl.x_cstride + k) : zero4;
#pragma unroll
            for (int o = 0; o < kDotOut; ++o) wb[u][o] = in && o < sl.nout ? *reinterpret_cast<const v4f*>(sl.w + (size_t)o * sl.K + k) : zero4;
        }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int p = 0; p < kDotPx; ++p)
#pragma unroll
                for (int o = 0; o < kDotOut; ++o) acc[p][o] += xa[u][p] * wb[u][o];
    }
    float* r = red[wave];
#pragma unroll
    for (int round = 0; round < kDotPx / kDotRedPx; ++round) {      // (a wave's LDS operations complete in order: no wait between rounds)
#pragma unroll
        for (int p = 0; p < kDotRedPx; ++p)
#pragma unroll
            for (int o = 0; o < kDotOut; ++o) {
                const v4f t = acc[round * kDotRedPx + p][o];
                r[(p * kDotOut + o) * kDotRedPitch + lane] = (t[0] + t[1]) + (t[2] + t[3]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the wave reads its own rows only
        __builtin_amdgcn_sched_barrier(0);
        if (lane < kDotRedPx * kDotOut) {
            const int p = lane / kDotOut, o = lane % kDotOut;
            v4f q = *reinterpret_cast<const v4f*>(r + lane * kDotRedPitch);      // sixteen 16-byte reads in flight, a fixed tree: the same sum in every run
#pragma unroll
            for (int j = 1; j < 16; ++j) q += *reinterpret_cast<const v4f*>(r + lane * kDotRedPitch + 4 * j);
            float v = (q[0] + q[1]) + (q[2] + q[3]);
            const int m = m0 + round * kDotRedPx + p;
            if (m < a.M && o < sl.nout) {
                if (sl.bias) v += sl.bias[o];
                if (sl.flags & FCN_CONV_RELU) v = fmaxf(v, 0.f);
                sl.y[(size_t)m * sl.y_cstride + sl.y_coffset + o] = v;
                if (sl.flags & FCN_CONV_SIGMOID2) sl.y2[(size_t)m * sl.y2_cstride + sl.y2_coffset + o] = 1.f / (1.f + expf(-v));
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Does the lane-split kernel take this group?  (all problems 1x1 / stride 1 / unpadded float32 over the same pixels, at most
// four slices of eight output channels in all)
bool dot1x1_ok(const ConvP* ps, int n) {
    int slices = 0;
    for (int i = 0; i < n; ++i) {
        const ConvP& p = ps[i];
        if (p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || (p.flags & ~(FCN_CONV_RELU | FCN_CONV_SIGMOID2)) != 0) return false;
        if (p.M != ps[0].M || p.Cin % 4 != 0) return false;
        slices += cdiv(p.Cout, kDotOut);
    }
    return slices <= kDotMaxSlices;
}

// ---- host side -------------------------------------------------------------------------------

// tile configurations: X(index, WTM, WTN, WAVES_M, WAVES_N, WAVES_K, BK, ring slots, fragment prefetch)
#define FCN_CONV_CONFIGS(X)            \
    X(0, 2, 2, 2, 2, 1, 32, 4, false)  \
    X(1, 2, 1, 2, 2, 1, 32, 4, false)  \
    X(2, 1, 1, 2, 2, 1, 32, 4, true)   \
    X(3, 1, 1, 4, 1, 1, 32, 4, true)   \
    X(4, 1, 1, 2, 1, 2, 64, 4, true)   \
    X(5, 1, 1, 1, 1, 4, 64, 4, true)   \
    X(6, 1, 1, 2, 2, 1, 64, 4, true)   \
    X(7, 1, 1, 1, 1, 4, 64, 6, true)   \
    X(8, 1, 1, 1, 1, 8, 64, 4, true)   \
    X(9, 1, 1, 2, 2, 2, 64, 4, true)   \
    X(10, 1, 1, 2, 2, 2, 32, 4, true)  \
    X(11, 2, 1, 2, 2, 2, 32, 4, true)  \
    X(12, 1, 1, 2, 1, 4, 64, 4, true)  \
    X(13, 2, 2, 2, 2, 2, 32, 3, false) \
    X(14, 1, 1, 2, 2, 1, 32, 3, false) \
    X(15, 2, 1, 2, 2, 1, 32, 3, false) \
    X(16, 1, 2, 2, 2, 1, 32, 3, false) \
    X(17, 1, 1, 1, 1, 4, 64, 3, false) \
    X(18, 1, 1, 1, 1, 4, 32, 4, true)  \
    X(19, 1, 1, 2, 1, 2, 32, 4, true)  \
    X(20, 1, 1, 2, 1, 2, 32, 3, false) \
    X(21, 1, 1, 1, 1, 4, 32, 3, false) \
    X(22, 1, 1, 4, 1, 1, 32, 3, false) \
    X(23, 1, 1, 1, 1, 4, 32, 16 + 4, true)  \
    X(24, 1, 1, 1, 1, 4, 64, 16 + 4, true)  \
    X(25, 1, 1, 2, 1, 2, 32, 16 + 4, true)  \
    X(26, 1, 1, 2, 2, 1, 32, 16 + 4, true)  \
    X(27, 1, 1, 1, 1, 4, 32, 16 + 6, true)  \
    X(28, 1, 1, 2, 2, 1, 64, 16 + 4, true)  \
    X(29, 2, 1, 2, 2, 1, 32, 16 + 4, true)

struct TileCfg { int bm, bn, bk; bool prefetch; };
constexpr int kCfgThreads[] = {
#define X(I, A, B, C_, D, E, F, G, H) Cfg<A, B, C_, D, E, F, G, H>::NT,
    FCN_CONV_CONFIGS(X)
#undef X
};
constexpr int kCfgLdsBytes[] = {
#define X(I, A, B, C_, D, E, F, G, H) Cfg<A, B, C_, D, E, F, G, H>::LDS_FLOATS * 4,
    FCN_CONV_CONFIGS(X)
#undef X
};
constexpr bool kCfgRoles[] = {
#define X(I, A, B, C_, D, E, F, G, H) Cfg<A, B, C_, D, E, F, G, H>::ROLES,
    FCN_CONV_CONFIGS(X)
#undef X
};
constexpr int kCfgWavesK[] = {
#define X(I, A, B, C_, D, E, F, G, H) E,
    FCN_CONV_CONFIGS(X)
#undef X
};
// Round 4, tried and dropped (profiles/experiments/r04_sweep_cfg_16waves.txt, r04_sweep_loadx.txt): sixteen waves on one tile with
// 64-float chunks (K over eight waves, or a 64 x 32 tile) - 5-15 % slower than configuration 23 on every batch-1 launch, like every
// other 64-float-chunk shape; and the split-role shapes 23 / 26 with TWICE the loading waves, one LDS-DMA piece per wave and chunk
// (Cfg::LOADX, kept: ring slots >= 32) - 40-60 % slower: the loop is not bound by a loading wave's issue rate.
constexpr int kNumTileCfg = 30;            // configurations of the implicit-GEMM kernel (the X table)
constexpr int kFirst7Cfg = kNumTileCfg;    // conv_first7_kernel: single 7x7 / stride 2 / 4-channel problems only (first7_ok)
constexpr int kDot1x1Cfg = kNumTileCfg + 1;  // conv_dot1x1_kernel: groups of narrow 1x1 problems only (dot1x1_ok)
constexpr int kStreamCfg0 = kNumTileCfg + 2;  // conv_stream_f16 (conv_stream.hip): persistent half-float streaming kernel, configurations 32 ..
inline int num_cfgs() { return kStreamCfg0 + stream_num_cfgs(); }
inline bool is_stream_cfg(int cfg) { return cfg >= kStreamCfg0 && cfg < num_cfgs(); }
constexpr TileCfg kCfgs[kNumTileCfg] = {
#define X(I, A, B, C_, D, E, F, G, H) {Cfg<A, B, C_, D, E, F, G, H>::BM, Cfg<A, B, C_, D, E, F, G, H>::BN, F, H},
    FCN_CONV_CONFIGS(X)
#undef X
};

int validate(const fcn_conv_desc& d) {
    FCN_REQUIRE(d.x && d.w && d.y, FCN_E_ARG, "conv: null x/w/y");
    FCN_REQUIRE(d.N > 0 && d.H > 0 && d.W > 0 && d.Cin > 0 && d.Cout > 0 && d.kh > 0 && d.kw > 0 && d.stride > 0 && d.pad >= 0,
                FCN_E_ARG, "conv: non-positive extent");
    const int eps = (d.flags & FCN_CONV_F16) ? 8 : 4;      // elements per 16-byte segment
    FCN_REQUIRE(d.Cin % eps == 0 && d.x_cstride % eps == 0 && d.x_cstride >= d.Cin, FCN_E_ALIGN,
                "conv: Cin (%d) and x_cstride (%d) must be multiples of %d (pad the input channels)", d.Cin, d.x_cstride, eps);
    FCN_REQUIRE((d.flags & FCN_CONV_F16) || !(d.flags & FCN_CONV_OUT_F32), FCN_E_ARG, "conv: FCN_CONV_OUT_F32 only qualifies FCN_CONV_F16");
    FCN_REQUIRE(!(d.flags & FCN_CONV_F16) || !(d.flags & FCN_CONV_OUT_F16), FCN_E_ARG, "conv: FCN_CONV_OUT_F16 only qualifies float32 inputs");
    FCN_REQUIRE(!(d.flags & FCN_CONV_IMAGE_ONES) || ((d.flags & FCN_CONV_F16) && d.Cin == 8 && d.x_cstride == 8), FCN_E_ARG,
                "conv: FCN_CONV_IMAGE_ONES describes an 8-half pixel image (FCN_CONV_F16, Cin = x_cstride = 8)");
    FCN_REQUIRE(((uintptr_t)d.x & 15) == 0 && ((uintptr_t)d.w & 15) == 0, FCN_E_ALIGN, "conv: x/w must be 16-byte aligned");
    FCN_REQUIRE(d.OH == (d.H + 2 * d.pad - d.kh) / d.stride + 1 && d.OW == (d.W + 2 * d.pad - d.kw) / d.stride + 1,
                FCN_E_ARG, "conv: OH/OW (%d,%d) do not match floor((H+2p-k)/s)+1", d.OH, d.OW);
    FCN_REQUIRE(d.OH > 0 && d.OW > 0, FCN_E_ARG, "conv: empty output");
    FCN_REQUIRE(d.y_cstride >= d.y_coffset + d.Cout && d.y_coffset >= 0, FCN_E_ARG, "conv: output slice exceeds y_cstride");
    FCN_REQUIRE(d.kh * d.kw < 4096, FCN_E_UNSUPPORTED, "conv: kernel window %dx%d too large", d.kh, d.kw);
    FCN_REQUIRE((long long)d.N * d.H * d.W * d.x_cstride < (1ll << 31) && (long long)d.Cout * d.kh * d.kw * d.Cin < (1ll << 31),
                FCN_E_UNSUPPORTED, "conv: tensor too large for 32-bit element offsets");
    FCN_REQUIRE(d.in_shift == 0.f, FCN_E_UNSUPPORTED, "conv: in_shift is applied by the producer of the input (fcn_nchw_to_nhwc_f32 / fcn_preprocess_bgr8)");
    if (d.flags & (FCN_CONV_SIGMOID2 | FCN_CONV_MASK))
        FCN_REQUIRE(d.y2 && d.y2_cstride >= d.y2_coffset + d.Cout, FCN_E_ARG, "conv: FCN_CONV_SIGMOID2 / FCN_CONV_MASK need y2");
    FCN_REQUIRE(!((d.flags & FCN_CONV_SIGMOID2) && (d.flags & FCN_CONV_MASK)), FCN_E_ARG, "conv: FCN_CONV_SIGMOID2 and FCN_CONV_MASK both use y2");
    FCN_REQUIRE(!((d.flags & FCN_CONV_MASK) && (d.flags & FCN_CONV_F16)), FCN_E_UNSUPPORTED, "conv: FCN_CONV_MASK is a float32 (training) feature");
    FCN_REQUIRE((long long)d.N * d.OH * d.OW < (1ll << 31), FCN_E_UNSUPPORTED, "conv: problem too large for int32 indexing");
    // the kernel decodes pixel and tile indices with multiply-high by host-computed reciprocals (no division code on the device)
    FCN_REQUIRE((long long)d.N * d.OH * d.OW * (d.OW > d.OH ? d.OW : d.OH) < (1ll << 32) && d.Cin <= 16384, FCN_E_UNSUPPORTED,
                "conv: N*OH*OW*max(OH,OW) must stay below 2^32 and Cin at most 16384 (split the batch)");
    return 0;
}

void fill(ConvP& p, const fcn_conv_desc& d, const float* zero_page) {
    p.x = d.x; p.w = d.w; p.bias = d.bias; p.y = d.y; p.y2 = d.y2;
    p.N = d.N; p.H = d.H; p.W = d.W; p.Cin = d.Cin; p.x_cstride = d.x_cstride;
    p.Cout = d.Cout; p.kh = d.kh; p.kw = d.kw; p.pad = d.pad; p.stride = d.stride; p.OH = d.OH; p.OW = d.OW;
    p.y_cstride = d.y_cstride; p.y_coffset = d.y_coffset; p.y2_cstride = d.y2_cstride; p.y2_coffset = d.y2_coffset;
    p.flags = d.flags; p.lean_chunks = 0;
    p.M = d.N * d.OH * d.OW;
    p.K = d.kh * d.kw * d.Cin;
    p.tiles_m = p.tiles_n = p.tile_end = 0;
    p.kw_magic = (65536 + d.kw - 1) / d.kw;
    // m / OW == umulhi(m, ceil(2^32 / OW)) for every m with m * OW < 2^32 (error term < OW per 2^32)
    p.ow_magic = d.OW > 1 ? (unsigned)(((1ull << 32) + d.OW - 1) / d.OW) : 0u;
    p.oh_magic = d.OH > 1 ? (unsigned)(((1ull << 32) + d.OH - 1) / d.OH) : 0u;
    p.zero_page = zero_page;
    p.tiles_n_magic = 0;      // plan_tiles_cfg
    p.cin_magic24 = (unsigned)(((1u << 24) + d.Cin - 1) / d.Cin);
}

int plan_tiles_cfg(int cfg, ConvP* ps, int n);

// Heuristic used when the caller does not autotune (cfg_request = -1).  Fitted to tools/conv_sweep.py on
// MI355X: loads that miss a CU's L1 stream at ~12-16 B/clk into LDS whatever the ring depth, so a workgroup's chunk
// costs about max(MFMA cycles, staged bytes / 12) plus a barrier; workgroups run in rounds over 256 CUs.
int choose_cfg(const ConvP* ps, int n) {
    const char* force = getenv("FCN_CONV_CFG");
    if (force && force[0] >= '0' && force[0] <= '9' && atoi(force) < kNumTileCfg) return atoi(force);
    if (force && is_stream_cfg(atoi(force)) && (ps[0].flags & FCN_CONV_F16)) {      // (tests / sweeps: the streaming kernel where it applies)
        ConvP tmp[16];
        for (int i = 0; i < n && i < 16; ++i) tmp[i] = ps[i];
        if (n <= kMaxGroup && plan_tiles_cfg(atoi(force), tmp, n) > 0) return atoi(force);
    }
    if (n == 1 && (first7_ok(ps[0]) || first7_f16_ok(ps[0])) && !(getenv("FCN_CONV_FIRST7") && atoi(getenv("FCN_CONV_FIRST7")) == 0)) return kFirst7Cfg;
    int best = 0;
    double best_cost = 1e300;
    for (int c = 0; c < kNumTileCfg; ++c) {
        const double bm = kCfgs[c].bm, bn = kCfgs[c].bn, bk = kCfgs[c].bk;
        const double mf = bm * bn * bk / 128.0, ld = (bm + bn) * bk * 4.0 / 12.0;
        const double per_chunk = (mf > ld ? mf : ld) + 150.0 + (kCfgs[c].prefetch ? 0.0 : 200.0 * bk / 8.0);
        long long tiles = 0;
        double work = 0, longest = 0;
        for (int i = 0; i < n; ++i) {
            const long long t = (long long)cdiv(ps[i].M, kCfgs[c].bm) * cdiv(ps[i].Cout, kCfgs[c].bn);
            const int chunks = cdiv(ps[i].K, kCfgs[c].bk);
            tiles += t;
            work += (double)t * chunks;
            if (chunks > longest) longest = chunks;
        }
        const double rounds = (double)((tiles + 255) / 256);
        const double avg = work / (double)tiles;
        double cyc = tiles <= 256 ? longest * per_chunk : rounds * avg * per_chunk;
        if (cyc < longest * per_chunk) cyc = longest * per_chunk;
        const double cost = 9000.0 + cyc;
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

int plan_tiles_cfg(int cfg, ConvP* ps, int n) {
    if (cfg == kDot1x1Cfg) {
        if (!dot1x1_ok(ps, n)) return -2;
        for (int i = 0; i < n; ++i) {
            ps[i].tiles_m = cdiv(ps[0].M, kDotPx);
            ps[i].tiles_n = 1;
            ps[i].tiles_n_magic = 0;
            ps[i].lean_chunks = 0;
            ps[i].tile_end = ps[i].tiles_m;
        }
        return ps[0].tiles_m;
    }
    if (cfg == kFirst7Cfg) {
        if (n != 1 || !(first7_ok(ps[0]) || first7_f16_ok(ps[0]))) return -2;
        ps[0].tiles_m = ps[0].N * cdiv(ps[0].OH, kD7Th) * cdiv(ps[0].OW, kD7Tw);
        ps[0].tiles_n = 1;
        ps[0].tiles_n_magic = 0;
        ps[0].lean_chunks = 0;
        ps[0].tile_end = ps[0].tiles_m;
        return ps[0].tiles_m;
    }
    if (is_stream_cfg(cfg)) {
        // conv_stream_f16: half problems with half outputs, bias + ReLU only, channel counts / strides in whole 16-byte groups,
        // stride-1 "same" convolutions with 1x1 / 3x3 / 5x5 filters whose slab (the tile's 256 pixels in padded raster order,
        // plus the taps of a filter row) fits the configuration's slab buffer; 1x1 filters need three slab buffers
        const StreamCfgInfo sc = stream_cfg_info(cfg - kStreamCfg0);
        int total = 0;
        for (int i = 0; i < n; ++i) {
            ConvP& q = ps[i];
            const long long xb = (((long long)q.N * q.H * q.W - 1) * q.x_cstride + q.Cin) * 2, wb = (long long)q.Cout * q.K * 2;
            const long long yb = (((long long)q.M - 1) * q.y_cstride + q.y_coffset + q.Cout) * 2;
            const int k = q.kh, pw = q.W + 2 * q.pad;
            const int wraps = (sc.bm - 2 + q.W) / q.W;      // image-row ends a tile's pixels can cross
            const int rows = sc.bm - 1 + 2 * q.pad * wraps + 2 * q.pad + 1;
            if ((q.flags & ~FCN_CONV_RELU) != FCN_CONV_F16 || (q.Cout | q.y_cstride | q.y_coffset) % 8 != 0 || ((uintptr_t)q.y & 15) != 0 ||
                ((uintptr_t)q.bias & 15) != 0 || xb >= (1ll << 31) || wb >= (1ll << 31) || yb >= (1ll << 31) ||
                q.kh != q.kw || (k != 1 && k != 3 && k != 5) || q.stride != 1 || q.pad != (k - 1) / 2 || rows > sc.slab_rows ||
                (k == 1 && sc.slab_buffers < 3) || pw < 16 || (long long)q.N * q.H * pw * pw >= (1ll << 32))
                return -2;
            q.tiles_m = cdiv(q.M, sc.bm);
            q.tiles_n = cdiv(q.Cout, sc.bn);
            q.tiles_n_magic = q.tiles_n > 1 ? (unsigned)(((1ull << 32) + q.tiles_n - 1) / q.tiles_n) : 0u;
            total += q.tiles_m * q.tiles_n;
            q.tile_end = total;
            // Packed taps (conv_stream.hip): a 5x5 / 3x3 filter row over a dense 16- or 32-channel blob runs 4 / 2 taps per 64-wide
            // chunk instead of one tap that is 75 / 50 % padding - lean_chunks carries log2(taps per chunk) for this kernel.  Only
            // where a slab still serves two chunks or more (the single-chunk protocol is the 1x1 one: three slab buffers).
            static const bool pack_ok = !(getenv("FCN_STREAM_PACK") && atoi(getenv("FCN_STREAM_PACK")) == 0);
            int tsh = 0;
            if (pack_ok && k > 1 && q.x_cstride == q.Cin && ((uintptr_t)q.x & 15) == 0) tsh = q.Cin == 16 ? 2 : q.Cin == 32 ? 1 : 0;
            if (tsh && cdiv(k, 1 << tsh) < 2) tsh = 0;
            q.lean_chunks = tsh;
            q.cin_magic24 = (unsigned)(((1ull << 32) + pw - 1) / pw);      // (stream problems: entry / PW of the padded raster)
        }
        return total;
    }
    const int bm = kCfgs[cfg].bm, bn = kCfgs[cfg].bn;
    // longest tiles first (chunks of K per tile; the order of a group's problems is free - each writes its own output): what
    // the snake dealing of rounds in conv_fwd_group assumes
    if (n > 1) std::stable_sort(ps, ps + n, [&](const ConvP& a, const ConvP& b) { return cdiv(a.K, kCfgs[cfg].bk) > cdiv(b.K, kCfgs[cfg].bk); });
    static const bool lean_ok = !(getenv("FCN_CONV_LEAN") && atoi(getenv("FCN_CONV_LEAN")) == 0);      // (experiments: per-lane loader only)
    int total = 0;
    for (int i = 0; i < n; ++i) {
        ps[i].tiles_m = cdiv(ps[i].M, bm);
        ps[i].tiles_n = cdiv(ps[i].Cout, bn);
        const long long tl = (long long)ps[i].tiles_m * ps[i].tiles_n;
        if (tl * ps[i].tiles_n >= (1ll << 32)) return -1;      // (M < 2^31 and Cout * K < 2^31 keep this far away)
        ps[i].tiles_n_magic = ps[i].tiles_n > 1 ? (unsigned)(((1ull << 32) + ps[i].tiles_n - 1) / ps[i].tiles_n) : 0u;
        total += ps[i].tiles_m * ps[i].tiles_n;
        ps[i].tile_end = total;
        // Which loader feeds this problem under this configuration (conv_body): the scalar-addressed one needs chunks that
        // do not straddle filter taps - 1x1 filters, or every tap's Cin padded to whole chunks (the pad lanes load zeros and
        // cost MFMA work: accepted up to 1.5x) - and operands below 2 GiB (32-bit buffer offsets).
        const int esz = (ps[i].flags & FCN_CONV_F16) ? 2 : 4;
        const int bke = kCfgs[cfg].bk * 4 / esz;
        const int taps = ps[i].kh * ps[i].kw, cpt = cdiv(ps[i].Cin, bke);
        const long long xb = (((long long)ps[i].N * ps[i].H * ps[i].W - 1) * ps[i].x_cstride + ps[i].Cin) * esz;
        const long long wb = (long long)ps[i].Cout * ps[i].K * esz;
        const bool lean = lean_ok && xb < (1ll << 31) && wb < (1ll << 31) && taps <= 31 && (taps == 1 || 2ll * cpt * bke <= 3ll * ps[i].Cin);      // (one bit per tap)
        ps[i].lean_chunks = lean ? taps * cpt : 0;
        // The scalar-light form of that loader (conv_body; split-role shapes): a tap change costs it a few vector instructions and its
        // prologue one bit per tap, a chunk inside a tap nothing - it pays for 1x1 filters and for taps of several chunks (measured, cfg 23,
        // M = 784, N = 320, 3x3: one chunk per tap 4.9 -> 5.5 us, two 6.5 -> 7.2, five 11.3 -> 10.7, ten 19.3 -> 16.6;
        // profiles/experiments/r04_sweep_loader.txt).  $FCN_CONV_LIGHT: the least chunks per tap that take it (0: never).
        static const int light_min = getenv("FCN_CONV_LIGHT") ? atoi(getenv("FCN_CONV_LIGHT")) : 4;
        ps[i].kw_magic = (65536 + ps[i].kw - 1) / ps[i].kw;      // (what fill() set: this array may have been planned for another configuration before)
        // (larger split-role tiles stage two or more pieces per wave and operand: the old loader pays its per-piece vector instructions for
        //  each, and the scalar-light one wins at every chunk count - 64 x 32, conv2/3x3 with two chunks per tap: 37.1 -> 32.9 us)
        const bool big_tile = bm * bn > 32 * 32;
        if (lean && kCfgRoles[cfg] && light_min > 0 && (taps == 1 || cpt >= light_min || big_tile)) ps[i].kw_magic = -1;
    }
    return total;
}

// compute units of the current device (cached per device: asked at every launch)
int device_cus() {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int v = 0;
        cached[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cached[dev];
}

template <typename T>
void launch_one_cfg(int cfg, const ConvP& p, int total, hipStream_t st) {
    if (cfg == kFirst7Cfg) {
        if (p.flags & FCN_CONV_F16) {      // persistent: one workgroup per CU (a wave per SIMD holds its filters in registers), each walks its share of the tiles
            static const bool x4_ok = !(getenv("FCN_FIRST7_X4") && atoi(getenv("FCN_FIRST7_X4")) == 0);      // (experiments: the 8-half kernel)
            const bool small = (long long)p.N * p.H * p.W * 16 < (1ll << 31) &&      // (32-bit buffer offsets in the 4-half kernel)
                               (((long long)p.N * p.OH * p.OW - 1) * p.y_cstride + p.y_coffset + p.Cout) * 2 < (1ll << 31);
            if ((p.flags & FCN_CONV_IMAGE_ONES) && x4_ok && small) {      // channels 3 and 4 are the constant 1: three multiplied channels, two workgroups per CU
                const int grid = total < 2 * device_cus() ? total : 2 * device_cus();
                hipLaunchKernelGGL(conv_first7_f16x4_kernel, dim3(grid), dim3(kD7Threads), 0, st, p, total);
                return;
            }
            const int grid = total < device_cus() ? total : device_cus();      // persistent: one workgroup per compute unit
            hipLaunchKernelGGL(conv_first7_f16_kernel, dim3(grid), dim3(kD7Threads), 0, st, p, total);
        } else {
            hipLaunchKernelGGL(conv_first7_kernel, dim3(total), dim3(kD7Threads), 0, st, p);
        }
        return;
    }
    switch (cfg) {
#define X(I, A, B, C_, D, E, F, G, H)                                                                                                 \
    case I:                                                                                                                           \
        hipLaunchKernelGGL((conv_fwd_one<T, A, B, C_, D, E, F, G, H>), dim3(total), dim3(Cfg<A, B, C_, D, E, F, G, H>::NT), 0, st, p); \
        break;
        FCN_CONV_CONFIGS(X)
#undef X
    }
}

template <typename T>
void launch_group_cfg(int cfg, const GroupArgs& ga, int pool_wgs, int snake, int total, hipStream_t st) {
    switch (cfg) {
#define X(I, A, B, C_, D, E, F, G, H)                                                                                                     \
    case I:                                                                                                                               \
        hipLaunchKernelGGL((conv_fwd_group<T, A, B, C_, D, E, F, G, H>), dim3(total), dim3(Cfg<A, B, C_, D, E, F, G, H>::NT), 0, st, ga.nprob,      \
                           ga.tile_end[0], ga.tile_end[1], ga.tile_end[2], ga.tile_end[3], ga.tile_end[4], ga.tile_end[5], ga.tile_end[6],          \
                           ga.tile_end[7], pool_wgs, snake, ga);                                                                                   \
        break;
        FCN_CONV_CONFIGS(X)
#undef X
    }
}

// the tail variant: float32 split-role shapes with 32-channel tiles (conv_body's static_assert)
constexpr bool tail_cfg_ok(int cfg) { return cfg == 23 || cfg == 24 || cfg == 25 || cfg == 27; }
void launch_group_tail(int cfg, const GroupArgsTail& ga, int pool_wgs, int snake, int total, hipStream_t st) {
    const GroupArgs& g = ga.g;
    switch (cfg) {
#define XT(I, A, B, C_, D, E, F, G, H)                                                                                                         \
    case I:                                                                                                                                    \
        hipLaunchKernelGGL((conv_fwd_group<float, A, B, C_, D, E, F, G, H, true>), dim3(total), dim3(Cfg<A, B, C_, D, E, F, G, H>::NT), 0, st, g.nprob, \
                           g.tile_end[0], g.tile_end[1], g.tile_end[2], g.tile_end[3], g.tile_end[4], g.tile_end[5], g.tile_end[6],             \
                           g.tile_end[7], pool_wgs, snake, ga);                                                                                \
        break;
        XT(23, 1, 1, 1, 1, 4, 32, 16 + 4, true)
        XT(24, 1, 1, 1, 1, 4, 64, 16 + 4, true)
        XT(25, 1, 1, 2, 1, 2, 32, 16 + 4, true)
        XT(27, 1, 1, 1, 1, 4, 32, 16 + 6, true)
#undef XT
    }
}

// host copies of prepared groups, keyed by their device workspace (the launch needs the problems by value)
struct HostTail { bool on; int finalize; int n; fcn_conv_desc heads[kTailMaxSlices]; float* scratch; unsigned* arrive; TailP t; };
struct HostGroup { int n; ConvP ps[16]; int npool; PoolP pools[kMaxPool]; HostTail tail; };
std::unordered_map<const void*, HostTail> g_tails;      // tails announced for a workspace (fcn_conv2d_group_attach_tail), taken by the next prepare()
std::mutex g_groups_mu;
std::unordered_map<const void*, HostGroup> g_groups;

}  // namespace

extern "C" {

#ifdef FCN_CONV_STAMPS
// diagnostic build: d_buf holds cap * 16 u64 (per workgroup: 7 x {100 MHz clock, shader clock}, XCC id, HW id)
int fcn_debug_conv_stamps(void* d_buf, int cap) {
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &d_buf, sizeof(d_buf)));
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps_cap), &cap, sizeof(cap)));
    return 0;
}
#endif

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
    FCN_REQUIRE(total > 0, FCN_E_UNSUPPORTED, "conv: too many tiles for the 32-bit tile decode");
    if (h_desc->flags & FCN_CONV_F16) launch_one_cfg<f16_t>(cfg, p, total, as_stream(s));
    else launch_one_cfg<float>(cfg, p, total, as_stream(s));
    FCN_LAUNCH_CHECK("conv_fwd_one");
    return 0;
}

size_t fcn_conv2d_group_workspace_bytes(int n) { return sizeof(ConvP) * (size_t)(n > 0 ? n : 0); }

int fcn_conv2d_num_configs(void) { return num_cfgs(); }

int fcn_conv2d_first_layer_config(void) { return kFirst7Cfg; }

int fcn_conv2d_config_lds_bytes(int cfg) {
    if (is_stream_cfg(cfg)) return stream_cfg_info(cfg - kStreamCfg0).lds_bytes;
    return cfg == kFirst7Cfg ? kD7LdsBytes :      /* (the half-float variant holds 84 KB) */ cfg == kDot1x1Cfg ? kDotMaxSlices * kDotRedPx * kDotOut * kDotRedPitch * 4 : cfg >= 0 && cfg < kNumTileCfg ? kCfgLdsBytes[cfg] : -1;
}

int fcn_conv2d_config_waves_k(int cfg) { return cfg == kFirst7Cfg || cfg == kDot1x1Cfg || is_stream_cfg(cfg) ? 1 : cfg >= 0 && cfg < kNumTileCfg ? kCfgWavesK[cfg] : -1; }

int fcn_conv2d_group_prepare(const fcn_conv_desc* h_descs, int n, void* d_workspace, int cfg_request, fcn_conv_group* h_out) {
    return fcn_conv2d_group_prepare_fused(h_descs, n, nullptr, 0, d_workspace, cfg_request, h_out);
}

int fcn_conv2d_group_prepare_fused(const fcn_conv_desc* h_descs, int n, const fcn_pool_desc* h_pools, int npools, void* d_workspace,
                                   int cfg_request, fcn_conv_group* h_out) {
    FCN_REQUIRE(h_descs && h_out && d_workspace && n > 0 && n <= 16, FCN_E_ARG, "fcn_conv2d_group_prepare: need 1..16 problems, workspace, out");
    FCN_REQUIRE(cfg_request >= -1 && cfg_request < num_cfgs(), FCN_E_ARG, "fcn_conv2d_group_prepare: tile configuration %d out of range", cfg_request);
    FCN_REQUIRE(npools >= 0 && npools <= kMaxPool && (npools == 0 || (h_pools && n <= kMaxGroup)), FCN_E_ARG,
                "fcn_conv2d_group_prepare_fused: at most %d poolings, and only beside at most %d convolutions", kMaxPool, kMaxGroup);
    ConvP ps[16];
    int zrc = 0;
    const float* zp = zero_page_for_current_device(&zrc);
    if (zrc) return zrc;
    for (int i = 0; i < n; ++i) {
        int rc = validate(h_descs[i]);
        if (rc) return rc;
        FCN_REQUIRE(((h_descs[i].flags ^ h_descs[0].flags) & FCN_CONV_F16) == 0, FCN_E_ARG, "conv group: f32 and f16 problems cannot share a launch");
        fill(ps[i], h_descs[i], zp);
    }
    const int group_f16 = (h_descs[0].flags & FCN_CONV_F16) ? 1 : 0;
    PoolP pools[kMaxPool] = {};
    for (int i = 0; i < npools; ++i) {
        const fcn_pool_desc& d = h_pools[i];
        FCN_REQUIRE(d.x && d.y && d.N > 0 && d.H > 0 && d.W > 0 && d.C > 0 && d.k > 0 && d.stride > 0 && d.pad >= 0 && d.pad < d.k && d.OH > 0 &&
                        d.OW > 0, FCN_E_ARG, "fused maxpool: bad args");
        FCN_REQUIRE((d.OH - 1) * d.stride - d.pad < d.H && (d.OW - 1) * d.stride - d.pad < d.W, FCN_E_ARG,
                    "fused maxpool: last window starts outside the image");
        const int peps = group_f16 ? 8 : 4;
        FCN_REQUIRE((d.f16 ? 1 : 0) == group_f16, FCN_E_ARG, "fused maxpool: element type differs from the group's convolutions");
        FCN_REQUIRE(d.C % peps == 0 && d.x_cstride % peps == 0 && d.y_cstride % peps == 0 && d.y_coffset % peps == 0 && d.x_cstride >= d.C &&
                        d.y_coffset >= 0 && d.y_cstride >= d.y_coffset + d.C, FCN_E_ALIGN, "fused maxpool: channels / strides must be multiples of %d", peps);
        FCN_REQUIRE((((uintptr_t)d.x | (uintptr_t)d.y | (uintptr_t)d.idx) & 15) == 0, FCN_E_ALIGN, "fused maxpool: pointers must be 16-byte aligned");
        FCN_REQUIRE((long long)d.N * d.OH * d.OW * (d.C / peps) < (1ll << 30), FCN_E_UNSUPPORTED, "fused maxpool: too large");
        PoolP& q = pools[i];
        q.x = d.x; q.y = d.y; q.idx = d.idx;
        q.N = d.N; q.H = d.H; q.W = d.W; q.C = d.C; q.x_cstride = d.x_cstride; q.k = d.k; q.stride = d.stride; q.pad = d.pad;
        q.OH = d.OH; q.OW = d.OW; q.y_cstride = d.y_cstride; q.y_coffset = d.y_coffset;
        q.items = d.N * d.OH * d.OW * (d.C / peps);
        q.wg_end = 0;      // depends on the workgroup size of the tile configuration: set at launch
    }
    HostTail tail = {};
    {
        std::lock_guard<std::mutex> lock(g_groups_mu);
        auto it = g_tails.find(d_workspace);
        if (it != g_tails.end()) tail = it->second;
    }
    if (tail.on) {
        // which problems write the narrow problems' input?  Each of them contributes partial sums (conv_common.h, TailP)
        const fcn_conv_desc& h0 = tail.heads[0];
        int contributors = 0;
        FCN_REQUIRE(n <= kMaxGroup && !group_f16, FCN_E_UNSUPPORTED, "conv tail: at most %d float32 problems in the launch", kMaxGroup);
        for (int i = 0; i < n; ++i) {
            const ConvP& q = ps[i];
            if ((const float*)q.y != h0.x) continue;
            FCN_REQUIRE(q.y_cstride == h0.x_cstride && q.M == h0.N * h0.H * h0.W && q.Cout % 32 == 0 && q.y_coffset % 32 == 0 && q.y_coffset + q.Cout <= h0.Cin &&
                            (q.flags & ~FCN_CONV_RELU) == 0 && ((uintptr_t)q.y & 15) == 0 && q.y_cstride % 4 == 0 && (!q.bias || ((uintptr_t)q.bias & 15) == 0),
                        FCN_E_UNSUPPORTED, "conv tail: a producer of the narrow problems' input must be a float32 bias + ReLU convolution over the same pixels "
                                           "that writes whole 32-channel groups of it");
            ps[i].flags |= FCN_CONV_TAILF;
            ++contributors;
        }
        FCN_REQUIRE(contributors > 0, FCN_E_ARG, "conv tail: no problem of this group writes the narrow problems' input");
    }
    int cfg = cfg_request >= 0 ? cfg_request : choose_cfg(ps, n);
    if (tail.on && cfg_request < 0 && !tail_cfg_ok(cfg)) cfg = 23;
    FCN_REQUIRE(!tail.on || tail_cfg_ok(cfg), FCN_E_UNSUPPORTED, "conv tail: configuration %d has no tail variant (23, 24, 25, 27 do)", cfg);
    const int total = plan_tiles_cfg(cfg, ps, n);
    if (tail.on && total > 0) {
        TailP& t = tail.t;
        t = TailP{};
        t.h.scratch = tail.scratch;
        t.h.arrive = tail.arrive;
        t.h.K = tail.heads[0].Cin;
        t.h.M = tail.heads[0].N * tail.heads[0].H * tail.heads[0].W;
        t.h.nslices = tail.n;
        int rows = 0;
        for (int i = 0; i < tail.n; ++i) {
            const fcn_conv_desc& h = tail.heads[i];
            TailSlice& sl = t.s[i];
            sl.bias = h.bias; sl.y = h.y; sl.y2 = h.y2;
            sl.o0 = rows; sl.nout = h.Cout; sl.y_cstride = h.y_cstride; sl.y_coffset = h.y_coffset; sl.y2_cstride = h.y2_cstride; sl.y2_coffset = h.y2_coffset;
            sl.flags = h.flags;
            for (int r = 0; r < h.Cout; r += 4) t.h.gw[(rows + r) / 4] = h.w + (size_t)r * h.Cin;
            rows += h.Cout;
        }
        for (int j = rows / 4; j < kTailRows / 4; ++j) t.h.gw[j] = t.h.gw[0];
        t.h.rows = rows;
        t.h.dbg = getenv("FCN_TAIL_DEBUG") ? atoi(getenv("FCN_TAIL_DEBUG")) : 0;      // (elimination switches: 1 no staging, 2 no partial sums, 4 no partial stores)
        t.h.need = 0;
        if (tail.finalize)
            for (int i = 0; i < n; ++i)
                if (ps[i].flags & FCN_CONV_TAILF) t.h.need += ps[i].tiles_n;
    }
    FCN_REQUIRE(!(is_stream_cfg(cfg) && (total == -2 || npools || n > kMaxGroup)), FCN_E_UNSUPPORTED,
                "conv group: configuration %d (persistent half-float streaming kernel) takes at most %d half problems with half outputs, "
                "bias + ReLU only, Cout / y_cstride / y_coffset multiples of 8, stride-1 1x1 / 3x3 / 5x5 filters with pad (k - 1) / 2 on images wide "
                "enough for its slab buffer (1x1 filters: the configurations with three slab buffers), and no poolings", cfg, kMaxGroup);
    FCN_REQUIRE(total != -2 && !((cfg == kFirst7Cfg || cfg == kDot1x1Cfg) && npools), FCN_E_UNSUPPORTED,
                "conv group: configuration %d is shape-specific (%d: one 7x7 / stride 2 / pad 3 problem on 4-channel pixels, 33..64 outputs; "
                "%d: 1x1 / stride 1 float32 problems over the same pixels, at most 32 output channels in slices of 8) and takes no poolings",
                cfg, kFirst7Cfg, kDot1x1Cfg);
    FCN_REQUIRE(total > 0, FCN_E_UNSUPPORTED, "conv group: too many tiles for the 32-bit tile decode");
    FCN_HIP(hipMemcpy(d_workspace, ps, sizeof(ConvP) * n, hipMemcpyHostToDevice));
    {
        std::lock_guard<std::mutex> lock(g_groups_mu);
        HostGroup& hg = g_groups[d_workspace];
        hg.n = n;
        for (int i = 0; i < n; ++i) hg.ps[i] = ps[i];
        hg.npool = npools;
        for (int i = 0; i < kMaxPool; ++i) hg.pools[i] = pools[i];
        hg.tail = tail;
    }
    h_out->d_probs = d_workspace;
    h_out->n = n;
    h_out->cfg = cfg;
    h_out->total_tiles = total;
    return 0;
}

int fcn_conv2d_group_release(void* d_workspace) {
    std::lock_guard<std::mutex> lock(g_groups_mu);
    g_groups.erase(d_workspace);
    g_tails.erase(d_workspace);
    return 0;
}

namespace {
int tail_check(const fcn_conv_tail* t) {
    FCN_REQUIRE(t && t->n >= 1 && t->n <= kTailMaxSlices, FCN_E_ARG, "conv tail: 1..%d narrow problems", kTailMaxSlices);
    int rows = 0;
    const fcn_conv_desc& h0 = t->heads[0];
    for (int i = 0; i < t->n; ++i) {
        const fcn_conv_desc& h = t->heads[i];
        int rc = validate(h);
        if (rc) return rc;
        FCN_REQUIRE(h.kh == 1 && h.kw == 1 && h.stride == 1 && h.pad == 0 && (h.flags & ~(FCN_CONV_RELU | FCN_CONV_SIGMOID2)) == 0, FCN_E_UNSUPPORTED,
                    "conv tail: the narrow problems are float32 1x1 / stride 1 convolutions (bias, ReLU, sigmoid second output)");
        FCN_REQUIRE(h.x == h0.x && h.x_cstride == h0.x_cstride && h.Cin == h0.Cin && h.N == h0.N && h.H == h0.H && h.W == h0.W, FCN_E_ARG,
                    "conv tail: the narrow problems read the same blob");
        FCN_REQUIRE(h.Cout % 4 == 0 && h.Cin % 32 == 0 && h.Cin <= 1024 && h.y_cstride % 4 == 0 && h.y_coffset % 4 == 0 && ((uintptr_t)h.y & 15) == 0 &&
                        (!h.bias || ((uintptr_t)h.bias & 15) == 0) && (!h.y2 || (((uintptr_t)h.y2 & 15) == 0 && h.y2_cstride % 4 == 0 && h.y2_coffset % 4 == 0)),
                    FCN_E_UNSUPPORTED, "conv tail: outputs in whole, 16-byte aligned groups of four channels; input channels in whole groups of 32, at most 1024");
        rows += h.Cout;
    }
    FCN_REQUIRE(rows <= kTailRows, FCN_E_UNSUPPORTED, "conv tail: at most %d output channels in all", kTailRows);
    return 0;
}
}  // namespace

size_t fcn_conv2d_tail_scratch_bytes(const fcn_conv_tail* t) {
    if (tail_check(t)) return 0;
    size_t rows = 0;
    for (int i = 0; i < t->n; ++i) rows += (size_t)t->heads[i].Cout;
    return (size_t)(t->heads[0].Cin / 32) * (size_t)t->heads[0].N * t->heads[0].H * t->heads[0].W * rows * sizeof(float);
}

size_t fcn_conv2d_tail_arrive_bytes(const fcn_conv_tail* t) {
    if (tail_check(t)) return 0;
    return ((size_t)t->heads[0].N * t->heads[0].H * t->heads[0].W / 32 + 1) * sizeof(unsigned);
}

int fcn_conv2d_group_attach_tail(void* d_workspace, const fcn_conv_tail* t) {
    FCN_REQUIRE(d_workspace, FCN_E_ARG, "fcn_conv2d_group_attach_tail: null workspace");
    if (!t) {
        std::lock_guard<std::mutex> lock(g_groups_mu);
        g_tails.erase(d_workspace);
        return 0;
    }
    int rc = tail_check(t);
    if (rc) return rc;
    FCN_REQUIRE(t->scratch && t->arrive && ((uintptr_t)t->scratch & 15) == 0, FCN_E_ARG, "fcn_conv2d_group_attach_tail: scratch (16-byte aligned) and arrival words");
    HostTail ht = {};
    ht.on = true;
    ht.finalize = t->finalize ? 1 : 0;
    ht.n = t->n;
    for (int i = 0; i < t->n; ++i) ht.heads[i] = t->heads[i];
    ht.scratch = t->scratch;
    ht.arrive = reinterpret_cast<unsigned*>(t->arrive);
    std::lock_guard<std::mutex> lock(g_groups_mu);
    g_tails[d_workspace] = ht;
    return 0;
}

int fcn_conv2d_fwd_group_f32(const fcn_conv_group* g, fcn_stream_t s) {
    FCN_REQUIRE(g && g->d_probs && g->n > 0 && g->total_tiles > 0, FCN_E_ARG, "fcn_conv2d_fwd_group_f32: unprepared group");
    FCN_REQUIRE(g->cfg >= 0 && g->cfg < num_cfgs(), FCN_E_ARG, "fcn_conv2d_fwd_group_f32: bad cfg %d", g->cfg);
    HostGroup hg;
    {
        std::lock_guard<std::mutex> lock(g_groups_mu);
        auto it = g_groups.find(g->d_probs);
        FCN_REQUIRE(it != g_groups.end() && it->second.n == g->n, FCN_E_STATE, "fcn_conv2d_fwd_group_f32: group was not prepared by this library instance");
        hg = it->second;
    }
    if (g->cfg == kDot1x1Cfg) {
        FCN_REQUIRE(hg.npool == 0 && dot1x1_ok(hg.ps, hg.n), FCN_E_STATE, "fcn_conv2d_fwd_group_f32: group was not prepared for the lane-split 1x1 kernel");
        DotArgs da = {};
        da.M = hg.ps[0].M;
        for (int i = 0; i < hg.n; ++i)
            for (int n0 = 0; n0 < hg.ps[i].Cout; n0 += kDotOut) {
                const ConvP& q = hg.ps[i];
                DotSlice& d = da.s[da.nslices++];
                d.x = q.x; d.w = q.w + (size_t)n0 * q.K; d.bias = q.bias ? q.bias + n0 : nullptr; d.y = q.y; d.y2 = q.y2;
                d.K = q.K; d.x_cstride = q.x_cstride; d.y_cstride = q.y_cstride; d.y_coffset = q.y_coffset + n0;
                d.y2_cstride = q.y2_cstride; d.y2_coffset = q.y2_coffset + n0;
                d.nout = q.Cout - n0 < kDotOut ? q.Cout - n0 : kDotOut;
                d.flags = q.flags;
            }
        hipLaunchKernelGGL(conv_dot1x1_kernel, dim3(g->total_tiles), dim3(64 * kDotMaxSlices), 0, as_stream(s), da);
        FCN_LAUNCH_CHECK("conv_dot1x1");
        return 0;
    }
    if (g->cfg == kFirst7Cfg) {
        FCN_REQUIRE(hg.n == 1 && hg.npool == 0 && (first7_ok(hg.ps[0]) || first7_f16_ok(hg.ps[0])), FCN_E_STATE,
                    "fcn_conv2d_fwd_group_f32: group was not prepared for the first-layer kernel");
        launch_one_cfg<float>(kFirst7Cfg, hg.ps[0], g->total_tiles, as_stream(s));
        FCN_LAUNCH_CHECK("conv_first7");
        return 0;
    }
    if (is_stream_cfg(g->cfg)) {
        FCN_REQUIRE(hg.npool == 0 && hg.n <= kMaxGroup && (hg.ps[0].flags & FCN_CONV_F16), FCN_E_STATE, "fcn_conv2d_fwd_group_f32: group was not prepared for the streaming kernel");
        GroupArgs ga;
        ga.npool = 0;
        for (int i = 0; i < kMaxPool; ++i) ga.pool[i] = PoolP{};
        ga.nprob = hg.n;
        for (int i = 0; i < kMaxGroup; ++i) {
            const ConvP& src = hg.ps[i < hg.n ? i : hg.n - 1];
            ga.p[i] = src;
            ga.tile_end[i] = src.tile_end;
        }
        launch_stream(g->cfg - kStreamCfg0, ga, g->total_tiles, as_stream(s));
        FCN_LAUNCH_CHECK("conv_stream_f16");
        return 0;
    }
    // at most kMaxGroup problems ride in one launch's kernel arguments; larger groups take several launches
    for (int first = 0; first < hg.n; first += kMaxGroup) {
        GroupArgs ga;
        ga.npool = 0;
        for (int i = 0; i < kMaxPool; ++i) ga.pool[i] = PoolP{};
        ga.nprob = hg.n - first < kMaxGroup ? hg.n - first : kMaxGroup;
        const int base = first ? hg.ps[first - 1].tile_end : 0;
        for (int i = 0; i < kMaxGroup; ++i) {
            const ConvP& src = hg.ps[first + (i < ga.nprob ? i : ga.nprob - 1)];
            ga.p[i] = src;
            ga.tile_end[i] = src.tile_end - base;
        }
        int grid = ga.tile_end[ga.nprob - 1];
        int pool_wgs = 0;
        if (hg.npool > 0) {      // only with n <= kMaxGroup: a single launch
            int end = 0;
            ga.npool = hg.npool;
            for (int i = 0; i < kMaxPool; ++i) {
                ga.pool[i] = hg.pools[i < hg.npool ? i : hg.npool - 1];
                const int per_wg = kCfgThreads[g->cfg] * (ga.pool[i].items <= kPoolSmallItems ? 1 : kPoolItemsPerThread);
                if (i < hg.npool) end += (ga.pool[i].items + per_wg - 1) / per_wg;
                ga.pool[i].wg_end = end;
            }
            grid += end;
            static const bool pool_last = getenv("FCN_POOL_LAST") && atoi(getenv("FCN_POOL_LAST")) == 1;
            pool_wgs = pool_last ? -end : end;
        }
        // snake dealing of the rounds (conv_fwd_group): mixed tile lengths, at least three rounds, a power-of-two CU count
        int snake = 0;
        {
            static const bool snake_ok = !(getenv("FCN_CONV_SNAKE") && atoi(getenv("FCN_CONV_SNAKE")) == 0);
            const int cus = device_cus();
            if (snake_ok && ga.nprob > 1 && (cus & (cus - 1)) == 0 && grid > 2 * cus && pool_wgs >= 0 && pool_wgs < cus) {
                int sh = 0;
                while ((1 << sh) < cus) ++sh;
                snake = ((grid + cus - 1) / cus) << 8 | sh;
            } else if (getenv("FCN_CONV_XCD") && atoi(getenv("FCN_CONV_XCD")) == 1 && pool_wgs == 0 && first == 0 && hg.n <= kMaxGroup) {
                snake = (int)0x80000000;
            } else if (snake_ok && ga.nprob > 1 && pool_wgs == 0 && grid > cus && grid <= 2 * cus) {
                snake = -(grid - cus);      // two rounds: rotate the tile order by the overhang (conv_fwd_group)
            }
        }
        if (hg.tail.on) {      // (n <= kMaxGroup: the only launch)
            FCN_REQUIRE(tail_cfg_ok(g->cfg) && hg.n <= kMaxGroup, FCN_E_STATE, "fcn_conv2d_fwd_group_f32: group was not prepared with its tail");
            GroupArgsTail gt;
            gt.g = ga;
            gt.tail = hg.tail.t;
            launch_group_tail(g->cfg, gt, pool_wgs, snake, grid, as_stream(s));
        } else if (hg.ps[0].flags & FCN_CONV_F16) launch_group_cfg<f16_t>(g->cfg, ga, pool_wgs, snake, grid, as_stream(s));
        else launch_group_cfg<float>(g->cfg, ga, pool_wgs, snake, grid, as_stream(s));
        FCN_LAUNCH_CHECK("conv_fwd_group");
    }
    return 0;
}

}  // extern "C"
