// Backward pass, losses and solver updates for gfx950.
//
// Stands in for the part of the reference's hot path that runs inside `caffe train`
// (reference: train/train.sh:25-28; program = models/train_val.prototxt:53-72,2237-2281 and the
// solver settings of train/*/solver.prototxt): Net::Backward of every layer type of the DetectNet
// training net, the NVIDIA-Caffe L1Loss / EuclideanLoss layers, Dropout in TRAIN phase, and
// SGDSolver / AdamSolver::ComputeUpdateValue with L2 regularisation and per-blob lr_mult / decay_mult.
//
//   * weight gradient  dW[n][k] = sum_m dY[m][n] * A[m][k]   (k = (r*kw+q)*Cin + c, same order as the weights):
//     an MFMA GEMM whose REDUCTION runs over the pixels m.  Both operands are staged pixel-major with LDS-DMA
//     (dY rows and im2col rows are channel-contiguous in NHWC) and read with ds_read_b32 — lane (i, p) of the
//     32x32x2 MFMA reads element i of pixel row p, i.e. a transposed-operand read that is conflict-free because
//     32 consecutive lanes walk 32 consecutive floats.  Pixels are split over `splits` workgroups per tile; the
//     partial slabs are summed in a fixed order (bit-reproducible, no float atomics).  The bias gradient (column
//     sums of dY) rides along in the workgroups of the first k-tile.
//   * data gradient = the forward convolution kernel run on dY with the flipped / transposed filter bank
//     (fcn_conv_weights_flip_f32), accumulating into the bottom's gradient when the blob fans out.
//   * everything else is HBM-bound NHWC pointwise work, 16 bytes per lane where the shapes allow.
#include <math.h>

#include <type_traits>

#include "common.h"

using namespace fcn;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

namespace {

typedef const void __attribute__((address_space(1))) * gvoid_cptr;
typedef void __attribute__((address_space(3))) * lds_ptr;
typedef float __attribute__((address_space(1))) * gf_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// ---------------------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------------------
struct WgradP {
    const float* x;      // NHWC input of the layer
    const float* dy;     // NHWC gradient of the layer's output (view: dy_cstride, dy_coffset folded into the pointer)
    float* dw_part;      // partial slabs: split s holds [Cout][K] weights followed by [Cout] bias sums, slab_floats apart
    float* db_part;      // dw_part + Cout*K (bias part of slab 0) or nullptr
    const float* zero_page;
    int N, H, W, Cin, x_cstride;
    int Cout, kh, kw, pad, stride, OH, OW;
    int dy_cstride;
    int M, K;
    int tiles_n, tiles_k, splits, chunks_per_split;   // chunk = 32 pixels
    int slab_floats;                                   // Cout*K + Cout
    unsigned ow_magic, oh_magic;                       // ceil(2^32 / OW), ceil(2^32 / OH) for the pixel decode (0: the extent is 1)
    int kw_magic;
    int rn, rk;                                        // role-split kernel: the workgroup's region in 32-wide sub-tiles (channels, k)
    // role-split kernel: region classes 2 a + b (a / b: the last region along channels / k, usually a partial one).  The class with the
    // most MFMAs per step splits the pixels into runs of cps0 chunks; class c takes cls_d[c] runs per workgroup (cls_splits[c] splits),
    // so that every workgroup of the launch lasts about as long.  cls_dmagic = ceil(2^16 / cls_d), cls_count = regions in the class.
    int cls_splits[4], cls_d[4], cls_dmagic[4], cls_count[4], cps0, runs, wgs;
};

constexpr int WG_BP = 32;      // pixels per chunk

// Tile shapes of the dW GEMM.  A workgroup owns BN output channels x BK weight k-indices with WAVES_N x WAVES_K
// waves, each wave TN x TK 32x32 MFMA tiles.  Measured on MI355X (tools/wgrad_sweep.py): what pays is the number of
// workgroups RESIDENT per CU (each wave has one instruction stream, so loads, LDS reads and MFMAs of one workgroup
// serialise and only another workgroup's waves fill the matrix core meanwhile) - a 3-slot ring (48-72 KiB of LDS)
// beats both deeper rings and larger tiles that leave one workgroup per CU.
template <int TN, int TK, int WAVES_N, int WAVES_K, int NBUF>
struct WgCfg {
    static constexpr int BN = 32 * TN * WAVES_N, BK = 32 * TK * WAVES_K;
    static constexpr int NW = WAVES_N * WAVES_K, NT = 64 * NW;
    static constexpr int RN = 256 / BN, RK = 256 / BK;          // pixel rows one 1 KiB LDS-DMA instruction covers
    static constexpr int IN_ = WG_BP / RN / NW, IK = WG_BP / RK / NW;   // instructions per wave per chunk (dY rows, im2col rows)
    static constexpr int INST = IN_ + IK;
    static constexpr int BUF_FLOATS = WG_BP * (BN + BK);
    static_assert(BN <= 256 && BK <= 256 && (WG_BP / RN) % NW == 0 && (WG_BP / RK) % NW == 0, "rows must split into whole wave-instructions");
    static_assert(NBUF >= 3 && NBUF * BUF_FLOATS * 4 <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
    static_assert(INST * (NBUF - 2) <= 63, "vmcnt is a 6-bit counter");
};

template <int TN, int TK, int WAVES_N, int WAVES_K, int NBUF>
__device__ __forceinline__ void wgrad_body(const WgradP& p, int block, float* smem) {
    constexpr int WG_NBUF = NBUF;
    using C = WgCfg<TN, TK, WAVES_N, WAVES_K, NBUF>;
    constexpr int BN = C::BN, BK = C::BK, NW = C::NW, RN = C::RN, RK = C::RK, IN_ = C::IN_, IK = C::IK, INST = C::INST;
    constexpr int BUF_FLOATS = C::BUF_FLOATS;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wid / WAVES_K, wkk = wid % WAVES_K;

    int b = block;
    const int split = b % p.splits;
    b /= p.splits;
    const int tile_k = b % p.tiles_k;
    const int tile_n = b / p.tiles_k;
    const int n0 = tile_n * BN;
    const int k0 = tile_k * BK;

    // ---- loader: one LDS-DMA instruction = 1 KiB = RN pixel rows of BN floats (dY) or RK rows of BK floats (im2col);
    //      lane -> (row = lane / (B/4), 16-byte slot = lane % (B/4)).  Source column / k index are fixed per lane.
    const int n_row = lane / (BN / 4), n_slot = lane % (BN / 4);
    const int k_row = lane / (BK / 4), k_slot = lane % (BK / 4);
    const int dy_col = n0 + n_slot * 4;
    const bool dy_col_ok = dy_col < p.Cout;          // Cout is padded to 4 in the gradient buffer (cstride), columns past it read zeros
    const int kidx = k0 + k_slot * 4;
    const int ktap = kidx / p.Cin;
    const int kch = kidx - ktap * p.Cin;
    const int kr = (ktap * p.kw_magic) >> 16;
    const int kq = ktap - kr * p.kw;
    const bool k_ok = kidx < p.K;

    const int chunk0 = split * p.chunks_per_split;
    const int total_chunks = (p.M + WG_BP - 1) / WG_BP;
    int nchunks = total_chunks - chunk0;
    if (nchunks > p.chunks_per_split) nchunks = p.chunks_per_split;
    if (nchunks < 0) nchunks = 0;
    const int chunk_end = chunk0 + nchunks;


    // Both operands are fetched with `buffer_load ... lds`: 32-bit offsets, and lanes that must read zeros (rows past M,
    // columns past Cout, taps outside the image, chunks past this split's range) simply carry an out-of-range offset - the
    // hardware writes zeros for them (tools/probes/bufload_lds_probe.hip: the scalar offset is part of the range check).
    // That removes the zero page, the 64-bit address arithmetic and two selects from every piece: v_mfma_f32_32x32x2_f32
    // shares the vector ALU with this arithmetic, so it does not hide behind the MFMAs, it adds to them.
    constexpr int OOB = (int)0x80000000u;      // >= num_records (wgrad_validate: both operands stay below 2 GiB)
#if defined(__HIP_DEVICE_COMPILE__)      // (hipcc's host pass instantiates this body too and has no buffer-resource type: the loader is device-pass text only)
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.dy), 0, (int)((((long long)p.M - 1) * p.dy_cstride + (p.Cout + 3) / 4 * 4) * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * 4), 0x00020000);
    int n_vo[IN_];      // dY: byte offset of (row of chunk 0, column) - the chunk's row offset travels in the scalar offset
#pragma unroll
    for (int i = 0; i < IN_; ++i) n_vo[i] = dy_col_ok ? ((RN * (NW * i + wid) + n_row) * p.dy_cstride + dy_col) * 4 : OOB;
    int issue_chunk_idx = chunk0;
    int s_dy = 0;       // scalar offset of the chunk's first row in dY (out of range for chunks past this split's end)
    float* is_dst = smem;
    auto issue_pre = [&](const int buf) {
        is_dst = smem + buf * BUF_FLOATS;
        s_dy = issue_chunk_idx < chunk_end ? issue_chunk_idx * WG_BP * p.dy_cstride * 4 : 0x7F000000;      // rows >= M are past num_records by themselves
    };
    auto issue_n = [&](const int i) {          // dY[m][n0 + 4*slot ..]
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr)(is_dst + RN * (NW * i + wid) * BN), 16, n_vo[i], s_dy, 0, 0);
    };
    auto issue_k = [&](const int i) {          // im2col row m, k segment
        const int row = RK * (NW * i + wid) + k_row;
        const int m = issue_chunk_idx * WG_BP + row;
        const bool m_ok = (int)(m < p.M) & (int)(issue_chunk_idx < chunk_end);
        const int mm = m_ok ? m : 0;
        // m -> (img, oy, ox) by OW, then by OH: umulhi(m, ceil(2^32 / d)) is exact only while m * d < 2^32 (wgrad_validate).  Round 2
        // divided by OH * OW in one step: at 448x448 and batch >= 2 conv1 has m * OH * OW > 2^32, the last ~5 pixels of every image
        // decoded to the NEXT image and conv1's dW was off by 8e-4 (found by tests/test_gpu_fullsize.py's full-resolution backward).
        const int t = p.ow_magic ? (int)__umulhi((unsigned)mm, p.ow_magic) : mm;
        const int ox = mm - t * p.OW;
        const int img = p.oh_magic ? (int)__umulhi((unsigned)t, p.oh_magic) : t;
        const int oy = t - img * p.OH;
        const int iy = oy * p.stride - p.pad + kr;
        const int ix = ox * p.stride - p.pad + kq;
        const bool ok = (int)m_ok & (int)k_ok & (int)((unsigned)iy < (unsigned)p.H) & (int)((unsigned)ix < (unsigned)p.W);
        int vo = ok ? (((img * p.H + iy) * p.W + ix) * p.x_cstride + kch) * 4 : OOB;
        asm volatile("" : "+v"(vo));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxx, (lds_ptr)(is_dst + WG_BP * BN + RK * (NW * i + wid) * BK), 16, vo, 0, 0, 0);
    };
#else
    int issue_chunk_idx = chunk0;
    auto issue_pre = [&](const int) {};
    auto issue_n = [&](const int) {};
    auto issue_k = [&](const int) {};
    (void)OOB; (void)dy_col_ok; (void)k_ok; (void)kr; (void)kq; (void)kch; (void)chunk_end; (void)n_row; (void)k_row;
    (void)NW; (void)RN; (void)RK; (void)issue_chunk_idx;
#endif
    auto issue_chunk = [&](const int buf) {
        issue_pre(buf);
#pragma unroll
        for (int i = 0; i < IN_; ++i) issue_n(i);
#pragma unroll
        for (int i = 0; i < IK; ++i) issue_k(i);
        ++issue_chunk_idx;
    };

    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float bsum = 0.f;   // bias gradient partial of channel n0 + tid (threads 0..BN-1 of the k-tile-0 workgroups)
    const bool do_bias = p.db_part != nullptr && tile_k == 0 && tid < BN;

    constexpr int D = WG_NBUF - 1;
    int buf_issue = 0, buf_cur = 0;
    auto next = [](int v) { return v + 1 == WG_NBUF ? 0 : v + 1; };
#pragma unroll
    for (int c = 0; c < D; ++c) {
        issue_chunk(buf_issue);
        buf_issue = next(buf_issue);
    }
    // fragment addresses: lane (i = lane & 31, p = lane >> 5) of MFMA step s reads element i of pixel row 2s + p
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    const unsigned a_addr = lds0 + 4u * ((lane >> 5) * BN + wn * TN * 32 + (lane & 31));
    const unsigned b_addr = lds0 + 4u * (WG_BP * BN + (lane >> 5) * BK + wkk * TK * 32 + (lane & 31));
    const unsigned bias_addr = lds0 + 4u * (unsigned)(tid < BN ? tid : 0);

    // fragments of a whole chunk (16 MFMA steps of 2 pixels), read with inline asm: hipcc cannot tell an LDS-DMA in
    // flight from a slot that landed long ago and would drain vmcnt in front of compiler-generated LDS reads
    float av[1][16][TN], bv[1][16][TK];
    // (the per-step displacement goes into the instruction's offset field: as a computed address every one of the chunk's 32
    // reads cost a v_add, and v_mfma_f32_32x32x2_f32 shares the vector ALU with them - they do not hide behind the MFMAs)
    auto ds_read32 = [](float& dst, unsigned addr, auto off) {
        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value));
    };
    auto read_all = [&](const unsigned slot) {
        const unsigned a0 = a_addr + slot, b0 = b_addr + slot;
        auto step = [&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            if constexpr (TN >= 1) ds_read32(av[0][st][0], a0, std::integral_constant<int, 4 * (2 * st * BN)>{});
            if constexpr (TN >= 2) ds_read32(av[0][st][TN >= 2 ? 1 : 0], a0, std::integral_constant<int, 4 * (2 * st * BN + 32)>{});
            if constexpr (TK >= 1) ds_read32(bv[0][st][0], b0, std::integral_constant<int, 4 * (2 * st * BK)>{});
            if constexpr (TK >= 2) ds_read32(bv[0][st][TK >= 2 ? 1 : 0], b0, std::integral_constant<int, 4 * (2 * st * BK + 32)>{});
        };
        step(std::integral_constant<int, 0>{}); step(std::integral_constant<int, 1>{}); step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{}); step(std::integral_constant<int, 4>{}); step(std::integral_constant<int, 5>{});
        step(std::integral_constant<int, 6>{}); step(std::integral_constant<int, 7>{}); step(std::integral_constant<int, 8>{});
        step(std::integral_constant<int, 9>{}); step(std::integral_constant<int, 10>{}); step(std::integral_constant<int, 11>{});
        step(std::integral_constant<int, 12>{}); step(std::integral_constant<int, 13>{}); step(std::integral_constant<int, 14>{});
        step(std::integral_constant<int, 15>{});
    };
    static_assert(TN <= 2 && TK <= 2 && 4 * (30 * (BN > BK ? BN : BK) + 32) < 65536, "fragment reads are written out for at most 2x2 MFMA tiles per wave");
    // wait until at most `left` of the reads issued so far are outstanding (LDS operations return in order), then pin the
    // fragment registers of steps [s0, s1) so that nothing is scheduled across the wait
    auto landed = [&](auto left_c, const int s0, const int s1) {
        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(left_c)::value) : "memory");
#pragma unroll
        for (int st = s0; st < s1; ++st) {
#pragma unroll
            for (int i = 0; i < TN; ++i) asm volatile("" : "+v"(av[0][st][i]));
#pragma unroll
            for (int j = 0; j < TK; ++j) asm volatile("" : "+v"(bv[0][st][j]));
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto mfma_steps = [&](const int s0, const int s1) {
#pragma unroll
        for (int st = s0; st < s1; ++st)
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0][st][i], bv[0][st][j], acc[i][j], 0, 0, 0);
    };

    {
        // plain order: refill the ring, read the whole chunk's fragments, multiply.  With two workgroups resident per CU
        // (small tiles, 64 KiB of LDS each) the other workgroup's waves fill the matrix core meanwhile.
        for (int c = 0; c < nchunks; ++c) {
            wait_vmcnt<INST*(D - 1)>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            issue_chunk(buf_issue);
            buf_issue = next(buf_issue);
            const unsigned slot = (unsigned)buf_cur * (BUF_FLOATS * 4);
            // all 16 steps' fragments are requested at once; the MFMAs of the first half start as soon as ITS reads are back
            // (lgkmcnt counts down in order), the second half's latency hides behind them
            read_all(slot);
            // (lgkmcnt is a 4-bit counter: the tail that may stay outstanding is the largest whole number of steps within 15 reads)
            constexpr int kTailSteps = 15 / (TN + TK), kSplit = 16 - kTailSteps, kTailReads = kTailSteps * (TN + TK);
            landed(std::integral_constant<int, kTailReads>{}, 0, kSplit);
            mfma_steps(0, kSplit);
            landed(std::integral_constant<int, 0>{}, kSplit, 16);
            mfma_steps(kSplit, 16);
            if (do_bias) {
                float t = 0.f;
#pragma unroll
                for (int r0 = 0; r0 < WG_BP; r0 += 8) {
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(bias_addr + slot + 4u * ((r0 + r) * BN)));
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        asm volatile("" : "+v"(v[r]));
                        t += v[r];
                    }
                }
                bsum += t;
            }
            buf_cur = next(buf_cur);
        }
    }
    wait_vmcnt<0>();

    // dW partial slab [split][cout][k]: C/D map col = lane & 31 (k), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cout)
    float* slab = p.dw_part + (size_t)split * p.slab_floats;
#pragma unroll
    for (int j = 0; j < TK; ++j) {
        const int kcol = k0 + (wkk * TK + j) * 32 + (lane & 31);
        if (kcol >= p.K) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < p.Cout) *(gf_ptr)(slab + (size_t)n * p.K + kcol) = acc[i][j][r];
            }
        }
    }
    if (do_bias && n0 + tid < p.Cout) p.db_part[(size_t)split * p.slab_floats + n0 + tid] = bsum;
}

template <int TN, int TK, int WAVES_N, int WAVES_K, int NBUF>
__global__ __launch_bounds__(64 * WAVES_N * WAVES_K) void conv_wgrad_kernel(WgradP p) {
    __shared__ __attribute__((aligned(16))) float smem[NBUF * WgCfg<TN, TK, WAVES_N, WAVES_K, NBUF>::BUF_FLOATS];
    wgrad_body<TN, TK, WAVES_N, WAVES_K, NBUF>(p, blockIdx.x, smem);
}

// Several layers' weight gradients in ONE launch (the four output convolutions of an inception module become ready
// together, and most of them are small: alone they are launch-latency bound).  The problems travel in the kernel
// arguments like the forward groups (conv_fwd.hip).
constexpr int kMaxWgGroup = 4;
struct WgradGroupArgs {
    int n;
    int wg_end[kMaxWgGroup];      // exclusive prefix of workgroups per problem; the unused entries repeat the last one
    WgradP p[kMaxWgGroup];
};

template <int TN, int TK, int WAVES_N, int WAVES_K, int NBUF>
__global__ __launch_bounds__(64 * WAVES_N * WAVES_K) void conv_wgrad_group_kernel(const WgradGroupArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[NBUF * WgCfg<TN, TK, WAVES_N, WAVES_K, NBUF>::BUF_FLOATS];
    const int b = blockIdx.x;
    int pi = 0, begin = 0;
#pragma unroll
    for (int i = 0; i < kMaxWgGroup - 1; ++i) {
        const bool past = i + 1 < a.n && b >= a.wg_end[i];
        pi += past ? 1 : 0;
        begin = past ? a.wg_end[i] : begin;
    }
    typedef const WgradGroupArgs __attribute__((address_space(4))) * karg_ptr;
    karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    union { WgradP p; unsigned w[sizeof(WgradP) / 4]; } u;
    const unsigned __attribute__((address_space(4)))* src = (const unsigned __attribute__((address_space(4)))*)&ka->p[pi];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(WgradP) / 4); ++i) u.w[i] = src[i];
    wgrad_body<TN, TK, WAVES_N, WAVES_K, NBUF>(u.p, b - begin, smem);
}

// ---- role-split weight gradient (round 3) ------------------------------------------------------------------------------------------
// Why a second kernel (tools/train_profile.py at batch 8, round 3): the 64 x 64 tiles above re-read their operands once per tile
// position - conv2/3x3 pulls 4.2 GB through the L2s for 103 MB of operands (16 TB/s at its 261 us) - and one instruction stream
// per wave serialises address arithmetic, LDS-DMA issue, 32 fragment reads and 16 MFMAs per chunk; three resident workgroups per CU
// hide some of that and the family sustains 45-85 TF/s.  Larger tiles halve the restaging but leave one workgroup per CU, where the
// serialisation shows (round 2's sweep), and 128-wide tiles waste up to a third of their MFMAs on this net's channel counts
// (Cout = 16 .. 384 in steps of 16 or 32, K = 192 .. 3456).  Here
//   * a workgroup has EIGHT waves with fixed roles: waves 4-7 stage, waves 0-3 multiply.  The multiplying waves run fragment reads
//     and MFMAs only (the next step's fragments in flight, the next chunk's first fragments read during this chunk's last step),
//     one raw s_barrier per chunk;
//   * a workgroup owns a REGION of rn x rk sub-tiles of 32 channels x 32 k (rn <= 4, rk <= 8, rn rk <= 16; chosen per problem by
//     the host so that the padded work is least).  The region's VALID sub-tiles are dealt to the four multiplying waves round-robin,
//     up to four accumulators per wave, each bound to its sub-tile by two LDS addresses: a region at the edge of the matrix costs
//     ceil(valid / 4) MFMA streams instead of 4 - padding is paid in units of one sub-tile per wave, not of a 128 x 128 tile;
//   * a chunk is 16 pixels: dY rows at a pitch of 128 floats, im2col rows at a pitch of 256 floats (24 KiB per slot, six slots: up to
//     four chunks in flight - with 32-pixel chunks and three slots only one was, and a workgroup with one or two MFMAs per step took
//     a trip to memory per chunk, 3300 cycles instead of 1024-2048).
//     Constant pitches keep every fragment read's displacement in the instruction's offset field for any region shape; lanes
//     outside the region or the matrix carry an out-of-range offset (zeros, no traffic).  An im2col row is ONE LDS-DMA instruction:
//     its pixel is wave-uniform and decoded on the scalar unit (the per-lane decode of the family above is what the verdict of
//     round 2 asked to remove), a lane adds its tap's constant offset and tests its tap against the image;
//   * workgroups are numbered split-major and dealt to the XCDs in consecutive runs (b % 8 labels the XCD): the workgroups that
//     share an L2 work on the same pixels at the same time, so a chunk's operands are fetched from HBM/MALL once per XCD.
// Results: per split the same sum order as the family above (pixels ascending, two per MFMA step); the split counts differ.
// Diagnostic build only (make exp EXP=-DFCN_WS_STAMPS EXPSRC=train; tools/wgrad_timeline.py): per workgroup, when it ran (100 MHz clock),
// the shader cycles it took, and the cycles multiplying wave 0 / staging wave 0 waited.  No stamp exists in the product build.
#ifdef FCN_WS_STAMPS
__device__ unsigned long long* g_ws_stamps = nullptr;
__device__ int g_ws_stamps_cap = 0;
#define WS_CYC() __builtin_amdgcn_s_memtime()
#else
#define WS_CYC() 0ull
#endif
#ifndef FCN_WS_BP
#define FCN_WS_BP 16
#endif
constexpr int WS_BP = FCN_WS_BP;                     // pixels per chunk (16: six 24 KiB slots; 32: three 48 KiB slots)
constexpr int WS_STEPS = WS_BP / 2;                  // MFMA steps (two pixels each) per chunk
constexpr int WS_DY_FLOATS = WS_BP * 128, WS_X_FLOATS = WS_BP * 256, WS_SLOT_FLOATS = WS_DY_FLOATS + WS_X_FLOATS;
constexpr int WS_NBUF = 144 * 1024 / (WS_SLOT_FLOATS * 4);
constexpr int WS_NDY = WS_BP / 2 / 4, WS_NX = WS_BP / 4;      // per staging wave and chunk: dY pieces (two rows each), im2col rows
constexpr int WS_INST = WS_NDY + WS_NX;              // LDS-DMA instructions per staging wave and chunk
static_assert(WS_BP == 16 || WS_BP == 32, "chunk sizes the step list below is written for");
static_assert((WS_NBUF * WS_SLOT_FLOATS + 4 * 128) * 4 <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
static_assert(WS_INST * (WS_NBUF - 2) <= 63, "vmcnt is a 6-bit counter");

__global__ __launch_bounds__(512) void conv_wgrad_split_kernel(const WgradGroupArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float smem[WS_NBUF * WS_SLOT_FLOATS + 4 * 128];      // + the staging waves' bias partials
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int v;      // this workgroup's place in the split-major order: the blocks of one XCD take a consecutive run
    {
        const int G = gridDim.x, q = G >> 3, r = G & 7, x = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    }
    int pi = 0, begin = 0;
#pragma unroll
    for (int i = 0; i < kMaxWgGroup - 1; ++i) {
        const bool past = i + 1 < a.n && v >= a.wg_end[i];
        pi += past ? 1 : 0;
        begin = past ? a.wg_end[i] : begin;
    }
    typedef const WgradGroupArgs __attribute__((address_space(4))) * karg_ptr;
    karg_ptr ka = (karg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    union { WgradP p; unsigned w[sizeof(WgradP) / 4]; } u;
    const unsigned __attribute__((address_space(4)))* src = (const unsigned __attribute__((address_space(4)))*)&ka->p[pi];
#pragma unroll
    for (int i = 0; i < (int)(sizeof(WgradP) / 4); ++i) u.w[i] = src[i];
    const WgradP& p = u.p;
#ifdef FCN_WS_STAMPS
    const unsigned long long ws_rt0 = __builtin_amdgcn_s_memrealtime(), ws_c0 = WS_CYC();
    unsigned long long ws_wait = 0;
#endif

    // Workgroups of a problem are ordered by the pixel run they START at (then class, then region): the neighbours in this order -
    // which share an XCD and run at the same time - read the same pixels, whatever their class.  before(j) = workgroups that start
    // in front of run j; this workgroup's run is found by bisection (scalar arithmetic, once per workgroup).
    const int lv = v - begin;
    auto before = [&](const int j) __attribute__((always_inline)) {
        int t = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int started = ((j + p.cls_d[c] - 1) * p.cls_dmagic[c]) >> 16;      // ceil(j / d): exact for j < 8192 / d
            t += p.cls_count[c] * min(started, p.cls_splits[c]);
        }
        return t;
    };
    int run = 0;
    {
        int lo = 0, hi = p.runs;      // before(0) = 0 <= lv < before(runs) = the problem's workgroups
#pragma unroll 1
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (before(mid) <= lv) lo = mid; else hi = mid;
        }
        run = lo;
    }
    int off = lv - before(run), cls = 0, split = 0, tile = 0;
    bool found = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int sp = (run * p.cls_dmagic[c]) >> 16;      // run / d
        const bool starts = sp * p.cls_d[c] == run && sp < p.cls_splits[c];
        const int cnt_c = starts ? p.cls_count[c] : 0;
        if (!found && off < cnt_c) { found = true; cls = c; split = sp; tile = off; }
        if (!found) off -= cnt_c;
    }
    const int ca = cls >> 1, cb = cls & 1;
    const int cnk = cb ? 1 : p.tiles_k - 1;
    const int cd = cls == 0 ? p.cls_d[0] : cls == 1 ? p.cls_d[1] : cls == 2 ? p.cls_d[2] : p.cls_d[3];
    const int tn_i = tile / cnk;
    const int tile_n = ca ? p.tiles_n - 1 : tn_i, tile_k = cb ? p.tiles_k - 1 : tile - tn_i * cnk;
    const int n0 = tile_n * 32 * p.rn, k0 = tile_k * 32 * p.rk;
    const int total_chunks = (p.M + WS_BP - 1) / WS_BP;
    const int cps = cd * p.cps0;
    const int chunk0 = split * cps;
    int nchunks = total_chunks - chunk0;
    if (nchunks > cps) nchunks = cps;
    if (nchunks < 0) nchunks = 0;
    const int chunk_end = chunk0 + nchunks;
    constexpr int SLOT_BYTES = WS_SLOT_FLOATS * 4;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
    auto next = [](int b) __attribute__((always_inline)) { return b + 1 == WS_NBUF ? 0 : b + 1; };

    // Barrier protocol (every barrier is passed by all eight waves): B(-1): chunk 0 has landed.  B(c), c = 0 .. nchunks - 1: chunk
    // c + 1 has landed (the multiplying waves read its first fragments during chunk c's last step) and the multiplying waves are
    // done with chunk c - 1, whose slot takes chunk c + 2.
    if (wid >= 4) {
        // ---- staging waves ---------------------------------------------------------------------------------------------------------
        const int lw = wid - 4, ltid = tid - 256;
        constexpr int OOB = (int)0x80000000u;      // >= num_records (wgrad_validate: both operands stay below 2 GiB)
        const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dy), 0, (int)((((long long)p.M - 1) * p.dy_cstride + (p.Cout + 3) / 4 * 4) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t rxx = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x), 0, (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * 4), 0x00020000);
        // dY piece j = lw + 4 i holds pixel rows 2 j, 2 j + 1 of the chunk: lane -> row 2 j + lane / 32, channels n0 + 4 (lane % 32) ..
        const int col = (lane & 31) * 4;
        const bool col_ok = col < 32 * p.rn && n0 + col < p.Cout;
        int n_vo[WS_NDY];
#pragma unroll
        for (int i = 0; i < WS_NDY; ++i) n_vo[i] = col_ok ? ((2 * (lw + 4 * i) + (lane >> 5)) * p.dy_cstride + n0 + col) * 4 : OOB;
        // im2col row = lw + 4 i: lane -> k = k0 + 4 lane .. + 3 (one tap: Cin % 4 == 0), a constant offset from the row's pixel.
        // v_mfma_f32_32x32x2_f32 runs at the vector ALU's own rate and occupies it: a vector instruction of a staging wave waits for the
        // MFMA in flight on its SIMD (up to 64 cycles) and then delays the next one.  A first version computed every row's lane
        // offsets and halo tests with ~12 vector instructions; eight rows of that per chunk took longer than the chunk's MFMAs, and
        // the multiplying waves waited at every barrier (65 % of the MFMA rate with NO loads at all).  So: the pixel is decoded on the
        // scalar unit; a row whose taps all lie inside the image - most rows - is fetched with a CONSTANT lane offset and the pixel's
        // base in the instruction's scalar offset: no vector instruction at all.  Only rows at the image border test their taps.
        const int kidx = k0 + 4 * lane;
        const bool k_ok = 4 * lane < 32 * p.rk && kidx < p.K;
        const int ktap = kidx / p.Cin;
        const int kch = kidx - ktap * p.Cin;
        const int kr = (ktap * p.kw_magic) >> 16;
        const int kq = ktap - kr * p.kw;
        const int x_lane = ((kr * p.W + kq) * p.x_cstride + kch) * 4;
        int x_lane_k = k_ok ? x_lane : OOB;
        asm volatile("" : "+v"(x_lane_k));
        int issue_chunk_idx = chunk0;
        auto issue_chunk = [&](const int buf) __attribute__((always_inline)) {
            float* const dst = smem + buf * WS_SLOT_FLOATS;
            const bool live = issue_chunk_idx < chunk_end;
            const int s_dy = live ? issue_chunk_idx * WS_BP * p.dy_cstride * 4 : 0x7F000000;      // rows >= M are past num_records by themselves
#pragma unroll
            for (int i = 0; i < WS_NDY; ++i) {
#ifndef FCN_WS_NOLOAD
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rdy, (lds_ptr)(dst + (lw + 4 * i) * 256), 16, n_vo[i], s_dy, 0, 0);
#endif
            }
#pragma unroll
            for (int i = 0; i < WS_NX; ++i) {
                const int row = lw + 4 * i;
                const int m = issue_chunk_idx * WS_BP + row;      // wave-uniform: everything up to the branch runs on the scalar unit
                const bool m_ok = live && m < p.M;
                const int mm = m_ok ? m : 0;
                const int t = p.ow_magic ? (int)__umulhi((unsigned)mm, p.ow_magic) : mm;      // exact: wgrad_validate
                const int ox = mm - t * p.OW;
                const int img = p.oh_magic ? (int)__umulhi((unsigned)t, p.oh_magic) : t;
                const int oy = t - img * p.OH;
                const int by = oy * p.stride - p.pad, bx = ox * p.stride - p.pad;
                const int base = ((img * p.H + by) * p.W + bx) * p.x_cstride * 4;
                lds_ptr const d = (lds_ptr)(dst + WS_DY_FLOATS + row * 256);
                const bool inside = by >= 0 && by + p.kh <= p.H && bx >= 0 && bx + p.kw <= p.W;
                if (inside || !m_ok) {
                    const int s_x = m_ok ? base : 0x7F000000;      // (base >= 0 inside the image; a dead row: every lane out of range)
#ifndef FCN_WS_NOLOAD
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rxx, d, 16, x_lane_k, s_x, 0, 0);
#endif
                } else {
                    const bool ok = (int)k_ok & (int)((unsigned)(by + kr) < (unsigned)p.H) & (int)((unsigned)(bx + kq) < (unsigned)p.W);
                    int vo = ok ? base + x_lane : OOB;
                    asm volatile("" : "+v"(vo));
#ifndef FCN_WS_NOLOAD
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rxx, d, 16, vo, 0, 0, 0);
#endif
                }
            }
            ++issue_chunk_idx;
        };
        // bias gradient = column sums of dY, in the workgroups of the first k-tile, from the landed chunk: every staging wave sums
        // a quarter of the chunk's rows (a vector add beside v_mfma_f32_32x32x2_f32 costs its SIMD's matrix stream a slot: spread over
        // the four SIMDs), two channels per lane; the four partials meet in LDS behind the last chunk.
        const bool do_bias = p.db_part != nullptr && tile_k == 0;
        const unsigned bias_addr = lds0 + 4u * (unsigned)(WS_NX * lw * 128 + 2 * lane);
        float bsum0 = 0.f, bsum1 = 0.f;
        int buf_issue = 0, buf_cur = 0;
#pragma unroll 1
        for (int c = 0; c < WS_NBUF - 1; ++c) {
            issue_chunk(buf_issue);
            buf_issue = next(buf_issue);
        }
        wait_vmcnt<WS_INST*(WS_NBUF - 2)>();
        __builtin_amdgcn_s_barrier();      // B(-1)
#pragma unroll 1
        for (int c = 0; c < nchunks; ++c) {
#ifdef FCN_WS_STAMPS
            const unsigned long long w0 = WS_CYC();
#endif
            wait_vmcnt<WS_INST*(WS_NBUF - 3)>();
#ifdef FCN_WS_STAMPS
            ws_wait += WS_CYC() - w0;
#endif
            __builtin_amdgcn_s_barrier();      // B(c)
            asm volatile("" ::: "memory");
            issue_chunk(buf_issue);
            buf_issue = next(buf_issue);
            if (do_bias) {
                const unsigned slot = (unsigned)buf_cur * SLOT_BYTES;
                typedef float v2f __attribute__((ext_vector_type(2)));
                v2f vv[WS_NX];
#pragma unroll
                for (int r = 0; r < WS_NX; ++r) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(vv[r]) : "v"(bias_addr + slot), "n"(r * 128 * 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float t0 = 0.f, t1 = 0.f;
#pragma unroll
                for (int r = 0; r < WS_NX; ++r) {
                    asm volatile("" : "+v"(vv[r]));
                    t0 += vv[r][0];
                    t1 += vv[r][1];
                }
                bsum0 += t0;
                bsum1 += t1;
            }
            buf_cur = next(buf_cur);
        }
        wait_vmcnt<0>();      // the all-zero chunks behind the last one
        float* const bias_lds = smem + WS_NBUF * WS_SLOT_FLOATS;
        if (do_bias) {
            bias_lds[lw * 128 + 2 * lane] = bsum0;
            bias_lds[lw * 128 + 2 * lane + 1] = bsum1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();      // (the multiplying waves pass it behind their last chunk)
        asm volatile("" ::: "memory");
        if (do_bias && ltid < 32 * p.rn && n0 + ltid < p.Cout)
            p.db_part[(size_t)split * p.slab_floats + n0 + ltid] = ((bias_lds[ltid] + bias_lds[128 + ltid]) + bias_lds[256 + ltid]) + bias_lds[384 + ltid];
#ifdef FCN_WS_STAMPS
        if (g_ws_stamps && (int)blockIdx.x < g_ws_stamps_cap && tid == 256) g_ws_stamps[(size_t)blockIdx.x * 8 + 7] = ws_wait;
#endif
    } else {
        // ---- multiplying waves -----------------------------------------------------------------------------------------------------
        // The region's valid sub-tiles are taken in PAIRS along k (one dY fragment serves both): pair list dealt round-robin.
        const int rn_v = min(p.rn, (p.Cout - n0 + 31) / 32), rk_v = min(p.rk, (p.K - k0 + 31) / 32);
        const int kp = (rk_v + 1) >> 1, pairs = rn_v * kp;
        const int cnt = pairs > wid ? (pairs - wid + 3) / 4 : 0;      // this wave's pairs: wid, wid + 4, ..
        unsigned a_ad[4], b_ad[4];
        int sub_n[4], sub_k[4];
        int full = 0;                                                  // bit t: pair t has its second sub-tile
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int pt = min(wid + 4 * t, pairs - 1);
            sub_n[t] = pt / kp;
            sub_k[t] = 2 * (pt - sub_n[t] * kp);
            if (t < cnt && sub_k[t] + 1 < rk_v) full |= 1 << t;
            // lane (i = lane & 31, p = lane >> 5) of MFMA step s reads element i of pixel row 2 s + p
            a_ad[t] = lds0 + 4u * (unsigned)((lane >> 5) * 128 + 32 * sub_n[t] + (lane & 31));
            b_ad[t] = lds0 + 4u * (unsigned)(WS_DY_FLOATS + (lane >> 5) * 256 + 32 * sub_k[t] + (lane & 31));
        }
        full = __builtin_amdgcn_readfirstlane(full);
        f32x16 acc[4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][h][r] = 0.f;
        auto ds_read32 = [](float& dst, unsigned addr, auto off) __attribute__((always_inline)) {
#ifdef FCN_WS_NOREAD
            asm volatile("" : "=v"(dst) : "v"(addr));
            return;
#endif
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value));
        };
        // CNT pairs; ALL: every pair is whole (no branch in the MFMA stream)
        // Round 4: the fragments of TWO consecutive steps come with one instruction.  A lane's operand of step s and of step s + 1 are
        // 1024 (dY) / 2048 (im2col) bytes apart - whole multiples of 256 - so ds_read2st64_b32 (two dwords at base + 256 * offset0 / offset1)
        // fetches both: 48 fragment reads per chunk and pair list instead of 96.  The multiplying waves' stream was 1540 cycles per
        // chunk where 1024 are MFMA: the reads' issue slots were a third of the rest.  The accumulation order per
        // accumulator is unchanged (steps ascending), so the gradients are the same bits.
        typedef float v2f __attribute__((ext_vector_type(2)));
        auto ds_read2 = [](v2f& dst, unsigned addr, auto o0, auto o1) __attribute__((always_inline)) {
#ifdef FCN_WS_NOREAD
            asm volatile("" : "=v"(dst) : "v"(addr));
            return;
#endif
            asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(addr), "n"(decltype(o0)::value), "n"(decltype(o1)::value));
        };
        auto run = [&](auto cnt_c, auto all_c) __attribute__((always_inline)) {
            constexpr int CNT = decltype(cnt_c)::value;
            constexpr bool ALL = decltype(all_c)::value;
            constexpr int SS = WS_STEPS / 2;      // super-steps of two MFMA steps
            v2f fa[2][CNT], fb[2][CNT][2];        // .x: the even step, .y: the odd one
            unsigned b1_ad[CNT];                  // the pair's second sub-tile: 32 floats on
#pragma unroll
            for (int t = 0; t < CNT; ++t) b1_ad[t] = b_ad[t] + 128u;
            unsigned slot = 0;
            __builtin_amdgcn_s_barrier();      // B(-1)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int t = 0; t < CNT; ++t) {
                ds_read2(fa[0][t], a_ad[t], std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                ds_read2(fb[0][t][0], b_ad[t], std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
                ds_read2(fb[0][t][1], b1_ad[t], std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
            }
#pragma unroll 1
            for (int c = 0; c < nchunks; ++c) {
#ifdef FCN_WS_STAMPS
                const unsigned long long w0 = WS_CYC();
#endif
                __builtin_amdgcn_s_barrier();      // B(c)
                asm volatile("" ::: "memory");
#ifdef FCN_WS_STAMPS
                ws_wait += WS_CYC() - w0;
#endif
                const unsigned slot_next = slot + SLOT_BYTES == (unsigned)(WS_NBUF * SLOT_BYTES) ? 0u : slot + SLOT_BYTES;
                unsigned aa[CNT], bb[CNT], bb1[CNT];
#pragma unroll
                for (int t = 0; t < CNT; ++t) {
                    aa[t] = a_ad[t] + slot;
                    bb[t] = b_ad[t] + slot;
                    bb1[t] = b1_ad[t] + slot;
                }
                auto sstep = [&](auto ss_c) __attribute__((always_inline)) {
                    constexpr int ss = decltype(ss_c)::value, par = ss & 1;
                    // the next super-step's fragments (behind the last one: the first ones of the next chunk)
#pragma unroll
                    for (int t = 0; t < CNT; ++t) {
                        if constexpr (ss < SS - 1) {
                            ds_read2(fa[par ^ 1][t], aa[t], std::integral_constant<int, 8 * (ss + 1)>{}, std::integral_constant<int, 8 * (ss + 1) + 4>{});
                            ds_read2(fb[par ^ 1][t][0], bb[t], std::integral_constant<int, 16 * (ss + 1)>{}, std::integral_constant<int, 16 * (ss + 1) + 8>{});
                            ds_read2(fb[par ^ 1][t][1], bb1[t], std::integral_constant<int, 16 * (ss + 1)>{}, std::integral_constant<int, 16 * (ss + 1) + 8>{});
                        } else {
                            ds_read2(fa[par ^ 1][t], a_ad[t] + slot_next, std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{});
                            ds_read2(fb[par ^ 1][t][0], b_ad[t] + slot_next, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
                            ds_read2(fb[par ^ 1][t][1], b1_ad[t] + slot_next, std::integral_constant<int, 0>{}, std::integral_constant<int, 8>{});
                        }
                    }
                    // LDS operations return in order: this super-step's fragments are back once only the 3 CNT just issued are outstanding
                    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(3 * CNT) : "memory");
#pragma unroll
                    for (int t = 0; t < CNT; ++t) asm volatile("" : "+v"(fa[par][t]), "+v"(fb[par][t][0]), "+v"(fb[par][t][1]));
                    __builtin_amdgcn_sched_barrier(0);
#ifndef FCN_WS_NOMFMA
#pragma unroll
                    for (int e = 0; e < 2; ++e)      // the even step of every pair, then the odd one: per accumulator the steps stay in order
#pragma unroll
                        for (int t = 0; t < CNT; ++t) {
                            acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][t][e], fb[par][t][0][e], acc[t][0], 0, 0, 0);
                            if (ALL || ((full >> t) & 1)) acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[par][t][e], fb[par][t][1][e], acc[t][1], 0, 0, 0);
                        }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                };
                static_assert(SS % 2 == 0, "the fragment parity of a chunk's first super-step must be 0");
                sstep(std::integral_constant<int, 0>{}); sstep(std::integral_constant<int, 1>{});
                sstep(std::integral_constant<int, 2>{}); sstep(std::integral_constant<int, 3>{});
                if constexpr (SS == 8) {
                    sstep(std::integral_constant<int, 4 % SS>{}); sstep(std::integral_constant<int, 5 % SS>{});
                    sstep(std::integral_constant<int, 6 % SS>{}); sstep(std::integral_constant<int, 7 % SS>{});
                }
                slot = slot_next;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the fragments read last belong to a chunk nobody multiplies
#pragma unroll
            for (int t = 0; t < CNT; ++t) asm volatile("" : "+v"(fa[0][t]), "+v"(fb[0][t][0]), "+v"(fb[0][t][1]));
        };
        const bool all = full == (1 << cnt) - 1;
        typedef std::true_type T_;
        typedef std::false_type F_;
        if (cnt == 4) { if (all) run(std::integral_constant<int, 4>{}, T_{}); else run(std::integral_constant<int, 4>{}, F_{}); }
        else if (cnt == 3) { if (all) run(std::integral_constant<int, 3>{}, T_{}); else run(std::integral_constant<int, 3>{}, F_{}); }
        else if (cnt == 2) { if (all) run(std::integral_constant<int, 2>{}, T_{}); else run(std::integral_constant<int, 2>{}, F_{}); }
        else if (cnt == 1) { if (all) run(std::integral_constant<int, 1>{}, T_{}); else run(std::integral_constant<int, 1>{}, F_{}); }
        else {
#pragma unroll 1
            for (int c = 0; c <= nchunks; ++c) __builtin_amdgcn_s_barrier();
        }
        __builtin_amdgcn_s_barrier();      // the staging waves' bias partials are in LDS
#ifdef FCN_WS_STAMPS
        const unsigned long long ws_c1 = WS_CYC();
#endif
        // dW partial slab [split][cout][k]: C/D map col = lane & 31 (k), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (cout)
        float* slab = p.dw_part + (size_t)split * p.slab_floats;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kcol = k0 + 32 * (sub_k[t] + h) + (lane & 31);
                if (t >= cnt || (h && !((full >> t) & 1)) || kcol >= p.K) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = n0 + 32 * sub_n[t] + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (n < p.Cout) *(gf_ptr)(slab + (size_t)n * p.K + kcol) = acc[t][h][r];
                }
            }
        }
#ifdef FCN_WS_STAMPS
        if (g_ws_stamps && (int)blockIdx.x < g_ws_stamps_cap && tid == 0) {
            unsigned long long* d_ = g_ws_stamps + (size_t)blockIdx.x * 8;
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            d_[0] = ws_rt0; d_[1] = __builtin_amdgcn_s_memrealtime(); d_[2] = ws_c1 - ws_c0; d_[3] = WS_CYC() - ws_c1;
            d_[4] = (unsigned long long)nchunks; d_[5] = (unsigned long long)cnt | ((unsigned long long)(xcc & 15) << 8); d_[6] = ws_wait;
        }
#endif
    }
#endif
}

// tile shapes: X(index, TN, TK, WAVES_N, WAVES_K)
#define FCN_WGRAD_CONFIGS(X) \
    X(0, 1, 1, 2, 2, 3)      \
    X(1, 1, 2, 2, 2, 3)      \
    X(2, 2, 1, 2, 2, 3)      \
    X(3, 1, 1, 2, 2, 4)
struct WgShape { int bn, bk, nw; double eff; };   // eff: sustained fraction of the MFMA rate seen in the sweep (ranking only)
constexpr int kNumWgCfg = 4;
constexpr WgShape kWgShapes[kNumWgCfg] = {{64, 64, 4, 0.55}, {64, 128, 4, 0.50}, {128, 64, 4, 0.60}, {64, 64, 4, 0.50}};
constexpr int kSplitCfg = kNumWgCfg;      // conv_wgrad_split_kernel: a region shape per problem instead of one tile shape per launch

// out[i] = sum_s parts[s][i] in a FIXED order (slab 0, 1, 2, ..: bit-reproducible).  One lane owns four consecutive outputs
// and walks the slabs with 16-byte loads, eight of them in flight at a time (the adds stay in slab order): every access is
// a full 1 KiB wave-instruction.  Round 1 gave each output 16 lanes that read 4 bytes per slab - 256 B per wave-instruction,
// 64 outputs per 1024-thread block, two barriers - and the pass cost 10 % of the training step (profiles/r01_train_*).
// Round 3: FOUR waves walk the slabs of one run of outputs (wave w takes the w-th quarter of the slabs, in order; the four quarter sums
// are added as (Q0 + Q1) + (Q2 + Q3) through LDS): a small layer's matrix is a few hundred float4 columns, and with one lane walking all
// of a column's 40-60 slabs eight loads at a time the pass was a handful of workgroups waiting for memory - as long as the MFMA launch it
// follows (tools/wgrad_timeline.py: inception_4a's 1x1 group, kernel 20.9 us, kernel + reduction 41.8 us).  Still one fixed order.
constexpr int RED_THREADS = 256, RED_WALKERS = 4, RED_PER_BLOCK = RED_THREADS / RED_WALKERS * 4;
// How many slabs hold element i.  The role-split kernel gives the region classes of a problem their own split counts: weights
// (i < wcount) by the class of (row, column) of the Cout x K matrix, the bias sums behind them by the class of their channel in
// the first k-region.  Every other producer fills all slabs: uniform.
struct RedSplits {
    long long wcount;
    int K, n_last0, k_last0, b_bias, uniform;
    int s[4];
};
__device__ __forceinline__ int red_splits_for(const RedSplits& rs, size_t i) {
    if (rs.uniform) return rs.s[0];
    int a, b;
    if ((long long)i < rs.wcount) {
        const int n = (int)(i / (size_t)rs.K), k = (int)(i - (size_t)n * rs.K);
        a = n >= rs.n_last0;
        b = k >= rs.k_last0;
    } else {
        a = (int)((long long)i - rs.wcount) >= rs.n_last0;
        b = rs.b_bias;
    }
    const int c = 2 * a + b;
    return c == 0 ? rs.s[0] : c == 1 ? rs.s[1] : c == 2 ? rs.s[2] : rs.s[3];
}
static RedSplits red_uniform(int splits) {
    RedSplits r = {0, 1, 0, 0, 0, 1, {splits, splits, splits, splits}};
    return r;
}
// i4: first of this lane's four outputs; walker: which quarter of the slabs this wave adds; red: RED_WALKERS x 64 float4 of LDS
__device__ __forceinline__ void reduce_slabs(const float* __restrict__ parts, float* __restrict__ out, size_t count, const RedSplits& rs, size_t stride,
                                             size_t i4, const int walker, float* red) {
    typedef float v4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63;
    const bool live = i4 < count;
    const bool vec = live && i4 + 3 < count && (stride & 3) == 0 && (((size_t)parts | (size_t)out) & 15) == 0;
    v4 acc = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        if (vec) {
            const int splits = red_splits_for(rs, i4);      // (four consecutive elements share a class: K, wcount and the class borders are multiples of 4)
            const int per = (splits + RED_WALKERS - 1) / RED_WALKERS;
            const int k0 = walker * per, k1 = min(k0 + per, splits);
            const v4* src = reinterpret_cast<const v4*>(parts + i4);
            const size_t st4 = stride / 4;
            int k = k0;
            for (; k + 7 < k1; k += 8) {
                v4 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = src[(size_t)(k + j) * st4];
#pragma unroll
                for (int j = 0; j < 8; ++j) acc += v[j];
            }
            for (; k < k1; ++k) acc += src[(size_t)k * st4];
        } else {
            for (int e = 0; e < 4; ++e) {
                const size_t i = i4 + e;
                if (i >= count) break;
                const int splits = red_splits_for(rs, i);
                const int per = (splits + RED_WALKERS - 1) / RED_WALKERS;
                const int k0 = walker * per, k1 = min(k0 + per, splits);
                float t = 0.f;
                for (int k = k0; k < k1; ++k) t += parts[(size_t)k * stride + i];
                acc[e] = t;
            }
        }
    }
    reinterpret_cast<v4*>(red)[walker * 64 + lane] = acc;
    __syncthreads();
    if (walker == 0 && live) {
        const v4* r = reinterpret_cast<const v4*>(red);
        const v4 t = (r[lane] + r[64 + lane]) + (r[128 + lane] + r[192 + lane]);
        if (vec) {
            *reinterpret_cast<v4*>(out + i4) = t;
        } else {
            for (int e = 0; e < 4 && i4 + e < count; ++e) out[i4 + e] = t[e];
        }
    }
    __syncthreads();      // (the grid-stride form reuses the scratch)
}

__global__ __launch_bounds__(RED_THREADS) void reduce_partials_kernel(const float* __restrict__ parts, float* __restrict__ out, size_t count,
                                                                      const RedSplits rs, size_t stride) {
    __shared__ __attribute__((aligned(16))) float red[RED_WALKERS * 64 * 4];
    const int walker = threadIdx.x >> 6;
    for (size_t base = (size_t)blockIdx.x * RED_PER_BLOCK; base < count; base += (size_t)gridDim.x * RED_PER_BLOCK)      // (uniform per workgroup)
        reduce_slabs(parts, out, count, rs, stride, base + (size_t)(threadIdx.x & 63) * 4, walker, red);
}

struct ReduceGroupArgs {
    int n;
    int blk_end[kMaxWgGroup];
    const float* parts[kMaxWgGroup];
    float* out[kMaxWgGroup];
    unsigned long long count[kMaxWgGroup], stride[kMaxWgGroup];
    RedSplits rs[kMaxWgGroup];
};

// the same for up to kMaxWgGroup problems in one launch (one problem per block range)
__global__ __launch_bounds__(RED_THREADS) void reduce_partials_group_kernel(const ReduceGroupArgs a) {
    int pi = 0, begin = 0;
#pragma unroll
    for (int i = 0; i < kMaxWgGroup - 1; ++i) {
        const bool past = i + 1 < a.n && (int)blockIdx.x >= a.blk_end[i];
        pi += past ? 1 : 0;
        begin = past ? a.blk_end[i] : begin;
    }
    const float* parts = a.parts[0];
    float* out = a.out[0];
    size_t count = a.count[0], stride = a.stride[0];
    RedSplits rs = a.rs[0];
#pragma unroll
    for (int i = 1; i < kMaxWgGroup; ++i)
        if (pi == i) { parts = a.parts[i]; out = a.out[i]; count = a.count[i]; stride = a.stride[i]; rs = a.rs[i]; }
    __shared__ __attribute__((aligned(16))) float red[RED_WALKERS * 64 * 4];
    reduce_slabs(parts, out, count, rs, stride, ((size_t)((int)blockIdx.x - begin) * 64 + (threadIdx.x & 63)) * 4, threadIdx.x >> 6, red);
}

// wt[c][kh-1-r][kw-1-q][k] = w[k][r][q][c]   (w: [Cout][kh][kw][Cin4], wt: [Cin][kh][kw][Cout4], pads of wt zero)
__global__ __launch_bounds__(256) void weights_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout, int kh, int kw,
                                                           int Cin, int Cin4, int Cout4) {
    const long long total = (long long)Cin * kh * kw * Cout4;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(t % Cout4);
        long long u = t / Cout4;
        const int q = (int)(u % kw);
        u /= kw;
        const int r = (int)(u % kh);
        const int c = (int)(u / kh);
        wt[t] = k < Cout ? w[(((size_t)k * kh + (kh - 1 - r)) * kw + (kw - 1 - q)) * Cin4 + c] : 0.f;
    }
}

// the same for every layer of a net at once: blockIdx.y = segment
__global__ __launch_bounds__(256) void weights_flip_batch_kernel(const float* __restrict__ w_base, float* __restrict__ wt_base,
                                                                 const fcn_flip_seg* __restrict__ segs) {
    const fcn_flip_seg sg = segs[blockIdx.y];
    const float* w = w_base + sg.w_offset;
    float* wt = wt_base + sg.wt_offset;
    const unsigned total = (unsigned)sg.Cin * sg.kh * sg.kw * sg.Cout4;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned k = t % (unsigned)sg.Cout4;
        unsigned u = t / (unsigned)sg.Cout4;
        const unsigned q = u % (unsigned)sg.kw;
        u /= (unsigned)sg.kw;
        const unsigned r = u % (unsigned)sg.kh;
        const unsigned c = u / (unsigned)sg.kh;
        wt[t] = k < (unsigned)sg.Cout ? w[(((size_t)k * sg.kh + (sg.kh - 1 - r)) * sg.kw + (sg.kw - 1 - q)) * sg.Cin4 + c] : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------
// pointwise backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                       size_t count) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

__global__ __launch_bounds__(256) void relu_bwd_view_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx,
                                                            long long pixels, int C, int cstride) {
    const long long total = pixels * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const size_t o = (size_t)(t / C) * cstride + (t % C);
        dx[o] = y[o] > 0.f ? dy[o] : 0.f;
    }
}

__global__ __launch_bounds__(256) void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy, float* __restrict__ dx,
                                                          size_t count, int accumulate) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        const float g = dy[i] * y[i] * (1.f - y[i]);
        dx[i] = accumulate ? dx[i] + g : g;
    }
}

// MAX pool backward in gather form: every input element sums the output gradients whose argmax it is
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                                          int N, int H, int W, int C, int dx_cstride, int dx_coffset, int k, int stride,
                                                          int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate,
                                                          const float* __restrict__ relu_y, int ry_cstride, int ry_coffset) {
    const long long total = (long long)N * H * W * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        long long pix = t / C;
        const int ix = (int)(pix % W);
        pix /= W;
        const int iy = (int)(pix % H);
        const int n = (int)(pix / H);
        // outputs whose window covers (iy, ix): oy*stride - pad <= iy < oy*stride - pad + k
        const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(OH - 1, (iy + pad) / stride);
        const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(OW - 1, (ix + pad) / stride);
        const int me = iy * W + ix;
        float g = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const size_t o = ((size_t)(n * OH + oy) * OW + ox);
                if (idx[o * C + c] == me) g += dy[o * dy_cstride + dy_coffset + c];
            }
        float* d = dx + ((size_t)(n * H + iy) * W + ix) * dx_cstride + dx_coffset + c;
        if (accumulate) g += *d;
        if (relu_y && !(relu_y[((size_t)(n * H + iy) * W + ix) * ry_cstride + ry_coffset + c] > 0.f)) g = 0.f;
        *d = g;
    }
}

// same, 4 channels (16 bytes) per lane: the shapes of the reference nets always allow it
__global__ __launch_bounds__(256) void maxpool_bwd_v4_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                                             int N, int H, int W, int C, int dx_cstride, int dx_coffset, int k, int stride,
                                                             int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate,
                                                             const float* __restrict__ relu_y, int ry_cstride, int ry_coffset) {
    const unsigned C4 = (unsigned)C >> 2;
    const unsigned total = (unsigned)N * H * W * C4;
    for (unsigned t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const unsigned pix = t / C4;
        const int c = (int)(t - pix * C4) * 4;
        const unsigned row = pix / (unsigned)W;
        const int ix = (int)(pix - row * (unsigned)W);
        const int n = (int)(row / (unsigned)H);
        const int iy = (int)(row - (unsigned)n * (unsigned)H);
        const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(OH - 1, (iy + pad) / stride);
        const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(OW - 1, (ix + pad) / stride);
        const int me = iy * W + ix;
        v4f g = {0.f, 0.f, 0.f, 0.f};
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const size_t o = ((size_t)(n * OH + oy) * OW + ox);
                const int4 id = *reinterpret_cast<const int4*>(idx + o * C + c);
                const v4f d = *reinterpret_cast<const v4f*>(dy + o * dy_cstride + dy_coffset + c);
                g[0] += id.x == me ? d[0] : 0.f;
                g[1] += id.y == me ? d[1] : 0.f;
                g[2] += id.z == me ? d[2] : 0.f;
                g[3] += id.w == me ? d[3] : 0.f;
            }
        v4f* dst = reinterpret_cast<v4f*>(dx + (size_t)pix * dx_cstride + dx_coffset + c);
        if (accumulate) g += *dst;
        if (relu_y) {       // the ReLU backward of the blob this gradient belongs to, when this pass is its last writer
            const v4f y = *reinterpret_cast<const v4f*>(relu_y + (size_t)pix * ry_cstride + ry_coffset + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
        }
        *dst = g;
    }
}

// same again with the lanes of a workgroup laid over an 8 x 8 patch of input pixels x 32 channels: the windows of neighbouring
// inputs overlap (each output is read by up to k*k inputs), and with one patch per workgroup that re-reading is served by the
// CU's own L1 instead of nine trips to L2 (grid-stride order put neighbours on different CUs)
__global__ __launch_bounds__(512) void maxpool_bwd_patch_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, float* __restrict__ dx,
                                                                int H, int W, int C, int dx_cstride, int dx_coffset, int k, int stride, int pad,
                                                                int OH, int OW, int dy_cstride, int dy_coffset, int accumulate,
                                                                const float* __restrict__ relu_y, int ry_cstride, int ry_coffset, int cgroups) {
    const int cg = (int)blockIdx.x % cgroups, tx = (int)blockIdx.x / cgroups;
    const int c = cg * 32 + ((int)threadIdx.x & 7) * 4;
    const int pp = (int)threadIdx.x >> 3;
    const int iy = (int)blockIdx.y * 8 + (pp >> 3), ix = tx * 8 + (pp & 7), n = (int)blockIdx.z;
    if (c >= C || iy >= H || ix >= W) return;
    const int oy_lo = max(0, (iy + pad - k + stride) / stride), oy_hi = min(OH - 1, (iy + pad) / stride);
    const int ox_lo = max(0, (ix + pad - k + stride) / stride), ox_hi = min(OW - 1, (ix + pad) / stride);
    const int me = iy * W + ix;
    v4f g = {0.f, 0.f, 0.f, 0.f};
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            const size_t o = ((size_t)(n * OH + oy) * OW + ox);
            const int4 id = *reinterpret_cast<const int4*>(idx + o * C + c);
            const v4f d = *reinterpret_cast<const v4f*>(dy + o * dy_cstride + dy_coffset + c);
            g[0] += id.x == me ? d[0] : 0.f;
            g[1] += id.y == me ? d[1] : 0.f;
            g[2] += id.z == me ? d[2] : 0.f;
            g[3] += id.w == me ? d[3] : 0.f;
        }
    const size_t pix = (size_t)(n * H + iy) * W + ix;
    v4f* dst = reinterpret_cast<v4f*>(dx + pix * dx_cstride + dx_coffset + c);
    if (accumulate) g += *dst;
    if (relu_y) {
        const v4f y = *reinterpret_cast<const v4f*>(relu_y + pix * ry_cstride + ry_coffset + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] = y[e] > 0.f ? g[e] : 0.f;
    }
    *dst = g;
}

// LRN backward: dX = dY*scale^-beta - (2 alpha beta / n) * X * sum_{window} (dY * Y / scale)
__global__ __launch_bounds__(256) void lrn_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ dy, float* __restrict__ dx, long long pixels, int C,
                                                      int x_cstride, int y_cstride, int local_size, float ratio2ab, float beta, int accumulate) {
    const long long total = pixels * C;
    const int pre = (local_size - 1) / 2;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long pix = t / C;
        const float* yp = y + (size_t)pix * y_cstride;
        const float* dyp = dy + (size_t)pix * y_cstride;
        const float* sp = scale + (size_t)pix * C;
        float acc = 0.f;
        // the window of outputs c' that contain input c: c' - pre <= c <= c' - pre + n - 1
        for (int j = c - (local_size - 1 - pre); j <= c + pre; ++j)
            if (j >= 0 && j < C) acc += dyp[j] * yp[j] / sp[j];
        const float g = dyp[c] * powf(sp[c], -beta) - ratio2ab * x[(size_t)pix * x_cstride + c] * acc;
        float* d = dx + (size_t)pix * x_cstride + c;
        *d = accumulate ? *d + g : g;
    }
}

// local_size 5 fast path: one lane = 4 channels, the window terms dY*Y/scale of channels c-4..c+7 come from three
// 16-byte loads per operand (same layout trick as the forward lrn5 kernel)
__global__ __launch_bounds__(256) void lrn5_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ scale,
                                                       const float* __restrict__ dy, float* __restrict__ dx, long long pixels, int C,
                                                       int x_cstride, int y_cstride, float ratio2ab, float beta, int accumulate) {
    const int cg = C / 4;
    const long long total = pixels * cg;
    const v4f zero = {0.f, 0.f, 0.f, 0.f};
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const long long pix = t / cg;
        const int g = (int)(t - pix * cg);
        const float* yp = y + (size_t)pix * y_cstride + g * 4;
        const float* dyp = dy + (size_t)pix * y_cstride + g * 4;
        const float* sp = scale + (size_t)pix * C + g * 4;
        const bool hl = g > 0, hr = g + 1 < cg;
        const v4f yc = *(const v4f*)yp, dc = *(const v4f*)dyp, sc = *(const v4f*)sp;
        const v4f qc = dc * yc / sc;
        const v4f ql = hl ? *(const v4f*)(dyp - 4) * *(const v4f*)(yp - 4) / *(const v4f*)(sp - 4) : zero;
        const v4f qr = hr ? *(const v4f*)(dyp + 4) * *(const v4f*)(yp + 4) / *(const v4f*)(sp + 4) : zero;
        const float q[12] = {ql[0], ql[1], ql[2], ql[3], qc[0], qc[1], qc[2], qc[3], qr[0], qr[1], qr[2], qr[3]};
        const v4f xc = *(const v4f*)(x + (size_t)pix * x_cstride + g * 4);
        v4f o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float acc = q[i + 2] + q[i + 3] + q[i + 4] + q[i + 5] + q[i + 6];     // channels c-2 .. c+2, ascending like the generic kernel
            float p75;
            if (beta == 0.75f) {
                const float r = sqrtf(sc[i]);
                p75 = 1.f / (r * sqrtf(r));
            } else {
                p75 = powf(sc[i], -beta);
            }
            o[i] = dc[i] * p75 - ratio2ab * xc[i] * acc;
        }
        v4f* d = (v4f*)(dx + (size_t)pix * x_cstride + g * 4);
        if (accumulate) o += *d;
        *d = o;
    }
}

// Depthwise (group == channels) deconvolution, gradient w.r.t. the input: the forward convolution of dY with the same filter
//   dX[n][iy][ix][c] = sum_{r,q} dY[n][iy*s - p + r][ix*s - p + q][c] * w[c][r][q]
__global__ __launch_bounds__(256) void deconv_dw_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                            int N, int H, int W, int C, int dx_cstride, int k, int stride, int pad, int OH,
                                                            int OW, int dy_cstride, int dy_coffset, int accumulate) {
    const long long total = (long long)N * H * W * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        long long pix = t / C;
        const int ix = (int)(pix % W);
        const long long row = pix / W;
        const int iy = (int)(row % H);
        const int n = (int)(row / H);
        const float* wc = w + (size_t)c * k * k;
        const float* dyb = dy + (size_t)n * OH * OW * dy_cstride + dy_coffset + c;
        float acc = 0.f;
        for (int r = 0; r < k; ++r) {
            const int oy = iy * stride - pad + r;
            if ((unsigned)oy >= (unsigned)OH) continue;
            for (int q = 0; q < k; ++q) {
                const int ox = ix * stride - pad + q;
                if ((unsigned)ox >= (unsigned)OW) continue;
                acc += dyb[((size_t)oy * OW + ox) * dy_cstride] * wc[r * k + q];
            }
        }
        float* d = dx + (size_t)pix * dx_cstride + c;
        *d = accumulate ? *d + acc : acc;
    }
}

// SoftmaxWithLoss (Caffe, legacy `normalize` flag), three passes so that any number of pixels reduces in a fixed order:
//   1. per pixel: p = softmax(x); loss -= log(max(p[label], FLT_MIN)); dx = p - onehot(label) (zero if ignored);
//      per-workgroup partial (loss, valid count) in double
//   2. one workgroup: fixed-order sum of the partials -> loss / denom, scale = weight / denom
//      (denom = valid count if normalize else N)
//   3. dx *= scale
constexpr int SML_MAX_BLOCKS = 1024;
__global__ __launch_bounds__(256) void softmax_loss_partial_kernel(const float* __restrict__ x, const float* __restrict__ label,
                                                                   float* __restrict__ dx, double* __restrict__ partial, long long pixels,
                                                                   int C, int x_cstride, int label_cstride, int has_ignore, int ignore_label) {
    __shared__ double sl[256], sc[256];
    double loss = 0.0, cnt = 0.0;
    for (long long pix = (long long)blockIdx.x * blockDim.x + threadIdx.x; pix < pixels; pix += (long long)gridDim.x * blockDim.x) {
        const float* xp = x + (size_t)pix * x_cstride;
        float m = xp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, xp[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) sum += expf(xp[c] - m);
        int lab = (int)label[(size_t)pix * label_cstride];
        const bool ignored = has_ignore && lab == ignore_label;
        lab = min(max(lab, 0), C - 1);      // Caffe DCHECKs the range; never index outside the pixel
        if (!ignored) {
            loss -= (double)logf(fmaxf(expf(xp[lab] - m) / sum, 1.175494351e-38f));
            cnt += 1.0;
        }
        if (dx) {
            float* dp = dx + (size_t)pix * x_cstride;
            for (int c = 0; c < C; ++c) dp[c] = ignored ? 0.f : expf(xp[c] - m) / sum - (c == lab ? 1.f : 0.f);
        }
    }
    sl[threadIdx.x] = loss;
    sc[threadIdx.x] = cnt;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            sl[threadIdx.x] += sl[threadIdx.x + o];
            sc[threadIdx.x] += sc[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = sl[0];
        partial[2 * blockIdx.x + 1] = sc[0];
    }
}

__global__ void softmax_loss_final_kernel(const double* __restrict__ partial, int nblocks, float* __restrict__ d_loss, float* __restrict__ d_scale,
                                          int N, int normalize, float weight) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double loss = 0.0, cnt = 0.0;
    for (int i = 0; i < nblocks; ++i) {
        loss += partial[2 * i];
        cnt += partial[2 * i + 1];
    }
    const double denom = normalize ? (cnt > 1.0 ? cnt : 1.0) : (double)N;
    *d_loss = (float)(loss / denom);
    *d_scale = (float)((double)weight / denom);
}

__global__ __launch_bounds__(256) void scale_view_kernel(float* __restrict__ x, const float* __restrict__ d_scale, long long pixels, int C,
                                                         int cstride) {
    const float sc = *d_scale;
    const long long total = pixels * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x)
        x[(size_t)(t / C) * cstride + (t % C)] *= sc;
}

// counter-based dropout mask (oracle/caffe_ref.py::dropout_hash): element index = NCHW linear index
__device__ __forceinline__ unsigned dropout_hash(unsigned index, unsigned seed) {
    unsigned x = index + seed * 0x9E3779B9u;
    x ^= x >> 16;
    x *= 0x7FEB352Du;
    x ^= x >> 15;
    x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int C, int H, int W,
                                                      int x_cstride, int x_coffset, int y_cstride, int y_coffset, unsigned thresh, float scale,
                                                      unsigned seed, unsigned index_offset) {
    const long long total = (long long)N * H * W * C;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        long long pix = t / C;
        const int w = (int)(pix % W);
        long long u = pix / W;
        const int h = (int)(u % H);
        const int n = (int)(u / H);
        const unsigned nchw = (unsigned)((((long long)n * C + c) * H + h) * W + w) + index_offset;
        const bool keep = dropout_hash(nchw, seed) >= thresh;
        y[(size_t)pix * y_cstride + y_coffset + c] = keep ? x[(size_t)pix * x_cstride + x_coffset + c] * scale : 0.f;
    }
}

// losses: one 1024-thread workgroup, fixed summation order -> reproducible
template <bool L1>
__global__ __launch_bounds__(1024) void loss_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ da,
                                                    float* __restrict__ loss, long long pixels, int C, int cstride, float inv_num, float weight) {
    __shared__ double part[1024];
    double s = 0.0;
    const long long total = pixels * C;
    for (long long t = threadIdx.x; t < total; t += 1024) {
        const size_t o = (size_t)(t / C) * cstride + (t % C);
        const float d = a[o] - b[o];
        s += L1 ? fabs((double)d) : (double)d * (double)d;
        if (da) da[o] = L1 ? ((d > 0.f) - (d < 0.f)) * weight * inv_num : d * weight * inv_num;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) part[threadIdx.x] += part[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = (float)(L1 ? part[0] * inv_num : part[0] * inv_num * 0.5);
}

// solver: segments of one flat parameter buffer share a launch; per-segment lr / decay multipliers
struct SolverSeg { unsigned long long offset, count; float lr_mult, decay_mult; };

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ hist,
                                                  const SolverSeg* __restrict__ segs, int nseg, float rate, float momentum, float decay,
                                                  float grad_scale) {
    const int s = blockIdx.y;
    if (s >= nseg) return;
    const SolverSeg seg = segs[s];
    if (seg.lr_mult == 0.f) return;
    const float lr = rate * seg.lr_mult, wd = decay * seg.decay_mult;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < seg.count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long o = seg.offset + i;
        const float gi = g[o] * grad_scale + wd * w[o];
        const float h = momentum * hist[o] + lr * gi;
        hist[o] = h;
        w[o] -= h;
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                   const SolverSeg* __restrict__ segs, int nseg, float rate_corr, float beta1, float beta2,
                                                   float delta, float decay, float grad_scale) {
    const int s = blockIdx.y;
    if (s >= nseg) return;
    const SolverSeg seg = segs[s];
    if (seg.lr_mult == 0.f) return;
    const float lr = rate_corr * seg.lr_mult, wd = decay * seg.decay_mult;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < seg.count; i += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long o = seg.offset + i;
        const float gi = g[o] * grad_scale + wd * w[o];
        const float mi = beta1 * m[o] + (1.f - beta1) * gi;
        const float vi = beta2 * v[o] + (1.f - beta2) * gi * gi;
        m[o] = mi;
        v[o] = vi;
        w[o] -= lr * mi / (sqrtf(vi) + delta);
    }
}

unsigned magic32(unsigned d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)((0x100000000ull + d - 1) / d); }

}  // namespace

extern "C" {

#ifdef FCN_WS_STAMPS
int fcn_debug_wgrad_stamps(void* d_buf, int cap) {
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps), &d_buf, sizeof(d_buf)));
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_ws_stamps_cap), &cap, sizeof(cap)));
    return 0;
}
#endif

struct WgPlan {
    int cfg, splits, rn, rk;      // splits: the most slabs any element has (workspace size)
    int cs[4];                    // role-split kernel: split count of region class 2 a + b
    int upc[4], count[4];         // ... MFMAs per step of the class's busiest wave; regions in the class
    int d[4], cps0;               // ... pixel runs (of cps0 chunks) per workgroup of the class
};

static int wgrad_forced_cfg() {
    const char* force = getenv("FCN_WGRAD_CFG");
    return force && force[0] >= '0' && force[0] <= '0' + kSplitCfg && !force[1] ? force[0] - '0' : -1;
}

static int wgrad_split_rounds_max() {      // FCN_WGRAD_SPLIT_ROUNDS: most workgroups per CU the planner may give a launch of the role-split kernel
    const char* e = getenv("FCN_WGRAD_SPLIT_ROUNDS");
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 16 ? v : 3;
}

static int device_cus() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) return v;
    return 256;
}

// cycles of one MFMA per step over a chunk; the least a chunk takes whatever its MFMAs (memory latency over the chunks in flight)
constexpr double kStepCycles = 64.0 * WS_STEPS, kChunkFloor = 3300.0 / (WS_NBUF - 2);

// Region classes of an rn x rk tiling of a Cout x K matrix: class 2 a + b, a / b = 1 for the last region along channels / k
// (usually a partial one).  Per class: how many regions, MFMAs per step of its busiest multiplying wave (the region's sub-tile
// pairs along k are dealt round-robin to four waves), cycles per 32-pixel chunk = max(MFMA time, staging time at 20 bytes per
// clock and CU) + 0.3 staging time (staging never overlaps perfectly, and the fabric is shared).
static double split_region_classes(int Cout, long long K, int rn, int rk, int* upc, int* count) {
    const int ns = cdiv(Cout, 32), ks = (int)cdiv(K, 32), tn = cdiv(ns, rn), tk = cdiv(ks, rk);
    double cost = 0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b) {
            const int vn = a ? ns - (tn - 1) * rn : rn, vk = b ? ks - (tk - 1) * rk : rk, c = 2 * a + b;
            count[c] = (a ? 1 : tn - 1) * (b ? 1 : tk - 1);
            const int kp = (vk + 1) / 2, pairs = vn * kp;
            int busiest = 0;      // exactly what the kernel deals: wave w takes pairs w, w + 4, ..; the last pair of a row is half when vk is odd
            for (int w = 0; w < 4 && w < pairs; ++w) {
                int m = 0;
                for (int pt = w; pt < pairs; pt += 4) m += ((pt % kp) * 2 + 1 < vk) ? 2 : 1;
                if (m > busiest) busiest = m;
            }
            upc[c] = busiest;
            const double mfma = busiest * kStepCycles > kChunkFloor ? busiest * kStepCycles : kChunkFloor, stage = (vn + vk) * 32.0 * WS_BP * 4 / 20.0;
            cost += count[c] * ((mfma > stage ? mfma : stage) + 0.3 * stage);
        }
    return cost;
}

// Region shape of the role-split kernel: every rn x rk (in 32-wide sub-tiles, rn <= 4, rk <= 8, rk even) is priced by
// split_region_classes; the cheapest wins, fewer regions on a tie.
static void plan_split_region(int Cout, long long K, WgPlan* pl) {
    const int ns = cdiv(Cout, 32), ks = (int)cdiv(K, 32);
    double best = 1e300;
    int btiles = 1 << 30;
    pl->rn = pl->rk = 1;
    for (int rn = 1; rn <= 4; ++rn)
        for (int rk = 1; rk <= 8; ++rk) {
            if (rk > 1 && (rk & 1)) continue;      // (pairs along k: an odd width would leave half a pair in every region)
            int upc[4], count[4];
            const double cost = split_region_classes(Cout, K, rn, rk, upc, count);
            const int tiles = cdiv(ns, rn) * cdiv(ks, rk);
            if (cost < best - 1e-9 || (cost < best + 1e-9 && tiles < btiles)) { best = cost; pl->rn = rn; pl->rk = rk; btiles = tiles; }
        }
    split_region_classes(Cout, K, pl->rn, pl->rk, pl->upc, pl->count);
}

// Pixel splits of the role-split kernel for the n problems of one launch.  One workgroup is resident per CU, so a launch runs in
// ROUNDS of `cus` workgroups and a round lasts as long as its longest workgroup: every region class gets the split count that
// brings its workgroups to a common duration tau (= the class's full-depth duration / tau), tau is the smallest one that fits
// r rounds, and r is the round count with the least estimated time (r x (tau + a workgroup's fixed cost) + the reduction's
// reads of the partial slabs).
static void plan_split_counts(const fcn_conv_desc* ds, int n, WgPlan* pls) {
    const int cus = device_cus();
    int chunks[kMaxWgGroup], cap[kMaxWgGroup];
    double cyc[kMaxWgGroup][4], cyc_max[kMaxWgGroup];      // shader cycles per chunk of a class's workgroup / of the problem's slowest class
    double wmax = 0;
    for (int i = 0; i < n; ++i) {
        chunks[i] = cdiv((long long)ds[i].N * ds[i].OH * ds[i].OW, WS_BP);
        cap[i] = chunks[i] / 4 < 1 ? 1 : chunks[i] / 4 > 256 ? 256 : chunks[i] / 4;      // (a few chunks per workgroup, at least)
        cyc_max[i] = 0;
        for (int c = 0; c < 4; ++c) {
            // a chunk takes its MFMAs or, with one chunk in flight behind the one being multiplied, a trip to memory under load
            // (tools/wgrad_timeline.py: 3100-3300 cycles per 32-pixel chunk, one in flight, in workgroups with one or two MFMAs per step)
            cyc[i][c] = pls[i].upc[c] * kStepCycles > kChunkFloor ? pls[i].upc[c] * kStepCycles : kChunkFloor;
            if (pls[i].count[c] && cyc[i][c] > cyc_max[i]) cyc_max[i] = cyc[i][c];
        }
        // a lighter class takes d whole runs of the heaviest class's pixels per workgroup (d <= 8: the kernel divides by multiply-shift)
        for (int c = 0; c < 4; ++c) {
            const int d = pls[i].count[c] ? (int)(cyc_max[i] / cyc[i][c]) : 1;
            pls[i].d[c] = d < 1 ? 1 : d > 8 ? 8 : d;
        }
        if ((double)chunks[i] * cyc_max[i] > wmax) wmax = (double)chunks[i] * cyc_max[i];
    }
    const double fixed = 15000.0, red_bytes_per_cycle = 1500.0;
    double best = 1e300;
    int best_s0[kMaxWgGroup], s0[kMaxWgGroup];
    for (int i = 0; i < kMaxWgGroup; ++i) best_s0[i] = s0[i] = 1;
    auto class_splits = [&](int i, int c, int sp0) { return cdiv(chunks[i], pls[i].d[c] * cdiv(chunks[i], sp0)); };
    auto count_wgs = [&](double tau) {
        long long wgs = 0;
        for (int i = 0; i < n; ++i) {
            int sp = (int)ceil((double)chunks[i] * cyc_max[i] / tau);
            sp = sp < 1 ? 1 : sp > cap[i] ? cap[i] : sp;
            s0[i] = sp;
            for (int c = 0; c < 4; ++c) wgs += (long long)pls[i].count[c] * class_splits(i, c, sp);
        }
        return wgs;
    };
    for (int r = 1; r <= wgrad_split_rounds_max(); ++r) {
        double lo = kStepCycles, hi = wmax;      // smallest tau whose split counts fit r rounds
        if (count_wgs(hi) > (long long)cus * r) continue;      // (more regions than r rounds hold even unsplit)
        for (int it = 0; it < 40; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (count_wgs(mid) <= (long long)cus * r) hi = mid; else lo = mid;
        }
        count_wgs(hi);
        double tau = 0, red = 0;
        for (int i = 0; i < n; ++i) {
            const long long K = (long long)ds[i].kh * ds[i].kw * ds[i].Cin;
            const int ns = cdiv(ds[i].Cout, 32), ks = (int)cdiv(K, 32);
            for (int c = 0; c < 4; ++c) {
                if (!pls[i].count[c]) continue;
                const double t = (double)pls[i].d[c] * cdiv(chunks[i], s0[i]) * cyc[i][c];
                if (t > tau) tau = t;
                const int vn = (c >> 1) ? ns - (cdiv(ns, pls[i].rn) - 1) * pls[i].rn : pls[i].rn, vk = (c & 1) ? ks - (cdiv(ks, pls[i].rk) - 1) * pls[i].rk : pls[i].rk;
                red += (double)class_splits(i, c, s0[i]) * pls[i].count[c] * vn * vk * 4096.0 / red_bytes_per_cycle;
            }
        }
        const double est = r * (tau + fixed) + red;
        if (est < best) {
            best = est;
            for (int i = 0; i < n; ++i) best_s0[i] = s0[i];
        }
    }
    for (int i = 0; i < n; ++i) {
        pls[i].splits = 1;
        pls[i].cps0 = cdiv(chunks[i], best_s0[i]);
        for (int c = 0; c < 4; ++c) {
            pls[i].cs[c] = class_splits(i, c, best_s0[i]);
            if (pls[i].count[c] && pls[i].cs[c] > pls[i].splits) pls[i].splits = pls[i].cs[c];
        }
    }
    if (getenv("FCN_WGRAD_PLAN_LOG"))
        for (int i = 0; i < n; ++i)
            fprintf(stderr, "wgrad plan: Cout %d K %lld chunks %d -> region %d x %d, runs of %d chunks; regions x splits (MFMAs/step, runs per workgroup) by class: "
                    "%d x %d (%d, %d), %d x %d (%d, %d), %d x %d (%d, %d), %d x %d (%d, %d)\n",
                    ds[i].Cout, (long long)ds[i].kh * ds[i].kw * ds[i].Cin, chunks[i], pls[i].rn, pls[i].rk, pls[i].cps0, pls[i].count[0], pls[i].cs[0], pls[i].upc[0],
                    pls[i].d[0], pls[i].count[1], pls[i].cs[1], pls[i].upc[1], pls[i].d[1], pls[i].count[2], pls[i].cs[2], pls[i].upc[2], pls[i].d[2], pls[i].count[3],
                    pls[i].cs[3], pls[i].upc[3], pls[i].d[3]);
}

// tile shape with the least padded work per unit of sustained rate, then enough pixel splits to fill the chip
static WgPlan plan_wgrad(const fcn_conv_desc* d, int cfg_request = -1) {
    const long long M = (long long)d->N * d->OH * d->OW, K = (long long)d->kh * d->kw * d->Cin;
    WgPlan pl = {};
    pl.splits = 1;
    const int force = cfg_request >= 0 && cfg_request <= kSplitCfg ? cfg_request : wgrad_forced_cfg();
    const int chunks = cdiv(M, WG_BP);
    if (force == kSplitCfg) {
        pl.cfg = kSplitCfg;
        plan_split_region(d->Cout, K, &pl);
        plan_split_counts(d, 1, &pl);
        return pl;
    } else {
        if (force >= 0) {
            pl.cfg = force;
        } else {
            double best_cost = 1e300;
            for (int c = 0; c < kNumWgCfg; ++c) {
                const double padded = (double)cdiv(d->Cout, kWgShapes[c].bn) * kWgShapes[c].bn * cdiv(K, kWgShapes[c].bk) * kWgShapes[c].bk;
                const double cost = padded / kWgShapes[c].eff;
                if (cost < best_cost) { best_cost = cost; pl.cfg = c; }
            }
        }
        const int tiles = cdiv(d->Cout, kWgShapes[pl.cfg].bn) * cdiv(K, kWgShapes[pl.cfg].bk);
        pl.splits = cdiv(kWgShapes[pl.cfg].nw == 8 ? 512 : 1024, tiles);
        if (pl.splits > chunks) pl.splits = chunks;
    }
    if (pl.splits < 1) pl.splits = 1;
    if (pl.splits > 256) pl.splits = 256;
    for (int c = 0; c < 4; ++c) pl.cs[c] = pl.splits;
    return pl;
}

int fcn_conv2d_wgrad_num_configs(void) { return kSplitCfg + 1; }
int fcn_conv2d_wgrad_split_config(void) { return kSplitCfg; }

size_t fcn_conv2d_wgrad_workspace_floats_cfg(const fcn_conv_desc* d, int cfg_request, int* h_splits) {
    if (!d || d->Cout <= 0) return 0;
    const long long K = (long long)d->kh * d->kw * d->Cin;
    const WgPlan pl = plan_wgrad(d, cfg_request);
    if (h_splits) *h_splits = pl.splits;
    return (size_t)pl.splits * (((size_t)d->Cout * K + d->Cout + 3) / 4 * 4);
}
size_t fcn_conv2d_wgrad_workspace_floats(const fcn_conv_desc* d, int* h_splits) { return fcn_conv2d_wgrad_workspace_floats_cfg(d, -1, h_splits); }

static int wgrad_validate(const fcn_conv_desc* d) {
    FCN_REQUIRE(d && d->x && d->y, FCN_E_ARG, "wgrad: null");
    FCN_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 && d->stride > 0, FCN_E_ARG,
                "wgrad: non-positive extent");
    FCN_REQUIRE(d->Cin % 4 == 0 && d->x_cstride % 4 == 0 && d->y_cstride % 4 == 0 && d->y_coffset % 4 == 0, FCN_E_ALIGN,
                "wgrad: channel counts / strides / offsets must be multiples of 4");
    FCN_REQUIRE(d->OH == (d->H + 2 * d->pad - d->kh) / d->stride + 1 && d->OW == (d->W + 2 * d->pad - d->kw) / d->stride + 1, FCN_E_ARG,
                "wgrad: OH/OW mismatch");
    FCN_REQUIRE((long long)d->N * d->OH * d->OW < (1ll << 31) && (long long)d->N * d->H * d->W * d->x_cstride < (1ll << 31), FCN_E_UNSUPPORTED,
                "wgrad: tensor too large");
    FCN_REQUIRE(d->y_cstride >= d->y_coffset + d->Cout, FCN_E_ARG, "wgrad: gradient slice exceeds its channel stride");
    FCN_REQUIRE((long long)d->N * d->OH * d->OW * (d->OW > d->OH ? d->OW : d->OH) < (1ll << 32), FCN_E_UNSUPPORTED,
                "wgrad: N*OH*OW*max(OH,OW) must stay below 2^32 (multiply-high pixel decode): split the batch");
    FCN_REQUIRE((long long)d->N * d->OH * d->OW * d->y_cstride * 4 < (1ll << 31) && (long long)d->N * d->H * d->W * d->x_cstride * 4 < (1ll << 31),
                FCN_E_UNSUPPORTED, "wgrad: x and dY must each stay below 2 GiB (32-bit buffer offsets): split the batch");
    return 0;
}

static void wgrad_fill(WgradP& p, const fcn_conv_desc* d, const WgPlan& pl, float* slabs, bool with_bias, const float* zp) {
    const int splits = pl.splits;
    p.x = d->x;
    p.dy = d->y + d->y_coffset;
    p.zero_page = zp;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.x_cstride = d->x_cstride;
    p.Cout = d->Cout; p.kh = d->kh; p.kw = d->kw; p.pad = d->pad; p.stride = d->stride; p.OH = d->OH; p.OW = d->OW;
    p.dy_cstride = d->y_cstride;
    p.M = d->N * d->OH * d->OW;
    p.K = d->kh * d->kw * d->Cin;
    p.rn = pl.rn; p.rk = pl.rk;
    p.tiles_n = cdiv(p.Cout, pl.cfg == kSplitCfg ? 32 * pl.rn : kWgShapes[pl.cfg].bn);
    p.tiles_k = cdiv(p.K, pl.cfg == kSplitCfg ? 32 * pl.rk : kWgShapes[pl.cfg].bk);
    p.splits = splits;
    p.chunks_per_split = cdiv(cdiv(p.M, WG_BP), splits);
    p.wgs = p.runs = 0;
    p.cps0 = pl.cps0 > 0 ? pl.cps0 : 1;
    for (int c = 0; c < 4; ++c) {
        p.cls_splits[c] = pl.cs[c] > 0 ? pl.cs[c] : 1;
        p.cls_d[c] = pl.d[c] > 0 ? pl.d[c] : 1;
        p.cls_dmagic[c] = (65536 + p.cls_d[c] - 1) / p.cls_d[c];
        p.cls_count[c] = pl.cfg == kSplitCfg ? pl.count[c] : 0;
        p.wgs += p.cls_count[c] * p.cls_splits[c];
        if (p.cls_count[c] && p.cls_splits[c] * p.cls_d[c] > p.runs) p.runs = p.cls_splits[c] * p.cls_d[c];
    }
    p.ow_magic = p.OW > 1 ? magic32((unsigned)p.OW) : 0u;
    p.oh_magic = p.OH > 1 ? magic32((unsigned)p.OH) : 0u;
    p.kw_magic = (65536 + p.kw - 1) / p.kw;
    p.dw_part = slabs;
    p.slab_floats = (p.Cout * p.K + p.Cout + 3) / 4 * 4;      // slabs stay 16-byte aligned for the reduction's float4 loads
    p.db_part = with_bias ? slabs + (size_t)p.Cout * p.K : nullptr;
}

// which slabs the reduction reads per element (weights then, if `with_bias_tail`, the bias sums right behind them; bias_only: the bias sums alone)
static RedSplits red_splits_of(const WgradP& p, const WgPlan& pl, bool bias_only) {
    if (pl.cfg != kSplitCfg) return red_uniform(p.splits);
    RedSplits r;
    r.wcount = bias_only ? 0 : (long long)p.Cout * p.K;
    r.K = p.K;
    r.n_last0 = (p.tiles_n - 1) * 32 * p.rn;
    r.k_last0 = (p.tiles_k - 1) * 32 * p.rk;
    r.b_bias = p.tiles_k == 1;      // (the bias sums come from the workgroups of the first k-region)
    r.uniform = 0;
    for (int c = 0; c < 4; ++c) r.s[c] = p.cls_splits[c];
    return r;
}
static int split_wgs(const WgradP& p) { return p.wgs; }

// dW (OHWI, [Cout][kh][kw][Cin]) and db from the layer input x and the output gradient passed in desc->y / y_cstride /
// y_coffset (desc->w and desc->bias are ignored).  d_workspace: fcn_conv2d_wgrad_workspace_floats() floats.
int fcn_conv2d_wgrad_f32(const fcn_conv_desc* d, float* dw, float* db, float* d_workspace, fcn_stream_t s) {
    return fcn_conv2d_wgrad_cfg_f32(d, dw, db, d_workspace, -1, s);
}

int fcn_conv2d_wgrad_cfg_f32(const fcn_conv_desc* d, float* dw, float* db, float* d_workspace, int cfg_request, fcn_stream_t s) {
    FCN_REQUIRE(d && dw && d_workspace, FCN_E_ARG, "wgrad: null");
    FCN_REQUIRE(cfg_request >= -1 && cfg_request <= kSplitCfg, FCN_E_ARG, "wgrad: configuration %d (have -1 .. %d)", cfg_request, kSplitCfg);
    int rc = wgrad_validate(d);
    if (rc) return rc;
    const float* zp = zero_page_for_current_device(&rc);
    if (rc) return rc;
    const WgPlan pl = plan_wgrad(d, cfg_request);
    const int splits = pl.splits;
    WgradP p;
    wgrad_fill(p, d, pl, d_workspace, db != nullptr, zp);
    hipStream_t st = as_stream(s);
    if (pl.cfg == kSplitCfg) {
        WgradGroupArgs ga;
        ga.n = 1;
        for (int i = 0; i < kMaxWgGroup; ++i) {
            ga.wg_end[i] = split_wgs(p);
            ga.p[i] = p;
        }
        hipLaunchKernelGGL(conv_wgrad_split_kernel, dim3(ga.wg_end[0]), dim3(512), 0, st, ga);
    } else
    switch (pl.cfg) {
#define X(I, A, B, C_, D, E)                                                                                                              \
    case I:                                                                                                                               \
        hipLaunchKernelGGL((conv_wgrad_kernel<A, B, C_, D, E>), dim3(p.tiles_n * p.tiles_k * splits), dim3(64 * C_ * D), 0, st, p); \
        break;
        FCN_WGRAD_CONFIGS(X)
#undef X
    }
    const size_t cnt = (size_t)p.Cout * p.K;
    const size_t stride = (size_t)p.slab_floats;
    // the solver keeps a layer's bias gradient right behind its weight gradient: one reduction then covers both
    const bool together = db && db == dw + cnt;
    const size_t cnt1 = together ? cnt + p.Cout : cnt;
    const int red_blocks = (int)((cnt1 + RED_PER_BLOCK - 1) / RED_PER_BLOCK < 4096 ? (cnt1 + RED_PER_BLOCK - 1) / RED_PER_BLOCK : 4096);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(red_blocks), dim3(RED_THREADS), 0, st, p.dw_part, dw, cnt1, red_splits_of(p, pl, false), stride);
    if (db && !together)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv(p.Cout, RED_PER_BLOCK)), dim3(RED_THREADS), 0, st, p.db_part, db, (size_t)p.Cout,
                           red_splits_of(p, pl, true), stride);
    FCN_LAUNCH_CHECK("conv_wgrad");
    return 0;
}

// One tile shape for the whole group (the one with the least padded work over all problems), pixel splits so that the
// launch has ~1024-1536 workgroups in total.  The role-split kernel takes a region shape per problem instead.
static void plan_wgrad_group(const fcn_conv_desc* ds, int n, WgPlan* pls, int cfg_request = -1) {
    int best = 0;
    const int force = cfg_request >= 0 && cfg_request <= kSplitCfg ? cfg_request : wgrad_forced_cfg();
    if (force >= 0) {
        best = force;
    } else {
        double best_cost = 1e300;
        for (int c = 0; c < kNumWgCfg; ++c) {
            double cost = 0;
            for (int i = 0; i < n; ++i) {
                const long long K = (long long)ds[i].kh * ds[i].kw * ds[i].Cin, M = (long long)ds[i].N * ds[i].OH * ds[i].OW;
                cost += (double)cdiv(ds[i].Cout, kWgShapes[c].bn) * kWgShapes[c].bn * cdiv(K, kWgShapes[c].bk) * kWgShapes[c].bk * (double)M /
                        kWgShapes[c].eff;
            }
            if (cost < best_cost) { best_cost = cost; best = c; }
        }
    }
    long long tiles = 0;
    for (int i = 0; i < n; ++i) {
        const long long K = (long long)ds[i].kh * ds[i].kw * ds[i].Cin;
        pls[i] = WgPlan{};
        pls[i].cfg = best;
        pls[i].splits = 1;
        if (best == kSplitCfg) plan_split_region(ds[i].Cout, K, &pls[i]);
        else tiles += (long long)cdiv(ds[i].Cout, kWgShapes[best].bn) * cdiv(K, kWgShapes[best].bk);
    }
    if (best == kSplitCfg) {
        plan_split_counts(ds, n, pls);
        return;
    }
    for (int i = 0; i < n; ++i) {
        const int chunks = cdiv((long long)ds[i].N * ds[i].OH * ds[i].OW, WG_BP);
        int sp = cdiv(1024, tiles);
        const int cap = chunks;
        if (sp > cap) sp = cap;
        if (sp < 1) sp = 1;
        if (sp > 256) sp = 256;
        pls[i].splits = sp;
        for (int c = 0; c < 4; ++c) pls[i].cs[c] = sp;
    }
}

size_t fcn_conv2d_wgrad_group_workspace_floats(const fcn_conv_desc* ds, int n) { return fcn_conv2d_wgrad_group_workspace_floats_cfg(ds, n, -1); }

size_t fcn_conv2d_wgrad_group_workspace_floats_cfg(const fcn_conv_desc* ds, int n, int cfg_request) {
    if (!ds || n <= 0 || n > kMaxWgGroup) return 0;
    WgPlan pls[kMaxWgGroup];
    plan_wgrad_group(ds, n, pls, cfg_request);
    size_t total = 0;
    for (int i = 0; i < n; ++i) total += (size_t)pls[i].splits * (((size_t)ds[i].Cout * ds[i].kh * ds[i].kw * ds[i].Cin + ds[i].Cout + 3) / 4 * 4) + 4;
    return total;
}

// n <= 4 layers in one launch + one grouped reduction.  dbs[i] may be NULL; when dbs[i] == dws[i] + Cout*K (the solver's
// layout) weight and bias partials are reduced together.
int fcn_conv2d_wgrad_group_f32(const fcn_conv_desc* ds, float* const* dws, float* const* dbs, int n, float* d_workspace, fcn_stream_t s) {
    return fcn_conv2d_wgrad_group_cfg_f32(ds, dws, dbs, n, d_workspace, -1, s);
}

int fcn_conv2d_wgrad_group_cfg_f32(const fcn_conv_desc* ds, float* const* dws, float* const* dbs, int n, float* d_workspace, int cfg_request,
                                   fcn_stream_t s) {
    FCN_REQUIRE(ds && dws && dbs && d_workspace && n > 0 && n <= kMaxWgGroup, FCN_E_ARG, "wgrad group: need 1..%d problems", kMaxWgGroup);
    FCN_REQUIRE(cfg_request >= -1 && cfg_request <= kSplitCfg, FCN_E_ARG, "wgrad group: configuration %d (have -1 .. %d)", cfg_request, kSplitCfg);
    int rc = 0;
    for (int i = 0; i < n; ++i) {
        rc = wgrad_validate(&ds[i]);
        if (rc) return rc;
        FCN_REQUIRE(dws[i], FCN_E_ARG, "wgrad group: null dw");
    }
    const float* zp = zero_page_for_current_device(&rc);
    if (rc) return rc;
    WgPlan pls[kMaxWgGroup];
    plan_wgrad_group(ds, n, pls, cfg_request);
    const int cfg = pls[0].cfg;
    WgradGroupArgs ga;
    ReduceGroupArgs ra;
    ga.n = ra.n = n;
    float* slabs = d_workspace;
    int wgs = 0, blks = 0, extra = 0;
    struct Extra { const float* parts; float* out; size_t count, stride; RedSplits rs; } extras[kMaxWgGroup];
    for (int i = 0; i < kMaxWgGroup; ++i) {
        const int j = i < n ? i : n - 1;
        if (i < n) {
            wgrad_fill(ga.p[i], &ds[i], pls[i], slabs, dbs[i] != nullptr, zp);
            const WgradP& p = ga.p[i];
            wgs += cfg == kSplitCfg ? split_wgs(p) : p.tiles_n * p.tiles_k * p.splits;
            const size_t cnt = (size_t)p.Cout * p.K;
            const bool together = dbs[i] && dbs[i] == dws[i] + cnt;
            ra.parts[i] = p.dw_part;
            ra.out[i] = dws[i];
            ra.count[i] = together ? cnt + p.Cout : cnt;
            ra.stride[i] = (unsigned long long)p.slab_floats;
            ra.rs[i] = red_splits_of(p, pls[i], false);
            blks += (int)((ra.count[i] + RED_PER_BLOCK - 1) / RED_PER_BLOCK);
            if (dbs[i] && !together) extras[extra++] = Extra{p.db_part, dbs[i], (size_t)p.Cout, (size_t)p.slab_floats, red_splits_of(p, pls[i], true)};
            slabs += ((size_t)p.splits * p.slab_floats + 3) / 4 * 4;
        } else {
            ga.p[i] = ga.p[j];
            ra.parts[i] = ra.parts[j]; ra.out[i] = ra.out[j]; ra.count[i] = ra.count[j]; ra.stride[i] = ra.stride[j]; ra.rs[i] = ra.rs[j];
        }
        ga.wg_end[i] = wgs;
        ra.blk_end[i] = blks;
    }
    hipStream_t st = as_stream(s);
    if (cfg == kSplitCfg) hipLaunchKernelGGL(conv_wgrad_split_kernel, dim3(wgs), dim3(512), 0, st, ga);
    else
    switch (cfg) {
#define X(I, A, B, C_, D, E)                                                                                          \
    case I:                                                                                                           \
        hipLaunchKernelGGL((conv_wgrad_group_kernel<A, B, C_, D, E>), dim3(wgs), dim3(64 * C_ * D), 0, st, ga); \
        break;
        FCN_WGRAD_CONFIGS(X)
#undef X
    }
    hipLaunchKernelGGL(reduce_partials_group_kernel, dim3(blks), dim3(RED_THREADS), 0, st, ra);
    for (int e = 0; e < extra; ++e)
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(cdiv((long long)extras[e].count, RED_PER_BLOCK)), dim3(RED_THREADS), 0, st, extras[e].parts,
                           extras[e].out, extras[e].count, extras[e].rs, extras[e].stride);
    FCN_LAUNCH_CHECK("conv_wgrad_group");
    return 0;
}

int fcn_conv_weights_flip_f32(const float* w, float* wt, int Cout, int kh, int kw, int Cin, int Cin4, int Cout4, fcn_stream_t s) {
    FCN_REQUIRE(w && wt && Cout > 0 && kh > 0 && kw > 0 && Cin > 0 && Cin4 >= Cin && Cout4 >= Cout, FCN_E_ARG, "weights_flip: bad args");
    const long long total = (long long)Cin * kh * kw * Cout4;
    hipLaunchKernelGGL(weights_flip_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, as_stream(s), w, wt, Cout, kh, kw, Cin, Cin4, Cout4);
    FCN_LAUNCH_CHECK("weights_flip");
    return 0;
}

int fcn_conv_weights_flip_batch_f32(const float* w_base, float* wt_base, const fcn_flip_seg* d_segs, int nseg, fcn_stream_t s) {
    FCN_REQUIRE(w_base && wt_base && d_segs && nseg > 0 && nseg <= 65535, FCN_E_ARG, "weights_flip_batch: bad args");
    hipLaunchKernelGGL(weights_flip_batch_kernel, dim3(96, nseg), dim3(256), 0, as_stream(s), w_base, wt_base, d_segs);
    FCN_LAUNCH_CHECK("weights_flip_batch");
    return 0;
}

int fcn_relu_bwd_f32(const float* dy, const float* y, float* dx, int pixels, int C, int cstride, fcn_stream_t s) {
    FCN_REQUIRE(dy && y && dx && pixels > 0 && C > 0 && cstride >= C, FCN_E_ARG, "relu_bwd: bad args");
    hipStream_t st = as_stream(s);
    if (cstride == C)
        hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, st, dy, y, dx, (size_t)pixels * C);
    else
        hipLaunchKernelGGL(relu_bwd_view_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, st, dy, y, dx, (long long)pixels, C,
                           cstride);
    FCN_LAUNCH_CHECK("relu_bwd");
    return 0;
}

int fcn_sigmoid_bwd_f32(const float* y, const float* dy, float* dx, size_t count, int accumulate, fcn_stream_t s) {
    FCN_REQUIRE(y && dy && dx, FCN_E_ARG, "sigmoid_bwd: null");
    if (!count) return 0;
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(stream_grid((long long)count, 256)), dim3(256), 0, as_stream(s), y, dy, dx, count, accumulate);
    FCN_LAUNCH_CHECK("sigmoid_bwd");
    return 0;
}

int fcn_maxpool_bwd_mask_f32(const float* dy, const int32_t* idx, float* dx, int N, int H, int W, int C, int dx_cstride, int dx_coffset, int k,
                             int stride, int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate, const float* relu_y,
                             int relu_y_cstride, int relu_y_coffset, fcn_stream_t s) {
    FCN_REQUIRE(dy && idx && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && OH > 0 && OW > 0, FCN_E_ARG, "maxpool_bwd: bad args");
    FCN_REQUIRE(dx_cstride >= dx_coffset + C && dy_cstride >= dy_coffset + C, FCN_E_ARG, "maxpool_bwd: channel slice out of range");
    FCN_REQUIRE(!relu_y || (relu_y_coffset >= 0 && relu_y_cstride >= relu_y_coffset + C), FCN_E_ARG, "maxpool_bwd: mask slice out of range");
    const bool v4 = C % 4 == 0 && dx_cstride % 4 == 0 && dx_coffset % 4 == 0 && dy_cstride % 4 == 0 && dy_coffset % 4 == 0 &&
                    (long long)N * H * W * (C / 4) < (1ll << 31) && (((uintptr_t)dy | (uintptr_t)idx | (uintptr_t)dx) & 15) == 0 &&
                    (!relu_y || (relu_y_cstride % 4 == 0 && relu_y_coffset % 4 == 0 && ((uintptr_t)relu_y & 15) == 0));
    const int cgroups = (C + 31) / 32;
    const long long gx = (long long)cgroups * ((W + 7) / 8);
    if (v4 && gx < (1ll << 31) && (H + 7) / 8 <= 65535 && N <= 65535)
        hipLaunchKernelGGL(maxpool_bwd_patch_kernel, dim3((unsigned)gx, (H + 7) / 8, N), dim3(512), 0, as_stream(s), dy, idx, dx, H, W, C, dx_cstride,
                           dx_coffset, k, stride, pad, OH, OW, dy_cstride, dy_coffset, accumulate, relu_y, relu_y_cstride, relu_y_coffset, cgroups);
    else if (v4)
        hipLaunchKernelGGL(maxpool_bwd_v4_kernel, dim3(stream_grid((long long)N * H * W * (C / 4), 256)), dim3(256), 0, as_stream(s), dy, idx, dx, N, H,
                           W, C, dx_cstride, dx_coffset, k, stride, pad, OH, OW, dy_cstride, dy_coffset, accumulate, relu_y, relu_y_cstride,
                           relu_y_coffset);
    else
        hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid((long long)N * H * W * C, 256)), dim3(256), 0, as_stream(s), dy, idx, dx, N, H, W, C,
                           dx_cstride, dx_coffset, k, stride, pad, OH, OW, dy_cstride, dy_coffset, accumulate, relu_y, relu_y_cstride, relu_y_coffset);
    FCN_LAUNCH_CHECK("maxpool_bwd");
    return 0;
}

int fcn_maxpool_bwd_f32(const float* dy, const int32_t* idx, float* dx, int N, int H, int W, int C, int dx_cstride, int dx_coffset, int k,
                        int stride, int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate, fcn_stream_t s) {
    return fcn_maxpool_bwd_mask_f32(dy, idx, dx, N, H, W, C, dx_cstride, dx_coffset, k, stride, pad, OH, OW, dy_cstride, dy_coffset, accumulate,
                                    nullptr, 0, 0, s);
}

int fcn_deconv_depthwise_bwd_f32(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int dx_cstride, int k, int stride,
                                 int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate, fcn_stream_t s) {
    FCN_REQUIRE(dy && w && dx && N > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0, FCN_E_ARG, "deconv_bwd: bad args");
    FCN_REQUIRE(OH == stride * (H - 1) + k - 2 * pad && OW == stride * (W - 1) + k - 2 * pad, FCN_E_ARG,
                "deconv_bwd: OH/OW do not match s(H-1)+k-2p");
    FCN_REQUIRE(dx_cstride >= C && dy_coffset >= 0 && dy_cstride >= dy_coffset + C, FCN_E_ARG, "deconv_bwd: channel slice out of range");
    hipLaunchKernelGGL(deconv_dw_bwd_kernel, dim3(stream_grid((long long)N * H * W * C, 256)), dim3(256), 0, as_stream(s), dy, w, dx, N, H, W, C,
                       dx_cstride, k, stride, pad, OH, OW, dy_cstride, dy_coffset, accumulate);
    FCN_LAUNCH_CHECK("deconv_depthwise_bwd");
    return 0;
}

size_t fcn_softmax_loss_workspace_bytes(void) { return (size_t)SML_MAX_BLOCKS * 2 * sizeof(double) + 64; }

int fcn_softmax_loss_f32(const float* x, const float* label, float* dx, float* d_loss, int N, int pixels, int C, int x_cstride,
                         int label_cstride, int normalize, int has_ignore, int ignore_label, float weight, void* d_workspace, fcn_stream_t s) {
    FCN_REQUIRE(x && label && d_loss && d_workspace && N > 0 && pixels > 0 && C > 0 && x_cstride >= C && label_cstride >= 1, FCN_E_ARG,
                "softmax_loss: bad args");
    FCN_REQUIRE(((uintptr_t)d_workspace & 7) == 0, FCN_E_ALIGN, "softmax_loss: workspace must be 8-byte aligned");
    hipStream_t st = as_stream(s);
    int blocks = cdiv(pixels, 256);
    if (blocks > SML_MAX_BLOCKS) blocks = SML_MAX_BLOCKS;
    double* partial = reinterpret_cast<double*>(d_workspace);
    float* d_scale = reinterpret_cast<float*>(partial + (size_t)SML_MAX_BLOCKS * 2);
    hipLaunchKernelGGL(softmax_loss_partial_kernel, dim3(blocks), dim3(256), 0, st, x, label, dx, partial, (long long)pixels, C, x_cstride,
                       label_cstride, has_ignore, ignore_label);
    hipLaunchKernelGGL(softmax_loss_final_kernel, dim3(1), dim3(64), 0, st, partial, blocks, d_loss, d_scale, N, normalize, weight);
    if (dx) hipLaunchKernelGGL(scale_view_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, st, dx, d_scale, (long long)pixels, C,
                               x_cstride);
    FCN_LAUNCH_CHECK("softmax_loss");
    return 0;
}

int fcn_lrn_bwd_f32(const float* x, const float* y, const float* scale, const float* dy, float* dx, int pixels, int C, int x_cstride,
                    int y_cstride, int local_size, float alpha, float beta, int accumulate, fcn_stream_t s) {
    FCN_REQUIRE(x && y && scale && dy && dx && pixels > 0 && C > 0 && local_size > 0, FCN_E_ARG, "lrn_bwd: bad args");
    const bool fast = local_size == 5 && C % 4 == 0 && x_cstride % 4 == 0 && y_cstride % 4 == 0 &&
                      (((uintptr_t)x | (uintptr_t)y | (uintptr_t)scale | (uintptr_t)dy | (uintptr_t)dx) & 15) == 0;
    if (fast)
        hipLaunchKernelGGL(lrn5_bwd_kernel, dim3(stream_grid((long long)pixels * (C / 4), 256)), dim3(256), 0, as_stream(s), x, y, scale, dy, dx,
                           (long long)pixels, C, x_cstride, y_cstride, 2.f * alpha * beta / (float)local_size, beta, accumulate);
    else
        hipLaunchKernelGGL(lrn_bwd_kernel, dim3(stream_grid((long long)pixels * C, 256)), dim3(256), 0, as_stream(s), x, y, scale, dy, dx,
                           (long long)pixels, C, x_cstride, y_cstride, local_size, 2.f * alpha * beta / (float)local_size, beta, accumulate);
    FCN_LAUNCH_CHECK("lrn_bwd");
    return 0;
}

// y = x * mask * 1/(1-ratio) with the counter-based mask of seed `seed`; run on activations (forward) or gradients (backward)
int fcn_dropout_f32(const float* x, float* y, int N, int C, int H, int W, int x_cstride, int x_coffset, int y_cstride, int y_coffset,
                    float ratio, unsigned seed, unsigned index_offset, fcn_stream_t s) {
    FCN_REQUIRE(x && y && N > 0 && C > 0 && H > 0 && W > 0 && ratio >= 0.f && ratio < 1.f, FCN_E_ARG, "dropout: bad args");
    FCN_REQUIRE((long long)N * C * H * W < (1ll << 32), FCN_E_UNSUPPORTED, "dropout: blob too large for the 32-bit counter");
    double t = (double)ratio * 4294967296.0;
    const unsigned thresh = t >= 4294967295.0 ? 4294967295u : (unsigned)t;
    hipLaunchKernelGGL(dropout_kernel, dim3(stream_grid((long long)N * C * H * W, 256)), dim3(256), 0, as_stream(s), x, y, N, C, H, W, x_cstride,
                       x_coffset, y_cstride, y_coffset, thresh, 1.f / (1.f - ratio), seed, index_offset);
    FCN_LAUNCH_CHECK("dropout");
    return 0;
}

// kind 0: L1Loss (NVIDIA Caffe): loss = sum|a-b| / num, da = sign(a-b) * weight / num
// kind 1: EuclideanLoss:         loss = sum (a-b)^2 / (2 num), da = (a-b) * weight / num        (da may be NULL)
int fcn_loss_f32(int kind, const float* a, const float* b, float* da, float* d_loss, int pixels, int C, int cstride, int num, float weight,
                 fcn_stream_t s) {
    FCN_REQUIRE(a && b && d_loss && pixels > 0 && C > 0 && cstride >= C && num > 0 && (kind == 0 || kind == 1), FCN_E_ARG, "loss: bad args");
    hipStream_t st = as_stream(s);
    if (kind == 0)
        hipLaunchKernelGGL(loss_kernel<true>, dim3(1), dim3(1024), 0, st, a, b, da, d_loss, (long long)pixels, C, cstride, 1.f / (float)num, weight);
    else
        hipLaunchKernelGGL(loss_kernel<false>, dim3(1), dim3(1024), 0, st, a, b, da, d_loss, (long long)pixels, C, cstride, 1.f / (float)num, weight);
    FCN_LAUNCH_CHECK("loss");
    return 0;
}

int fcn_sgd_update_f32(float* w, const float* g, float* hist, const fcn_solver_seg* d_segs, int nseg, float rate, float momentum,
                       float weight_decay, float grad_scale, fcn_stream_t s) {
    FCN_REQUIRE(w && g && hist && d_segs && nseg > 0 && nseg <= 65535, FCN_E_ARG, "sgd_update: bad args");
    hipLaunchKernelGGL(sgd_kernel, dim3(64, nseg), dim3(256), 0, as_stream(s), w, g, hist, reinterpret_cast<const SolverSeg*>(d_segs), nseg, rate,
                       momentum, weight_decay, grad_scale);
    FCN_LAUNCH_CHECK("sgd_update");
    return 0;
}

int fcn_adam_update_f32(float* w, const float* g, float* m, float* v, const fcn_solver_seg* d_segs, int nseg, float rate, float beta1,
                        float beta2, float delta, float weight_decay, int t, float grad_scale, fcn_stream_t s) {
    FCN_REQUIRE(w && g && m && v && d_segs && nseg > 0 && nseg <= 65535 && t >= 1, FCN_E_ARG, "adam_update: bad args");
    const double corr = sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_kernel, dim3(64, nseg), dim3(256), 0, as_stream(s), w, g, m, v, reinterpret_cast<const SolverSeg*>(d_segs), nseg,
                       (float)(rate * corr), beta1, beta2, delta, weight_decay, grad_scale);
    FCN_LAUNCH_CHECK("adam_update");
    return 0;
}

}  // extern "C"
