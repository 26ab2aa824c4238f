// Persistent streaming convolution for half-float activations on gfx950 (BASELINE configs[4]: batch-32 inference).
//
// Stands in for the same Caffe ConvolutionLayer::Forward (+ in-place ReLU) as conv_fwd.hip, for the launches of a half-float
// net whose M = N*OH*OW is large (models/deploy.prototxt at batch 32: M = 25 088 .. 401 408).  Same operands, same layouts
// (NHWC halves, OHWI halves, f32 bias), same implicit GEMM  out[m][n] = bias[n] + sum_k A[m][k] * Wt[n][k].
//
// Why a kernel of its own (round 3, tools/conv_timeline.py + the elimination builds of `make exp`, gpurun_out/r3/):
// v_mfma_f32_32x32x16_f16 is 16x faster than the f32 instruction, so a tile's K loop is short - conv2/3x3 (K = 576) as
// 256 x 128 tiles multiplies for 4 us - and everything AROUND the loop decides: with one workgroup per tile the first chunk
// is usable 3.2 us after the workgroup starts (set-up + one cold memory latency) and the epilogue through LDS (park the
// accumulators, read them back, 8-byte stores) takes 5.2 us; nothing overlaps them when a 144 KB ring leaves room for one
// workgroup per CU.  The launch took 196 us with its MFMAs, 182 us without them and without fragment reads, 157 us without
// any staging.  Here
//   * workgroups are PERSISTENT (one per CU) and walk tiles v, v + G, ...; the LDS ring never drains: the chunk that follows
//     a tile's last chunk is the next tile's first, issued by the loading waves D chunks ahead across the tile boundary, so a
//     tile has no prologue;
//   * the accumulator is TRANSPOSED - the weights are the MFMA's A operand and the pixels its B operand - and the weight rows
//     of a 32-channel tile are dealt to the MFMA rows so that a lane ends up with 16 CONSECUTIVE output channels of ONE pixel
//     (two runs of 8): bias arrives as the C operand of the tile's first MFMA, ReLU and rounding run on registers, and each
//     lane stores 16 bytes per run straight to HBM.  No LDS in the epilogue, no barrier, and the loaders keep streaming the
//     next tile meanwhile;
//   * roles are split as in conv_fwd.hip's cfg 23-29: waves 0-3 multiply (fragment reads + MFMAs only, up to 4 x 2 MFMA
//     tiles = 128 x 64 outputs per wave, the fragments of the k-step after the current one in flight), waves 4-7 load
//     (scalar-addressed "lean" loader: tap / channel position in SGPRs, `buffer_load ... lds` with out-of-range offsets as the
//     zero fill); one counted s_waitcnt vmcnt + one raw s_barrier per chunk;
//   * workgroup b's tiles are consecutive on its XCD (b % 8 labels the XCD: tiles that share an A or B operand meet in one L2).
// Takes: half inputs and outputs, flags within {RELU}, output channel counts / strides / offsets that are multiples of 8,
// every problem lean (1x1 filters, or each tap's Cin padded to whole chunks at a cost of at most 2x).  Everything else stays
// with the tiled family.
#include "conv_common.h"

using namespace fcn;

namespace {

template <int WTM_, int WTN_, int BK_, int NBUF_>
struct SCfg {
    static constexpr int WTM = WTM_, WTN = WTN_, BK = BK_, NBUF = NBUF_;
    static constexpr int NW = 4;                        // multiplying waves (2 x 2) = loading waves
    static constexpr int NT = 64 * NW * 2;
    static constexpr int BM = 64 * WTM, BN = 64 * WTN;  // pixels x output channels of a tile
    static constexpr int SEGS = BK / 4;                 // 16-byte slots per staged row
    static constexpr int RPI = 256 / BK;                // rows one LDS-DMA wave-instruction (1 KiB) fills
    static constexpr int STEP = RPI * NW;
    static constexpr int IA = BM / STEP, IB = BN / STEP, INST = IA + IB;
    static constexpr int D = NBUF - 1;                  // chunks in flight
    static constexpr int KS = BK / 8;                   // k-steps (one v_mfma_f32_32x32x16_f16 per MFMA tile) per chunk
    static constexpr int BKE = BK * 2;                  // halves of K per chunk
    static constexpr int BUF_BYTES = (BM + BN) * BK * 4;
    static constexpr int LDS_BYTES = NBUF * BUF_BYTES;
    static_assert(BK == 16 || BK == 32, "row swizzle: 4 or 8 slots per row");
    static_assert(KS % 2 == 0, "the fragment parity of a k-step must not depend on the chunk");
    static_assert(BM % STEP == 0 && BN % STEP == 0 && STEP % 16 == 0, "rows split into whole wave-instructions; a lane's rows agree mod 16");
    static_assert(NBUF >= 3 && INST * (D - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
    static_assert(WTM <= 4 && WTN <= 2, "fragment reads are written out for at most 4 x 2 MFMA tiles per wave");
};

// what a multiplying wave keeps of a problem
struct MulP {
    f16_t* y;
    const float* bias;
    int M, Cout, y_cstride, y_coffset, relu, nch, m0, n0;
};

template <class C>
__global__ __launch_bounds__(C::NT) void conv_stream_f16(const int nprob, const int te0, const int te1, const int te2, const int te3, const int te4,
                                                         const int te5, const int te6, const int te7, const int total, const int unused,
                                                         const GroupArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WTM = C::WTM, WTN = C::WTN, BK = C::BK, NBUF = C::NBUF, NW = C::NW, BM = C::BM, BN = C::BN, SEGS = C::SEGS, RPI = C::RPI;
    constexpr int STEP = C::STEP, IA = C::IA, IB = C::IB, INST = C::INST, D = C::D, KS = C::KS, BKE = C::BKE, BUF = C::BUF_BYTES;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    typedef const GroupArgs __attribute__((address_space(4))) * karg_ptr;
    constexpr size_t kArgsOffset = (11 * sizeof(int) + alignof(GroupArgs) - 1) / alignof(GroupArgs) * alignof(GroupArgs);
    karg_ptr ka = (karg_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kArgsOffset);
    const int head[1 + kMaxGroup] = {nprob, te0, te1, te2, te3, te4, te5, te6, te7};

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool is_loader = wid_all >= NW;
    const int wid = wid_all & (NW - 1);
    const int G = gridDim.x;
    // Blocks b and b + 8 share an XCD (round-robin placement: observed, used for speed only): give the blocks of one XCD
    // consecutive tiles, so that the tiles that share operand rows (the column tiles of one row of pixels) meet in one L2.
    int v;
    {
        const int q = G >> 3, r = G & 7, x = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    }
    // tile -> (problem index, first tile of that problem): scalar compares on the preloaded prefix table
    auto find_problem = [&](const int tile, int& pi, int& begin) {
        pi = 0;
        begin = 0;
#pragma unroll
        for (int i = 0; i < kMaxGroup - 1; ++i) {
            const int end_i = head[1 + i];
            const bool past = i + 1 < nprob && tile >= end_i;
            pi += past ? 1 : 0;
            begin = past ? end_i : begin;
        }
        pi = __builtin_amdgcn_readfirstlane(pi);
        begin = __builtin_amdgcn_readfirstlane(begin);
    };
    auto load_problem = [&](const int pi) {
        u32x16 ra, rb;
        u32x8 rc;
        const ConvP __attribute__((address_space(4)))* pbase = &ka->p[pi];
        asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx16 %1, %3, 0x40\n\ts_load_dwordx8 %2, %3, 0x80\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(ra), "=&s"(rb), "=&s"(rc)
                     : "s"(pbase)
                     : "memory");
        ConvP prob;
        __builtin_memcpy((char*)&prob, &ra, 64);
        __builtin_memcpy((char*)&prob + 64, &rb, 64);
        __builtin_memcpy((char*)&prob + 128, &rc, 32);
        return prob;
    };
    constexpr int OOB = (int)0x80000000u;      // >= num_records of every buffer (operands and outputs stay below 2 GiB: stream_problem_ok)

    if (is_loader) {
        // ---- loading waves: the scalar-addressed loader of conv_fwd.hip, walking tile after tile ---------------------------------
        const int lrow = RPI * wid + lane / SEGS;                  // row of instruction 0
        const int lseg = (lane % SEGS) ^ swz<SEGS>(lrow);          // k-segment this lane fetches (the same for all its rows)
        const int lane_c = lseg * 8;                               // channel of that segment inside a chunk
        char* const lds_wave = smem + RPI * wid * BK * 4;
        int a_iy0[IA], a_ix0[IA], a_vo[IA], b_vo[IB];
        int ti = v;                                                // tile whose chunks are being issued
        bool live = true;
        const f16_t* px = nullptr;
        const f16_t* pw = nullptr;
        int x_bytes = 0, w_bytes = 0, pH = 0, pW = 0, pCin = 0, pkw = 0, pxcs = 0;
        int s_kr = 0, s_kq = 0, s_kc = 0, s_koff = 0, s_kb = 0, s_left = 0;
        auto setup = [&]() {
            int pi, begin;
            find_problem(ti, pi, begin);
            const ConvP p = load_problem(pi);
            const int lt = ti - begin;
            const int tile_m = fast_div(lt, p.tiles_n_magic);
            const int tile_n = lt - tile_m * p.tiles_n;
            const int m0 = tile_m * BM, n0 = tile_n * BN;
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const int m = m0 + STEP * i + lrow;
                const bool ok = m < p.M;
                const int mm = ok ? m : 0;
                const int t = fast_div(mm, p.ow_magic);
                const int ox = mm - t * p.OW;
                const int img = fast_div(t, p.oh_magic);
                const int oy = t - img * p.OH;
                a_iy0[i] = ok ? oy * p.stride - p.pad : -(1 << 20);      // rows past M never pass the bounds test
                a_ix0[i] = ox * p.stride - p.pad;
                a_vo[i] = (((img * p.H + a_iy0[i]) * p.W + a_ix0[i]) * p.x_cstride + lane_c) * 2;
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                const int n = n0 + STEP * i + lrow;
                b_vo[i] = n < p.Cout ? (n * p.K + lane_c) * 2 : OOB;
            }
            px = reinterpret_cast<const f16_t*>(p.x);
            pw = reinterpret_cast<const f16_t*>(p.w);
            x_bytes = (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * 2);
            w_bytes = p.Cout * p.K * 2;
            pH = p.H; pW = p.W; pCin = p.Cin; pkw = p.kw; pxcs = p.x_cstride;
            s_kr = s_kq = s_kc = s_koff = s_kb = 0;
            s_left = p.lean_chunks;
        };
        auto issue_chunk = [&](const int buf) {
            if (s_left == 0 && live) {      // the tile is issued: on to the next one (or to all-zero chunks behind the last)
                ti += G;
                if (ti < total) setup();
                else live = false;
            }
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(px), 0, x_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(pw), 0, w_bytes, 0x00020000);
            const int thr = live ? pCin - s_kc : 0;      // channels of this tap the chunk still covers (0: behind the last tile)
            const bool cvalid = lane_c < thr;
            char* const dst = lds_wave + buf * BUF;
#pragma unroll
            for (int i = 0; i < IA; ++i) {
                const bool ok = (int)cvalid & (int)((unsigned)(a_iy0[i] + s_kr) < (unsigned)pH) & (int)((unsigned)(a_ix0[i] + s_kq) < (unsigned)pW);
                int vo = ok ? a_vo[i] + s_koff : OOB;
                asm volatile("" : "+v"(vo));      // one select, one DMA
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(dst + STEP * i * BK * 4), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < IB; ++i) {
                int vo = cvalid ? b_vo[i] + s_kb : OOB;
                asm volatile("" : "+v"(vo));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + (BM + STEP * i) * BK * 4), 16, vo, 0, 0, 0);
            }
            // the next chunk's K position: scalar unit only
            --s_left;
            const bool nt = s_kc + BKE >= pCin;                  // the next chunk starts the next tap
            s_kb += (nt ? pCin - s_kc : BKE) * 2;                // weights are [tap][Cin]: the tap's end, or one chunk on
            const int kq1 = s_kq + (nt ? 1 : 0);
            const bool wq = kq1 == pkw;
            s_kq = wq ? 0 : kq1;
            s_kr += wq ? 1 : 0;
            s_kc = nt ? 0 : s_kc + BKE;
            s_koff = nt ? (s_kr * pW + s_kq) * pxcs * 2 : s_koff + BKE * 2;
        };
        setup();
        // the barrier walker: which tile the multipliers are in (its chunk count decides when the workgroup is done)
        int tb = v, left_b = s_left;
        int buf_issue = 0;
        auto next = [](int b) { return b + 1 == NBUF ? 0 : b + 1; };
#pragma unroll 1
        for (int c = 0; c < D; ++c) {
            issue_chunk(buf_issue);
            buf_issue = next(buf_issue);
        }
        wait_vmcnt<INST*(D - 1)>();        // chunk 0 landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();      // ... and everybody else's
#pragma unroll 1
        while (tb < total) {
            wait_vmcnt<INST*(D - 2)>();    // the chunk after the one the multipliers start now has landed
            __builtin_amdgcn_s_barrier();  // the multipliers are done with the chunk before it: its slot takes the chunk D ahead
            asm volatile("" ::: "memory");
            issue_chunk(buf_issue);
            buf_issue = next(buf_issue);
            if (--left_b == 0) {
                tb += G;
                if (tb < total) {
                    int pi, begin;
                    find_problem(tb, pi, begin);
                    left_b = ka->p[pi].lean_chunks;
                }
            }
        }
        wait_vmcnt<0>();                   // the all-zero chunks behind the last tile
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    } else {
        // ---- multiplying waves -----------------------------------------------------------------------------------------------------
        const int wm = wid >> 1, wn = wid & 1;
        const int fi = lane & 31, kh = lane >> 5;
        // MFMA row i of a 32-channel tile multiplies weight row tau(i): register r of half-wave kh then holds channel
        // 8 kh + (r & 7) + 16 (r >> 3) of the tile - two runs of 8 consecutive channels per lane.
        const int tau = (fi & 3) + 4 * ((fi >> 3) & 1) + 8 * ((fi >> 2) & 1) + 16 * (fi >> 4);
        const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
        unsigned fx[KS], fw[KS];      // byte address of this lane's pixel / weight fragment inside ring slot 0, per k-step
#pragma unroll
        for (int st = 0; st < KS; ++st) {
            fx[st] = lds0 + (unsigned)((wm * 32 * WTM + fi) * BK * 4 + (((2 * st + kh) ^ swz<SEGS>(fi)) * 16));
            fw[st] = lds0 + (unsigned)((BM + wn * 32 * WTN + tau) * BK * 4 + (((2 * st + kh) ^ swz<SEGS>(tau)) * 16));
        }
        v4f xf[2][WTM], wf[2][WTN];
        auto ds_read = [](v4f& dst, unsigned addr, auto off) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value));
        };
        auto read_step = [&](const int par, const int st, const unsigned slot_bytes) {
            const unsigned ax = fx[st] + slot_bytes, aw = fw[st] + slot_bytes;
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                if (j == 0) ds_read(wf[par][0], aw, std::integral_constant<int, 0>{});
                if (j == 1) ds_read(wf[par][WTN > 1 ? 1 : 0], aw, std::integral_constant<int, 32 * BK * 4>{});
            }
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                if (i == 0) ds_read(xf[par][0], ax, std::integral_constant<int, 0>{});
                if (i == 1) ds_read(xf[par][WTM > 1 ? 1 : 0], ax, std::integral_constant<int, 32 * BK * 4>{});
                if (i == 2) ds_read(xf[par][WTM > 2 ? 2 : 0], ax, std::integral_constant<int, 64 * BK * 4>{});
                if (i == 3) ds_read(xf[par][WTM > 3 ? 3 : 0], ax, std::integral_constant<int, 96 * BK * 4>{});
            }
        };
        // LDS operations of a wave return in order: all but the WTM + WTN reads issued last (the next step's) are done.  The
        // fragment registers are pinned across the wait and a scheduling barrier follows: hipcc moves register-only MFMAs
        // past a bare inline-asm s_waitcnt.
        auto step_landed = [&](const int par, auto pending) {
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(pending)::value) : "memory");
#pragma unroll
            for (int i = 0; i < WTM; ++i) asm volatile("" : "+v"(xf[par][i]));
#pragma unroll
            for (int j = 0; j < WTN; ++j) asm volatile("" : "+v"(wf[par][j]));
            __builtin_amdgcn_sched_barrier(0);
        };
        f32x16 acc[WTM][WTN], biasv[WTN];
        auto mfma_step = [&](const int par, const bool first) {
#pragma unroll
            for (int i = 0; i < WTM; ++i)
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, wf[par][j]), __builtin_bit_cast(v8h, xf[par][i]),
                                                                       first ? biasv[j] : acc[i][j], 0, 0, 0);
        };
        unsigned slot_cur = 0;      // byte offset of the current chunk's ring slot
        auto chunk = [&](auto first_c) {
            constexpr bool FIRST = decltype(first_c)::value;
            __builtin_amdgcn_s_barrier();      // the next chunk is in LDS too (and the loaders may refill the previous chunk's slot)
            asm volatile("" ::: "memory");
            const unsigned slot_next = slot_cur + BUF == (unsigned)(NBUF * BUF) ? 0u : slot_cur + BUF;
#pragma unroll
            for (int st = 0; st < KS; ++st) {
                if (st + 1 < KS) read_step((st + 1) & 1, st + 1, slot_cur);
                else read_step(0, 0, slot_next);
                step_landed(st & 1, std::integral_constant<int, WTM + WTN>{});
                mfma_step(st & 1, FIRST && st == 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            slot_cur = slot_next;
        };
        // a tile's problem, as far as this role needs it
        auto mul_problem = [&](const int tile) {
            int pi, begin;
            find_problem(tile, pi, begin);
            const ConvP p = load_problem(pi);      // (ends in s_waitcnt lgkmcnt(0): the scalar loads do not return in order with the LDS reads)
            const int lt = tile - begin;
            const int tile_m = fast_div(lt, p.tiles_n_magic);
            MulP q;
            q.y = reinterpret_cast<f16_t*>(p.y);
            q.bias = p.bias;
            q.M = p.M; q.Cout = p.Cout; q.y_cstride = p.y_cstride; q.y_coffset = p.y_coffset;
            q.relu = p.flags & FCN_CONV_RELU;
            q.nch = p.lean_chunks;
            q.m0 = tile_m * BM;
            q.n0 = (lt - tile_m * p.tiles_n) * BN;
            return q;
        };
        // bias of this lane's channels as the C operand of the tile's first MFMAs (out-of-range channels read 0; no bias: 0 records)
        auto load_bias = [&](const MulP& q) {
            const __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.bias ? q.bias : reinterpret_cast<const float*>(q.y)), 0,
                                                                                q.bias ? q.Cout * 4 : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int c0 = q.n0 + wn * 32 * WTN + 32 * j + 8 * kh;
#pragma unroll
                for (int g = 0; g < 4; ++g) {      // channels c0 + 0..3, 4..7, 16..19, 20..23
                    const v4f b4 = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rb, (c0 + 4 * (g & 1) + 16 * (g >> 1)) * 4, 0, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) biasv[j][4 * g + e] = b4[e];
                }
            }
        };
        auto epilogue = [&](const MulP& q) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(
                q.y, 0, (int)((((long long)q.M - 1) * q.y_cstride + q.y_coffset + q.Cout) * 2), 0x00020000);
            const h2 zero2 = {(f16_t)0.f, (f16_t)0.f};
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const int m = q.m0 + wm * 32 * WTM + 32 * i + fi;
                const int row_off = (m * q.y_cstride + q.y_coffset) * 2;
#pragma unroll
                for (int j = 0; j < WTN; ++j) {
                    const int c0 = q.n0 + wn * 32 * WTN + 32 * j + 8 * kh;
#pragma unroll
                    for (int run = 0; run < 2; ++run) {
                        u32x4 pk;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h2 t = {(f16_t)acc[i][j][8 * run + 2 * e], (f16_t)acc[i][j][8 * run + 2 * e + 1]};      // rounded once
                            if (q.relu) t = __builtin_elementwise_max(t, zero2);      // (max after rounding = rounding after max)
                            pk[e] = __builtin_bit_cast(unsigned, t);
                        }
                        const int c = c0 + 16 * run;
                        int vo = (m < q.M && c < q.Cout) ? row_off + c * 2 : OOB;
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ry, vo, 0, 0);
                    }
                }
            }
        };

        int tile = v;
        MulP cur = mul_problem(tile);
        load_bias(cur);
        __builtin_amdgcn_s_barrier();      // chunk 0 is in LDS
        asm volatile("" ::: "memory");
        read_step(0, 0, 0u);
#pragma unroll 1
        while (true) {
            chunk(std::true_type{});
            // biasv is dead behind the tile's first k-step: fetch the next tile's now, in front of this tile's stores (the
            // vector-memory counter retires in order: waiting for these loads never waits for the stores behind them)
            const int tn = tile + G;
            const bool more = tn < total;
            MulP nxt = cur;
            if (more) {
                nxt = mul_problem(tn);
                load_bias(nxt);
            }
#pragma unroll 1
            for (int c = 1; c < cur.nch; ++c) chunk(std::false_type{});
            epilogue(cur);
            if (!more) break;
            cur = nxt;
            tile = tn;
        }
        // the fragments prefetched last belong to a chunk nobody multiplies: retire the reads before the registers die
        step_landed(0, std::integral_constant<int, 0>{});
        step_landed(1, std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }
#endif
}

// configurations: X(index, WTM, WTN, BK words, ring slots)
#define FCN_STREAM_CONFIGS(X) \
    X(0, 4, 2, 16, 6)         \
    X(1, 4, 1, 16, 6)         \
    X(2, 2, 2, 16, 6)         \
    X(3, 4, 2, 32, 3)         \
    X(4, 2, 2, 32, 4)

constexpr StreamCfgInfo kStreamCfgs[] = {
#define X(I, A, B, K, N) {SCfg<A, B, K, N>::BM, SCfg<A, B, K, N>::BN, K, SCfg<A, B, K, N>::LDS_BYTES, SCfg<A, B, K, N>::NT},
    FCN_STREAM_CONFIGS(X)
#undef X
};

}  // namespace

namespace fcn {

int stream_num_cfgs() { return (int)(sizeof(kStreamCfgs) / sizeof(kStreamCfgs[0])); }

StreamCfgInfo stream_cfg_info(int idx) { return kStreamCfgs[idx]; }

void launch_stream(int idx, const GroupArgs& ga, int total, hipStream_t st) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = total < cus ? total : cus;      // persistent: one workgroup per compute unit
    switch (idx) {
#define X(I, A, B, K, N)                                                                                                                              \
    case I:                                                                                                                                           \
        hipLaunchKernelGGL((conv_stream_f16<SCfg<A, B, K, N>>), dim3(grid), dim3(SCfg<A, B, K, N>::NT), 0, st, ga.nprob, ga.tile_end[0], ga.tile_end[1], \
                           ga.tile_end[2], ga.tile_end[3], ga.tile_end[4], ga.tile_end[5], ga.tile_end[6], ga.tile_end[7], total, 0, ga);              \
        break;
        FCN_STREAM_CONFIGS(X)
#undef X
    }
}

}  // namespace fcn
