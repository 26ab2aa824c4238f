// Persistent streaming convolution for half-float activations on gfx950 (BASELINE configs[4]: batch-32 inference).
//
// Stands in for the same Caffe ConvolutionLayer::Forward (+ in-place ReLU) as conv_fwd.hip, for the launches of a half-float
// net whose M = N*OH*OW is large (models/deploy.prototxt at batch 32: M = 25 088 .. 401 408): the stride-1 "same" convolutions
// (1x1, 3x3 / pad 1, 5x5 / pad 2).  Same operands, same layouts (NHWC halves, OHWI halves, f32 bias), same implicit GEMM
// out[m][n] = bias[n] + sum_k A[m][k] * Wt[n][k].
//
// Why a kernel of its own (round 3; tools/conv_timeline.py, the elimination builds of `make exp`, tools/probes/stage_swz_probe.hip;
// records in gpurun_out/r3 and DESIGN.md 4.7): v_mfma_f32_32x32x16_f16 is 16x faster than the f32 instruction, so what bounds a
// half-float launch is everything AROUND the MFMAs.  conv2/3x3 as 256 x 128 tiles of the tiled family took 196 us with its MFMAs,
// 182 us without MFMAs and fragment reads, 157 us without any staging: per tile 3.2 us until the first chunk is usable and 5.2 us
// of epilogue through LDS, and with one workgroup per CU nothing overlaps them.  A first persistent version (im2col chunks, halo
// tests per piece) still took 57 us with NO loads, reads, MFMAs or stores at all: ~100 vector instructions of loader bookkeeping
// per chunk set the chunk period.  Here
//   * workgroups are PERSISTENT (one per CU) and walk tiles v, v + G, ...; the LDS rings never drain: loaders run ahead across
//     tile boundaries, so a tile has no prologue;
//   * a tile is 256 consecutive output pixels.  For filter row r and a 64-channel chunk the input pixels under the tile are
//     staged ONCE as a SLAB in "padded raster order" (every image row followed by 2 pad zero pixels): the kw taps of the row
//     read the same slab kw times at row offsets 0 .. kw-1 - no halo tests and no re-staging per tap (3x / 5x less activation
//     traffic than im2col chunks), and a slab piece costs the loader four vector instructions;
//   * loading is split: waves 4-5 stage slabs, waves 6-7 stage weight chunks (BN channels x 64 k), each with its own
//     vector-memory counter, so one counted s_waitcnt per role; all LDS rows are 128 bytes (64-byte rows halve the LDS-DMA
//     rate: a 128-byte line is then fetched by two instructions);
//   * the accumulator is TRANSPOSED - weights are the MFMA's A operand, pixels its B operand - and the weight rows of a
//     32-channel tile are dealt to the MFMA rows so that a lane ends up with 16 consecutive output channels of ONE pixel (two
//     runs of 8): the accumulators start as the bias, ReLU and rounding run on registers, each lane stores
//     16 bytes per run straight to HBM.  No LDS in the epilogue, no barrier; the loaders stream the next tile meanwhile;
//   * waves 0-3 multiply (128 x 32 WTN outputs each = 4 x WTN MFMA tiles; fragment reads + MFMAs + one address instruction per
//     read, the fragments of the next k-step in flight); one raw s_barrier per chunk;
//   * workgroup b's tiles are consecutive on its XCD (b % 8 labels the XCD): tiles that share operands meet in one L2.
// Takes (plan_tiles_cfg): half inputs and outputs, flags within {RELU}, stride 1 with pad = (k - 1) / 2 and k in {1, 3, 5}, Cout /
// y_cstride / y_coffset multiples of 8, image rows wide enough that a tile's slab fits.  Everything else stays with the tiled family.
#include "conv_common.h"

using namespace fcn;

namespace {

// Diagnostic build only (make exp EXP=-DFCN_STREAM_STAMPS): thread 0 of each role sums the 100 MHz clock over its phases into a
// buffer of its own (tools/stream_timeline.py).  No stamp exists in the product build; no output value depends on one.
#ifdef FCN_STREAM_STAMPS
__device__ unsigned long long* g_stream_stamps = nullptr;
__device__ int g_stream_stamps_cap = 0;
#define SNOW() __builtin_amdgcn_s_memrealtime()
#define SSTAMP_DECL unsigned long long st_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t0_ = SNOW(), st_t_ = st_t0_, st_c0_ = __builtin_amdgcn_s_memtime()
#define SSTAMP(i)                                         \
    do {                                                  \
        const unsigned long long n_ = SNOW();             \
        st_[i] += n_ - st_t_;                             \
        st_t_ = n_;                                       \
    } while (0)
#define SSTAMP_FLUSH(role)                                                                                           \
    do {                                                                                                             \
        if (g_stream_stamps && (int)blockIdx.x < g_stream_stamps_cap && (threadIdx.x & 63) == 0) {                    \
            unsigned long long* d_ = g_stream_stamps + ((size_t)blockIdx.x * 3 + (role)) * 8;                        \
            for (int k_ = 0; k_ < 8; ++k_) d_[k_] = st_[k_];                                                          \
        }                                                                                                            \
    } while (0)
#else
#define SSTAMP_DECL do { } while (0)
#define SSTAMP(i) do { } while (0)
#define SSTAMP_FLUSH(role) do { } while (0)
#endif

#ifdef FCN_EXP_NOPRIO
constexpr bool getenv_free_noprio = true;
#else
constexpr bool getenv_free_noprio = false;
#endif
constexpr int kChunkK = 64;       // halves of K per chunk (128-byte LDS rows)

template <int WTN_, int SRP_, int NA_, int NB_, int WTM_ = 4>
struct SCfg {
    static constexpr int WTN = WTN_, SRP = SRP_, NA = NA_, NB = NB_;
    static constexpr int WTM = WTM_, BM = 64 * WTM, BN = 64 * WTN;      // the multiplying waves form a 2 x 2 grid of (32 WTM) x (32 WTN) outputs
    static_assert(WTM == 4 || WTM == 2, "pixel tiles of 256 or 128");
    static_assert(WTN >= 1 && WTN <= 3, "weight fragments are read with up to three fixed offsets");
    static constexpr int NT = 512;
    static constexpr int NPA = SRP / 2;                 // slab pieces (1 KiB = 8 rows) per slab-loading wave
    static constexpr int NPB = BN / 16;                 // weight pieces per weight-loading wave and chunk
    static constexpr int SLAB_BYTES = SRP * 1024, WCH_BYTES = BN * 128;
    static constexpr int LDS_BYTES = NA * SLAB_BYTES + NB * WCH_BYTES;
    static_assert(SRP % 2 == 0 && NA >= 2 && NB >= 3, "ring shapes");
    static_assert(NPA * (NA - 1) <= 63 && NPB * (NB - 1) <= 63, "vmcnt is a 6-bit counter");
    static_assert(LDS_BYTES <= 160 * 1024, "exceeds the CU's 160 KiB LDS");
};

template <class C>
__global__ __launch_bounds__(C::NT) void conv_stream_f16(const int nprob, const int te0, const int te1, const int te2, const int te3, const int te4,
                                                         const int te5, const int te6, const int te7, const int total, const int unused,
                                                         const GroupArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int WTM = C::WTM, WTN = C::WTN, NA = C::NA, NB = C::NB, NPA = C::NPA, NPB = C::NPB, BM = C::BM, BN = C::BN;
    constexpr int SLAB = C::SLAB_BYTES, WCH = C::WCH_BYTES, WRING0 = NA * SLAB;
    __shared__ __attribute__((aligned(16))) char smem[C::LDS_BYTES];
    typedef unsigned u32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
    typedef const GroupArgs __attribute__((address_space(4))) * karg_ptr;
    constexpr size_t kArgsOffset = (11 * sizeof(int) + alignof(GroupArgs) - 1) / alignof(GroupArgs) * alignof(GroupArgs);
    karg_ptr ka = (karg_ptr)((const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr() + kArgsOffset);
    const int head[1 + kMaxGroup] = {nprob, te0, te1, te2, te3, te4, te5, te6, te7};

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = gridDim.x;
    // Blocks b and b + 8 share an XCD (round-robin placement: observed, used for speed only): give the blocks of one XCD
    // consecutive tiles, so that the tiles that share operand rows (the column tiles of one row of pixels) meet in one L2.
    int v;
    {
        const int q = G >> 3, r = G & 7, x = (int)blockIdx.x & 7, k = (int)blockIdx.x >> 3;
        v = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
    }
    // tile -> (problem index, first tile of that problem): scalar compares on the preloaded prefix table
    auto find_problem = [&](const int tile, int& pi, int& begin) __attribute__((always_inline)) {
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
    auto load_problem = [&](const int pi) __attribute__((always_inline)) {      // (ends in s_waitcnt lgkmcnt(0): scalar loads do not return in order with LDS reads)
        u32x16 ra, rb;
        u32x8 rc;
        // (the address is wave-uniform by construction; say so, or a use under a branch the compiler cannot prove uniform lands in VGPRs)
        const unsigned long long pa = (unsigned long long)&ka->p[pi];
        const unsigned long long pu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pa >> 32)) << 32) |
                                      (unsigned)__builtin_amdgcn_readfirstlane((int)pa);
        const ConvP __attribute__((address_space(4)))* pbase = (const ConvP __attribute__((address_space(4)))*)pu;
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
    // The barrier walker every loading wave keeps: which chunk the multiplying waves start at the next barrier.
    struct Walk {
        int tile, kw, q, mac_left;
    };
    auto walk_begin = [&](Walk& w, const int tile) __attribute__((always_inline)) {
        w.tile = tile;
        w.q = 0;
        w.kw = 1;
        w.mac_left = 1;
        if (tile < total) {
            int pi, begin;
            find_problem(tile, pi, begin);
            const int kh = ka->p[pi].kh, kw = ka->p[pi].kw, cin = ka->p[pi].Cin, tsh = ka->p[pi].lean_chunks;
            w.kw = (kw + (1 << tsh) - 1) >> tsh;      // chunks per slab: one per tap, or one per 2 / 4 taps of a narrow input (packed taps)
            w.mac_left = tsh ? kh : kh * ((cin + kChunkK - 1) / kChunkK);
        }
    };
    auto walk_step = [&](Walk& w) __attribute__((always_inline)) {
        if (++w.q == w.kw) {
            w.q = 0;
            if (--w.mac_left == 0) walk_begin(w, w.tile + G);
        }
    };
    constexpr int OOB = (int)0x80000000u;      // >= num_records of every buffer (operands and outputs stay below 2 GiB: plan_tiles_cfg)

    if (wid_all >= 6) {
        // ---- weight-loading waves: chunk (r, cc, q) of a tile = rows n0 .. n0 + BN - 1 of the bank, 64 k each ---------------------
        const int wb = wid_all - 6;
        // piece p = wb + 2 t holds bank rows 8 p .. 8 p + 7; lane -> row 8 p + lane / 8, k-segment (lane % 8) ^ swz(row)
        // ((row >> 1) & 7 = (4 (p & 1) + (lane >> 4)) & 7 and p & 1 = wb: one segment per lane for all its pieces)
        const int lseg = (lane & 7) ^ ((4 * wb + (lane >> 4)) & 7);
        const int lane_c = lseg * 8;
        char* const lds_wave = smem + WRING0 + wb * 1024;
        int b_vo[NPB];
        int ti = v, r = 0, cc = 0, q = 0, kh = 1, kw = 1, ncc = 1, cin = 0, tsh = 0, cps = 1;
        bool live = true;
        const f16_t* pw = nullptr;
        int w_bytes = 0, cur_pi = -1, p_cout = 0, p_K = 0, p_tiles_n = 1;
        unsigned p_tiles_n_magic = 0;
        auto setup = [&]() __attribute__((always_inline)) {
            int pi, begin;
            find_problem(ti, pi, begin);
            if (pi != cur_pi) {      // (the problem changes at most a few times per launch: no scalar reload otherwise - every
                                     //  s_barrier needs all eight waves, so a load latency here stalls the multipliers)
                const ConvP p = load_problem(pi);
                cur_pi = pi;
                pw = reinterpret_cast<const f16_t*>(p.w);
                w_bytes = p.Cout * p.K * 2;
                kh = p.kh; kw = p.kw; cin = p.Cin;
                tsh = p.lean_chunks;                      // packed taps (plan_tiles_cfg): a chunk's 64 k are 2 / 4 taps x Cin = 32 / 16 channels,
                cps = (p.kw + (1 << tsh) - 1) >> tsh;     // contiguous in the OHWI bank; the last chunk of a filter row may hold fewer taps
                ncc = tsh ? 1 : (p.Cin + kChunkK - 1) / kChunkK;
                p_cout = p.Cout; p_K = p.K; p_tiles_n = p.tiles_n; p_tiles_n_magic = p.tiles_n_magic;
            }
            const int lt = ti - begin;
            const int tile_m = fast_div(lt, p_tiles_n_magic);
            const int n0 = (lt - tile_m * p_tiles_n) * BN;
#pragma unroll
            for (int t = 0; t < NPB; ++t) {
                const int n = n0 + 8 * (wb + 2 * t) + (lane >> 3);
                b_vo[t] = n < p_cout ? (n * p_K + lane_c) * 2 : OOB;
            }
            r = cc = q = 0;
        };
        auto issue_chunk = [&](const int slot) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(pw), 0, w_bytes, 0x00020000);
            const int soff = ((r * kw + (q << tsh)) * cin + cc * kChunkK) * 2;
            // (the last chunk of a tap may cover fewer than 64 channels; the last chunk of a packed filter row fewer taps)
            const bool cvalid = live && lane_c < (tsh ? min(1 << tsh, kw - (q << tsh)) * cin : cin - cc * kChunkK);
            char* const dst = lds_wave + slot * WCH;
#pragma unroll
            for (int t = 0; t < NPB; ++t) {
                int vo = cvalid ? b_vo[t] : OOB;
                asm volatile("" : "+v"(vo));
#ifndef FCN_EXP_NOLOAD
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr)(dst + 2048 * t), 16, vo, soff, 0, 0);
#endif
            }
            if (live) {      // next chunk: taps of the row, then channel chunks, then filter rows, then the next tile
                if (++q == cps) {
                    q = 0;
                    if (++cc == ncc) {
                        cc = 0;
                        if (++r == kh) {
                            ti += G;
                            if (ti < total) setup();
                            else live = false;
                        }
                    }
                }
            }
        };
        SSTAMP_DECL;
        setup();
        Walk wk;
        walk_begin(wk, v);
        int slot = 0;
        auto next = [](int b) __attribute__((always_inline)) { return b + 1 == NB ? 0 : b + 1; };
#pragma unroll 1
        for (int c = 0; c < NB - 1; ++c) {
            issue_chunk(slot);
            slot = next(slot);
        }
        wait_vmcnt<NPB*(NB - 2)>();        // chunk 0 landed
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        while (wk.tile < total) {
            SSTAMP(0);                     // [0] issue + walk
            wait_vmcnt<NPB*(NB - 3)>();    // the chunk after the one the multipliers start now has landed
            SSTAMP(1);                     // [1] waiting for loads
            __builtin_amdgcn_s_barrier();  // the multipliers are done with the chunk before it: its slot takes the chunk NB - 1 ahead
            asm volatile("" ::: "memory");
            SSTAMP(2);                     // [2] waiting at the barrier
            issue_chunk(slot);
            slot = next(slot);
            walk_step(wk);
        }
        wait_vmcnt<0>();                   // the all-zero chunks behind the last tile
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wb == 0) SSTAMP_FLUSH(2);
    } else if (wid_all >= 4) {
        // ---- slab-loading waves: slab (r, cc) of a tile = padded-raster pixels Pbase .. Pbase + 8 SRP - 1 of input row offset r ---
        const int wa = wid_all - 4;
        const int lseg = (lane & 7) ^ ((4 * wa + (lane >> 4)) & 7);      // as for the weight pieces: piece p = wa + 2 t, row 8 p + lane / 8
        const int lane_c = lseg * 8;
        char* const lds_wave = smem + wa * 1024;
        // what the loader keeps of a problem
        struct SlabP {
            const f16_t* px;
            int x_bytes, H, W, pad, kh, cin, ncc, row_bytes, xcs, PW, rows_total, tiles_n, pi, dpx;
            unsigned ow_magic, oh_magic, pw_magic, tiles_n_magic;
        };
        auto slab_problem = [&](const int pi) __attribute__((always_inline)) {
            const ConvP p = load_problem(pi);
            SlabP q;
            q.px = reinterpret_cast<const f16_t*>(p.x);
            q.x_bytes = (int)((((long long)p.N * p.H * p.W - 1) * p.x_cstride + p.Cin) * 2);
            q.H = p.H; q.W = p.W; q.pad = p.pad; q.kh = p.kh;
            // Packed taps (x_cstride == Cin == 16 or 32, plan_tiles_cfg): the 128 bytes behind a pixel ARE the next 3 (1) pixels of its
            // image row, so a slab row holds the channels of 4 (2) neighbouring padded-raster entries - four (two) taps of a filter row
            // in one 64-wide chunk.  Same address as ever; a lane's segment just belongs to entry p + dpx and is tested against the
            // image row with that offset (entries outside the image are the zero padding).
            const int tsh = p.lean_chunks;
            q.dpx = tsh ? lseg >> (3 - tsh) : 0;
            q.cin = tsh ? kChunkK : p.Cin;
            q.ncc = tsh ? 1 : (p.Cin + kChunkK - 1) / kChunkK;
            q.row_bytes = p.W * p.x_cstride * 2;
            q.xcs = p.x_cstride;
            q.PW = p.W + 2 * p.pad;
            q.rows_total = p.N * p.H;
            q.tiles_n = p.tiles_n; q.pi = pi;
            q.ow_magic = p.ow_magic; q.oh_magic = p.oh_magic; q.pw_magic = p.cin_magic24; q.tiles_n_magic = p.tiles_n_magic;
            return q;
        };
        // Per tile and piece: byte offset of the lane's segment in the CENTRE row (r = pad) and a bit per filter row whose input row
        // exists.  The wave's pieces are 16 slab rows apart: ONE division for its first row, then steps of 16 entries with a wrap at the
        // end of an image row (PW >= 16, plan_tiles_cfg).  Every s_barrier needs all eight waves, so a wave that spends a microsecond
        // on a tile's set-up stalls the multipliers for that long whatever the rings' lead: the NEXT tile's table is therefore built
        // in slices of kSlice pieces, one slice per chunk iteration, into a second register set, and swapped in at the tile boundary.
        constexpr int kSlice = 4, kSlices = (NPA + kSlice - 1) / kSlice;
        int off_c[NPA], vmask[NPA], off_n[NPA], vmask_n[NPA];
        SlabP cur, nxt;
        int prep = 0, prep_tile = 0;          // slices of the next tile's table that are built; the tile it is for
        int w_t1 = 0, w_xp = 0, w_y = 0, w_off = 0;      // the walk's state between slices
        auto prep_slice = [&]() __attribute__((always_inline)) {
            if (prep == 0) {
                int pi, begin;
                find_problem(prep_tile, pi, begin);
                if (pi != nxt.pi) nxt = slab_problem(pi);      // (the problem changes at most a few times per launch: no reload otherwise)
                const int lt = prep_tile - begin;
                const int tile_m = fast_div(lt, nxt.tiles_n_magic);
                const int m0 = tile_m * BM;
                // padded raster: P(img, y, x) = (img H + y) PW + x + pad; the slab starts pad entries in front of P(m0)
                const int t0 = fast_div(m0, nxt.ow_magic);
                const int pj = t0 * nxt.PW + (m0 - t0 * nxt.W) + 8 * wa + (lane >> 3);
                w_t1 = fast_div(pj, nxt.pw_magic);
                w_xp = pj - w_t1 * nxt.PW;                    // 0 .. PW - 1: entry inside the padded image row
                w_y = w_t1 - fast_div(w_t1, nxt.oh_magic) * nxt.H;
                w_off = ((w_t1 * nxt.W + w_xp - nxt.pad) * nxt.xcs + lane_c) * 2;
            }
            const int step_off = 16 * nxt.xcs * 2, wrap_off = 2 * nxt.pad * nxt.xcs * 2;
#pragma unroll
            for (int sl = 0; sl < kSlices; ++sl) {
                if (prep == sl) {      // (a scalar branch per slice: the register arrays are indexed statically inside it)
#pragma unroll
                    for (int t = sl * kSlice; t < (sl + 1) * kSlice && t < NPA; ++t) {
                        const bool okx = (unsigned)(w_xp + nxt.dpx - nxt.pad) < (unsigned)nxt.W && w_t1 < nxt.rows_total;
                        const int lo = max(nxt.pad - w_y, 0), hi = min(nxt.H + nxt.pad - w_y, nxt.kh);      // filter rows whose input row exists
                        vmask_n[t] = (okx && hi > lo) ? (1 << hi) - (1 << lo) : 0;
                        off_n[t] = w_off;
                        w_xp += 16;
                        w_off += step_off;
                        const bool wrap = w_xp >= nxt.PW;
                        w_xp -= wrap ? nxt.PW : 0;
                        w_off -= wrap ? wrap_off : 0;
                        w_t1 += wrap ? 1 : 0;
                        w_y += wrap ? 1 : 0;
                        w_y -= w_y >= nxt.H ? nxt.H : 0;
                    }
                }
            }
            ++prep;
        };
        int ti = v, r = 0, cc = 0;
        bool live = true;
        auto next_tile = [&]() __attribute__((always_inline)) {      // the table built for prep_tile becomes the current one
            while (prep < kSlices) prep_slice();
#pragma unroll
            for (int t = 0; t < NPA; ++t) {
                off_c[t] = off_n[t];
                vmask[t] = vmask_n[t];
            }
            cur = nxt;
            r = cc = 0;
            prep = 0;
            prep_tile = ti + G;
        };
        auto issue_slab = [&](const int buf) __attribute__((always_inline)) {
            const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<f16_t*>(cur.px), 0, cur.x_bytes, 0x00020000);
            const int soff = cc * kChunkK * 2;
            const int roff = (r - cur.pad) * cur.row_bytes;
            const bool cvalid = live && lane_c < cur.cin - cc * kChunkK;
            const int bit = 1 << r;
            char* const dst = lds_wave + buf * SLAB;
#pragma unroll
            for (int t = 0; t < NPA; ++t) {
                int vo = (cvalid && (vmask[t] & bit)) ? off_c[t] + roff : OOB;
                asm volatile("" : "+v"(vo));
#ifndef FCN_EXP_NOLOAD
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lds_ptr)(dst + 2048 * t), 16, vo, soff, 0, 0);
#endif
            }
            if (live) {
                if (++cc == cur.ncc) {
                    cc = 0;
                    if (++r == cur.kh) {
                        ti += G;
                        if (ti < total) next_tile();
                        else live = false;
                    }
                }
            }
        };
        SSTAMP_DECL;
        nxt.pi = -1;
        prep_tile = v;
        next_tile();
        Walk wk;
        walk_begin(wk, v);
        int buf = 0;
        auto next = [](int b) __attribute__((always_inline)) { return b + 1 == NA ? 0 : b + 1; };
#pragma unroll 1
        for (int c = 0; c < NA - 1; ++c) {      // slabs 0 .. NA - 2: the lead the loop keeps
            issue_slab(buf);
            buf = next(buf);
        }
        wait_vmcnt<NPA*(NA - 2)>();        // slab 0 landed
        __builtin_amdgcn_s_barrier();
#pragma unroll 1
        while (wk.tile < total) {
            // In front of the barrier that starts the LAST chunk of slab j the multipliers' next fragments may come from slab j + 1:
            // it must have landed.  Slabs are issued behind the FIRST barrier of a slab (NA - 1 ahead), so the slabs younger than
            // j + 1 that are in flight here number NA - 2 - or NA - 3 when the slab has a single chunk (1x1 filters: its own issue
            // comes behind this very barrier; plan_tiles_cfg keeps 1x1 problems away from two-buffer configurations).
            SSTAMP(0);
            if (wk.q == wk.kw - 1) {
                if (wk.kw == 1) wait_vmcnt<NPA*(NA >= 3 ? NA - 3 : 0)>();
                else wait_vmcnt<NPA*(NA - 2)>();
            }
            SSTAMP(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            SSTAMP(2);
            if (wk.q == 0) {
                issue_slab(buf);
                buf = next(buf);
            }
            if (prep < kSlices && prep_tile < total) prep_slice();      // one slice of the next tile's table per chunk
            walk_step(wk);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (wa == 0) SSTAMP_FLUSH(1);
    } else {
        // ---- multiplying waves -----------------------------------------------------------------------------------------------------
        // Every SIMD hosts one multiplying and one loading wave; vector issue between the two is arbitrated by priority, then age.
        // The loaders have slack (they wait at the barriers most of the time), the multipliers are the critical path: static priority.
        if (!getenv_free_noprio) __builtin_amdgcn_s_setprio(3);
        const int wm = wid_all >> 1, wn = wid_all & 1;
        const int fi = lane & 31, kh_ = lane >> 5;
        // MFMA row i of a 32-channel tile multiplies weight row tau(i): register r of half-wave kh then holds channel
        // 8 kh + (r & 7) + 16 (r >> 3) of the tile - two runs of 8 consecutive channels per lane.
        const int tau = (fi & 3) + 4 * ((fi >> 3) & 1) + 8 * ((fi >> 2) & 1) + 16 * (fi >> 4);
        const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smem;
        // k-segment 2 st + kh of a row whose swizzle term is sw sits at ((2 st + kh) ^ sw) * 16 = (32 st) ^ (((sw ^ kh) & 7) * 16): the
        // step rides in the address instruction as a constant (v_xad_u32)
        const unsigned kh16 = (unsigned)kh_ * 16u;
        const unsigned wrow = lds0 + (unsigned)(WRING0 + (wn * 32 * WTN + tau) * 128), xw = (unsigned)((((tau >> 1) ^ kh_) & 7) * 16);
        // Fragment registers.  FCN_STREAM_XF2 = 0 (the product): ONE set of pixel fragments, each re-read right behind the MFMAs that
        // consumed it - the read then has the other WTM - 1 tiles' MFMAs to land.  FCN_STREAM_XF2 = 1 (experiment, round 3): TWO sets,
        // like the weight fragments - all reads of the next k-step go out at the top of the step and have the whole step's MFMAs to
        // land, one wait per step, 4 WTM more registers (247 / 221).  Measured 1-4 % SLOWER on every shape (conv2/3x3 147 -> 151 us,
        // gpurun_out/r3/sweep_xf2.txt): the multiplying waves do not wait for fragment latency, so the second set buys nothing.
#ifndef FCN_STREAM_XF2
#define FCN_STREAM_XF2 0
#endif
        constexpr int XS = FCN_STREAM_XF2 ? 2 : 1;
        v4f xf[XS][WTM], wf[2][WTN];
        auto ds_read = [](v4f& dst, unsigned addr, auto off) __attribute__((always_inline)) {
#ifdef FCN_EXP_NOREAD
            asm volatile("" : "=v"(dst) : "v"(addr));
            return;
#endif
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value));
        };
        // pixel fragment addressing of a chunk: slab row of tile i's pixel + tap q, as a byte address and its swizzle term
        unsigned rowb[WTM], xk[WTM];            // of the chunk whose fragments are read next
        int rho[WTM], rho_nt[WTM];              // slab row of this lane's pixel per MFMA tile: this tile / the next tile
        auto chunk_addr = [&](const int (&rh)[WTM], const int q, const unsigned abase) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                const unsigned rq = (unsigned)(rh[i] + q);
                rowb[i] = lds0 + abase + rq * 128u;
                xk[i] = (((rq >> 1) & 7u) << 4) ^ kh16;
            }
        };
        auto read_w = [&](const int par, const int st, const unsigned wslot) __attribute__((always_inline)) {
            const unsigned aw = (xw ^ (unsigned)(32 * st)) + (wrow + wslot);
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                if (j == 0) ds_read(wf[par][0], aw, std::integral_constant<int, 0>{});
                if (j == 1) ds_read(wf[par][WTN > 1 ? 1 : 0], aw, std::integral_constant<int, 32 * 128>{});
                if (j == 2) ds_read(wf[par][WTN > 2 ? 2 : 0], aw, std::integral_constant<int, 64 * 128>{});
            }
        };
        auto read_x = [&](const int par, const int i, const int st) __attribute__((always_inline)) {
            ds_read(xf[XS == 2 ? par : 0][i], (xk[i] ^ (unsigned)(32 * st)) + rowb[i], std::integral_constant<int, 0>{});
        };
        // LDS operations of a wave return in order.  In front of the MFMAs of pixel tile i the reads still allowed in flight are the
        // WTM - 1 pixel fragments and WTN weight fragments issued after the ones it needs: one constant count for every wait of the
        // loop.  The fragment registers are pinned across the wait and a scheduling barrier follows: hipcc moves register-only
        // MFMAs past a bare inline-asm s_waitcnt.
        auto frag_landed = [&](const int par, const int i, auto pending) __attribute__((always_inline)) {
#ifdef FCN_EXP_NOWAIT      // elimination build: no fragment waits at all (wrong results: what do the waits cost?)
            return;
#endif
            asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(decltype(pending)::value) : "memory");
            if (XS == 2) {
#pragma unroll
                for (int ii = 0; ii < WTM; ++ii) asm volatile("" : "+v"(xf[par][ii]));
            } else {
                asm volatile("" : "+v"(xf[0][i]));
            }
#pragma unroll
            for (int j = 0; j < WTN; ++j) asm volatile("" : "+v"(wf[par][j]));
            __builtin_amdgcn_sched_barrier(0);
        };
        f32x16 acc[WTM][WTN], biasv[WTN];
        // first: the tile's first k-step - the bias of the lane's channels is the C operand (a scalar branch around the MFMAs of a
        // pair; biasv is dead behind that step: the next tile's is fetched into it during this tile's last chunk, in front of the
        // tile's stores - the vector-memory counter retires in order, so waiting for those loads never waits for the stores)
        auto mfma_pair = [&](const int par, const int i, const bool first) __attribute__((always_inline)) {
#ifdef FCN_CONV_NOMFMA
            return;
#endif
            if (first) {
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, wf[par][j]), __builtin_bit_cast(v8h, xf[XS == 2 ? par : 0][i]), biasv[j], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < WTN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, wf[par][j]), __builtin_bit_cast(v8h, xf[XS == 2 ? par : 0][i]), acc[i][j], 0, 0, 0);
            }
        };
        // what this role keeps of a tile's problem
        struct MulP {
            f16_t* y;
            const float* bias;
            int M, Cout, y_cstride, y_coffset, relu, m0, n0, kw, nch, W, PW, tsh;
            unsigned ow_magic;
        };
        int mul_pi = -1, mul_tiles_n = 1;
        unsigned mul_tiles_n_magic = 0;
        MulP mul_const;      // the problem's constants (m0 / n0 are the tile's)
        auto mul_problem = [&](const int tile) __attribute__((always_inline)) {
            int pi, begin;
            find_problem(tile, pi, begin);
            if (pi != mul_pi) {      // (no scalar reload - and no lgkmcnt(0) in the middle of the fragment pipeline - while the problem stays)
                const ConvP p = load_problem(pi);
                mul_pi = pi;
                mul_const.y = reinterpret_cast<f16_t*>(p.y);
                mul_const.bias = p.bias;
                mul_const.M = p.M; mul_const.Cout = p.Cout; mul_const.y_cstride = p.y_cstride; mul_const.y_coffset = p.y_coffset;
                mul_const.relu = p.flags & FCN_CONV_RELU;
                mul_const.tsh = p.lean_chunks;                                   // packed taps: chunk q of a slab starts at tap q << tsh
                mul_const.kw = (p.kw + (1 << p.lean_chunks) - 1) >> p.lean_chunks;      // chunks per slab
                mul_const.nch = p.kh * (p.lean_chunks ? 1 : (p.Cin + kChunkK - 1) / kChunkK) * mul_const.kw;
                mul_const.W = p.W; mul_const.PW = p.W + 2 * p.pad; mul_const.ow_magic = p.ow_magic;
                mul_tiles_n = p.tiles_n; mul_tiles_n_magic = p.tiles_n_magic;
            }
            const int lt = tile - begin;
            const int tile_m = fast_div(lt, mul_tiles_n_magic);
            MulP q = mul_const;
            q.m0 = tile_m * BM;
            q.n0 = (lt - tile_m * mul_tiles_n) * BN;
            return q;
        };
        // slab row of this lane's pixel in MFMA tile i: P(m) - P(m0); the tap offset q is added per chunk
        auto tile_rows = [&](const MulP& q, int (&rh)[WTM]) __attribute__((always_inline)) {
            const int t0 = fast_div(q.m0, q.ow_magic), x0 = q.m0 - t0 * q.W;
#pragma unroll
            for (int i = 0; i < WTM; ++i) {
                int m = q.m0 + wm * 32 * WTM + 32 * i + fi;
                m = m < q.M ? m : q.M - 1;      // (pixels past M: any row of the slab, the results are not stored)
                const int t = fast_div(m, q.ow_magic);
                rh[i] = (t - t0) * q.PW + (m - t * q.W) - x0;
            }
        };
        // bias of this lane's channels, the C operand of a tile's first MFMAs (out-of-range channels read 0; no bias: 0 records).
        // The loads are inline asm ON PURPOSE: the compiler's own wait in front of the first use was s_waitcnt vmcnt(0), which
        // also waits for the previous tile's 16 stores issued behind these loads - a write latency per tile.  bias_landed()
        // waits for exactly "all but the stores".
        typedef int v4i __attribute__((ext_vector_type(4)));
        auto load_bias = [&](const MulP& q) __attribute__((always_inline)) {
            const unsigned long long ba = (unsigned long long)(q.bias ? (const void*)q.bias : (const void*)q.y);
            const v4i rb = {(int)(unsigned)ba, (int)((unsigned)(ba >> 32) & 0xffffu), q.bias ? q.Cout * 4 : 0, 0x00020000};
#pragma unroll
            for (int j = 0; j < WTN; ++j) {
                const int c0 = q.n0 + wn * 32 * WTN + 32 * j + 8 * kh_;
#pragma unroll
                for (int g = 0; g < 4; ++g) {      // channels c0 + 0..3, 4..7, 16..19, 20..23
                    v4f b4;
                    const int off = (c0 + 4 * (g & 1) + 16 * (g >> 1)) * 4;
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(b4) : "v"(off), "s"(rb) : "memory");
#pragma unroll
                    for (int e = 0; e < 4; ++e) biasv[j][4 * g + e] = b4[e];
                }
            }
        };
        auto bias_landed = [&](auto stores_behind) __attribute__((always_inline)) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(stores_behind)::value) : "memory");
#pragma unroll
            for (int j = 0; j < WTN; ++j) asm volatile("" : "+v"(biasv[j]));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto epilogue = [&](const MulP& q) __attribute__((always_inline)) {
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
                    const int c0 = q.n0 + wn * 32 * WTN + 32 * j + 8 * kh_;
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
#ifdef FCN_EXP_NOSTORE
                        asm volatile("" ::"v"(pk), "v"(vo));
                        continue;
#endif
                        __builtin_amdgcn_raw_buffer_store_b128(pk, ry, vo, 0, 0);
                    }
                }
            }
        };

        SSTAMP_DECL;
        int tile = v;
        MulP cur = mul_problem(tile), nxt = cur;
        load_bias(cur);
        bias_landed(std::integral_constant<int, 0>{});
        tile_rows(cur, rho);
        unsigned abuf = 0, wslot = 0;      // byte offsets of the current chunk's slab / weight slot
        chunk_addr(rho, 0, abuf);
        __builtin_amdgcn_s_barrier();      // slab 0 and weight chunk 0 are in LDS
        asm volatile("" ::: "memory");
        read_w(0, 0, wslot);
#pragma unroll
        for (int i = 0; i < WTM; ++i) read_x(0, i, 0);
        // one chunk = 4 k-steps of a tap: [barrier] then per step { the next step's weight fragments; per pixel tile: wait, WTN MFMAs,
        // re-read the tile's fragment for the next step }.  ONE copy of this body in the kernel (the register allocator keeps the
        // accumulators in place across a single loop; four specialised copies of it made it shuffle and spill them).
        auto chunk = [&](const bool first, const int (&rh_next)[WTM], const int q_next, const unsigned abuf_next) __attribute__((always_inline)) {
            SSTAMP(0);                         // [0] chunk bodies (+ the loop around them)
            __builtin_amdgcn_s_barrier();      // the next chunk's weights (and slab) are in LDS; the loaders may refill what the previous chunk used
            asm volatile("" ::: "memory");
            SSTAMP(1);                         // [1] waiting at the barrier
            const unsigned wslot_next = wslot + WCH == (unsigned)(NB * WCH) ? 0u : wslot + WCH;
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                // the next step is step 0 of the NEXT chunk behind step 3 (its addresses replace this chunk's, whose reads are all issued)
                const int stn = st + 1 < 4 ? st + 1 : 0;
                if (st + 1 < 4) {
                    read_w((st + 1) & 1, stn, wslot);
                } else {
                    chunk_addr(rh_next, q_next, abuf_next);
                    read_w(0, 0, wslot_next);
                }
                if (XS == 2) {
#pragma unroll
                    for (int i = 0; i < WTM; ++i) read_x((st + 1) & 1, i, stn);
                }
                if (st == 0 && first) bias_landed(std::integral_constant<int, 2 * WTM * WTN>{});      // (behind the loads: the previous tile's stores)
                if (XS == 2) {
                    frag_landed(st & 1, 0, std::integral_constant<int, WTM + WTN>{});      // everything older than this step's reads: the step's operands
#pragma unroll
                    for (int i = 0; i < WTM; ++i) mfma_pair(st & 1, i, st == 0 && first);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int i = 0; i < WTM; ++i) {
                        frag_landed(st & 1, i, std::integral_constant<int, WTM - 1 + WTN>{});
                        mfma_pair(st & 1, i, st == 0 && first);
                        __builtin_amdgcn_sched_barrier(0);
                        read_x(0, i, stn);
                    }
                }
            }
            wslot = wslot_next;
        };
#pragma unroll 1
        while (true) {
            // the next tile's problem and pixel rows: the last chunk of this tile reads the first fragments of the next one
            const int tn = tile + G;
            const bool more = tn < total;
#pragma unroll
            for (int i = 0; i < WTM; ++i) rho_nt[i] = rho[i];
            SSTAMP(3);                         // [3] loop bookkeeping between tiles
            if (more) {
                nxt = mul_problem(tn);
                tile_rows(nxt, rho_nt);
            }
            SSTAMP(4);                         // [4] next tile's problem + pixel rows
            int q = 0;
#pragma unroll 1
            for (int ch = 0; ch < cur.nch; ++ch) {
                // the chunk after this one: the next tap of the slab, or tap 0 of the next slab (the next tile's behind the last chunk)
                const bool last_q = q + 1 == cur.kw;
                const int qn = last_q ? 0 : q + 1;
                const unsigned abuf_n = last_q ? (abuf + SLAB == (unsigned)(NA * SLAB) ? 0u : abuf + SLAB) : abuf;
                const bool tile_end = ch + 1 == cur.nch;
                if (tile_end && more) load_bias(nxt);
                int rsel[WTM];
#pragma unroll
                for (int i = 0; i < WTM; ++i) rsel[i] = tile_end ? rho_nt[i] : rho[i];
                chunk(ch == 0, rsel, qn << cur.tsh, abuf_n);
                q = qn;
                abuf = abuf_n;
            }
            SSTAMP(0);
            epilogue(cur);
            SSTAMP(2);                         // [2] epilogue
            if (!more) break;
            cur = nxt;
            tile = tn;
#pragma unroll
            for (int i = 0; i < WTM; ++i) rho[i] = rho_nt[i];
        }
        // the fragments prefetched last belong to a chunk nobody multiplies: retire the reads before the registers die
#pragma unroll
        for (int i = 0; i < WTM; ++i) frag_landed(0, i, std::integral_constant<int, 0>{});
        frag_landed(1, 0, std::integral_constant<int, 0>{});
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#ifdef FCN_STREAM_STAMPS
        st_[7] = SNOW() - st_t0_;
        st_[6] = __builtin_amdgcn_s_memtime() - st_c0_;      // shader cycles of the same span: the clock the chip held
        if (wid_all == 0) SSTAMP_FLUSH(0);
#endif
    }
#endif
}

// configurations: X(index, WTN, slab pieces, slab buffers, weight slots)
//   0/1: 3x3 and 5x5 launches (slabs of 304 rows, two buffers: a slab is needed kw >= 2 chunks after its issue)
//   2/3: 1x1 launches (256-row slabs, three / four buffers)       4/5: mixed 1x1 + 3x3 launches (288-row slabs, three buffers)
//   6: a whole inception level on 28-wide images - 3x3 + 5x5 + pool_proj in one launch (304-row slabs AND three buffers: 146 KiB
//      with 64-channel tiles; the 128-channel form would need 162)
//   7/8: 128 pixels x 192 channels (a wave multiplies 64 x 96: 96 accumulator registers) - for the layers with 192 / 384 output
//      channels (conv2/3x3, inception_3b/3x3, inception_5b/3x3), where 128-channel tiles leave a quarter of their MFMAs on padding;
//      7: 3x3 / 5x5 launches (160-row slabs, two buffers), 8: with 1x1 problems (three buffers)
//   9/10: 128 pixels x 128 / 64 channels, any filter size: twice the tiles of 4/5 - for launches of ~1.2 - 2.5 tiles per CU, where
//      a 256-pixel tile more or less on a CU is a quarter of the launch
#define FCN_STREAM_CONFIGS(X) \
    X(0, 2, 38, 2, 5, 4)      \
    X(1, 1, 38, 2, 5, 4)      \
    X(2, 2, 32, 3, 4, 4)      \
    X(3, 1, 32, 4, 4, 4)      \
    X(4, 2, 36, 3, 3, 4)      \
    X(5, 1, 36, 3, 4, 4)      \
    X(6, 1, 38, 3, 4, 4)      \
    X(7, 3, 20, 2, 5, 2)      \
    X(8, 3, 20, 3, 4, 2)      \
    X(9, 2, 20, 3, 4, 2)      \
    X(10, 1, 20, 4, 4, 2)

constexpr StreamCfgInfo kStreamCfgs[] = {
#define X(I, A, B, NA_, NB_, TM) {SCfg<A, B, NA_, NB_, TM>::BM, SCfg<A, B, NA_, NB_, TM>::BN, 32, SCfg<A, B, NA_, NB_, TM>::LDS_BYTES, SCfg<A, B, NA_, NB_, TM>::NT, B * 8, NA_},
    FCN_STREAM_CONFIGS(X)
#undef X
};

}  // namespace

#ifdef FCN_STREAM_STAMPS
extern "C" int fcn_debug_stream_stamps(void* d_buf, int cap) {
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stream_stamps), &d_buf, sizeof(d_buf)));
    FCN_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stream_stamps_cap), &cap, sizeof(cap)));
    return 0;
}
#endif

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
#define X(I, A, B, NA_, NB_, TM)                                                                                                                      \
    case I:                                                                                                                                           \
        hipLaunchKernelGGL((conv_stream_f16<SCfg<A, B, NA_, NB_, TM>>), dim3(grid), dim3(512), 0, st, ga.nprob, ga.tile_end[0], ga.tile_end[1],        \
                           ga.tile_end[2], ga.tile_end[3], ga.tile_end[4], ga.tile_end[5], ga.tile_end[6], ga.tile_end[7], total, 0, ga);              \
        break;
        FCN_STREAM_CONFIGS(X)
#undef X
    }
}

}  // namespace fcn
