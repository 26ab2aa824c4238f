// Types shared by the convolution kernels of libfcnhip.so (conv_fwd.hip: the tiled implicit-GEMM family; conv_stream.hip: the
// persistent half-float streaming kernel).  Internal: nothing here is part of the C ABI.
#pragma once

#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef _Float16 f16_t;
typedef _Float16 v8h __attribute__((ext_vector_type(8)));

namespace fcn {

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
    int lean_chunks;  // > 0: the scalar-addressed loader runs this many chunks (plan_tiles_cfg); 0: the per-lane loader
    int kw_magic;
    int M, K, tiles_m, tiles_n, tile_end;  // tile_end: exclusive prefix end of this problem's tiles in a group launch
    unsigned ow_magic, oh_magic;          // ceil(2^32 / OW), ceil(2^32 / OH): exact for every m < M (validate()); 0: OW / OH == 1
    const float* zero_page;               // 16 zero bytes in HBM: what out-of-image / out-of-tile lanes load
    unsigned tiles_n_magic;               // ceil(2^32 / tiles_n): tile / tiles_n without a division (tiles * tiles_n < 2^32), 0: tiles_n == 1
    unsigned cin_magic24;                 // ceil(2^24 / Cin): k / Cin for k < 256 (Cin <= 16384)
                                          // (160 bytes: the group kernel fetches a problem with three wide scalar loads)
};
static_assert(sizeof(ConvP) == 160, "conv_fwd_group loads a ConvP as 16 + 16 + 8 dwords");

// A group launch carries its problems in the kernel arguments: a workgroup finds its problem with scalar
// compares on the prefix table and ONE scalar load, instead of chasing a table in global memory (three or four
// dependent L2 round trips in front of the first LDS-DMA of a kernel that only runs for ~10 us).
constexpr int kMaxGroup = 8;
constexpr int kMaxPool = 2;
struct PoolP {
    const float* x;
    float* y;
    int* idx;
    int N, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride, y_coffset;
    int items, wg_end;     // float4 work items; exclusive prefix end of this pool's workgroups (after the conv tiles)
};
struct GroupArgs {
    int nprob;
    int tile_end[kMaxGroup];
    int npool;
    PoolP pool[kMaxPool];
    ConvP p[kMaxGroup];
};

// A "tail": narrow 1x1 problems (the detection heads cvg/classifier + bbox/regressor of models/deploy.prototxt: 4 + 16 outputs over the
// 1024 channels of inception_5b/output) evaluated by the launches that PRODUCE their input (conv_fwd.hip, conv_fwd_group<.., TAIL = true>):
// a tile that has just written 32 channels of a pixel block multiplies them by the heads' filters and leaves 4 * groups partial sums per
// pixel in scratch[slot = channel / 32][pixel][row]; the tile that arrives LAST at a pixel block (arrival word, no spinning) adds the
// K / 32 slots in slot order - a fixed order: bit-reproducible - applies bias / ReLU / sigmoid and writes the heads' outputs.  A
// launch with need == 0 only contributes partial sums (the branch of the module that rides in the earlier launch).
constexpr int kTailRows = 24;           // output channels of all narrow problems together, at most (whole fours)
constexpr int kTailMaxSlices = 4;
constexpr int FCN_CONV_TAILF = 1 << 20; // ConvP.flags (internal): this problem's output tiles contribute to the group's tail
struct TailSlice {
    const float* bias;
    float* y;
    float* y2;
    int o0, nout, y_cstride, y_coffset, y2_cstride, y2_coffset, flags, pad_;
};
struct TailHdr {                        // 96 bytes: conv_fwd_group fetches it with two wide scalar loads beside the problem's (ONE trip to the kernel arguments)
    float* scratch;                     // [K / 32][M][rows] partial sums
    unsigned* arrive;                   // one word per pixel block of the launch's tile height, zero between launches
    const float* gw[kTailRows / 4];     // filter rows 4j .. 4j+3: row r of group j at gw[j] + r * K
    int K, rows, need, M, nslices, dbg, pad1_, pad2_;
};
static_assert(sizeof(TailHdr) == 96, "conv_fwd_group loads a TailHdr as 16 + 8 dwords");
struct TailP {
    TailHdr h;
    TailSlice s[kTailMaxSlices];
};
struct GroupArgsTail {                  // what conv_fwd_group<.., TAIL = true> takes: the group, then its tail
    GroupArgs g;
    TailP tail;
};

// m / d through the host's multiplier ceil(2^32 / d) (exact while m * d < 2^32, which validate() / plan_tiles_cfg guarantee);
// magic 0 stands for d == 1.  No division fallback on purpose: an integer division is ~30 instructions, and the launch
// prologue is straight-line code that every workgroup runs once from a cold instruction cache.
__device__ __forceinline__ int fast_div(int m, unsigned magic) { return magic ? (int)__umulhi((unsigned)m, magic) : m; }



// swizzle of the 16-byte slots of staged row r: slot s of the LDS row holds k-segment s ^ swz(r).  An LDS-DMA
// wave-instruction writes 1 KiB lane-linearly, so the permutation is applied to the SOURCE address on the way in
// and to the slot index on the way out; 16 consecutive rows then hit 16 distinct 16-byte slots of the 256-byte
// bank row and every ds_read_b128 fragment read is conflict-free.
template <int SEGS>
__device__ __forceinline__ int swz(int row) { return SEGS == 4 ? (row >> 2) & 3 : SEGS == 8 ? (row >> 1) & 7 : row & 15; }

typedef float __attribute__((address_space(1))) * gf_ptr;
typedef const void __attribute__((address_space(1))) * gvoid_cptr;
typedef void __attribute__((address_space(3))) * lds_ptr;

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }


// ---- persistent half-float streaming kernel (conv_stream.hip): configurations 32 .. 32 + stream_num_cfgs() - 1 ---------------
struct StreamCfgInfo { int bm, bn, bk, lds_bytes, threads, slab_rows, slab_buffers; };      // bk counts 4-byte words, as in the tiled family
int stream_num_cfgs();
StreamCfgInfo stream_cfg_info(int idx);
// launches conv_stream_f16 over the prepared problems of `ga` (tile prefix, ConvP by value); total = number of tiles
void launch_stream(int idx, const GroupArgs& ga, int total, hipStream_t st);

}  // namespace fcn
