/*
 * fcnhip.h — C ABI of libfcnhip.so, the MI355X (gfx950) engine behind the
 * pycaffe-compatible shim in fcn_object_detector_amd/python/caffe.
 *
 * The reference has no FFI of its own for this path: its arithmetic lives in an
 * external Caffe install reached through pycaffe (reference:
 * scripts/fcn_object_detector.py:9,68-69,87,317 and
 * scripts/data_argumentation_layer/data_argumentation_layer.py:4,14) and through
 * the `caffe train` binary (reference: train/train.sh:25-28).  Each entry point
 * below names the Caffe/OpenCV call of the reference it stands in for.
 *
 * Conventions
 *   - plain C types only; every pointer is a DEVICE pointer unless its name starts
 *     with h_; shapes are int32; activations are NHWC float32 with an explicit
 *     channel stride (`*_cstride`, floats per pixel) so several producers can write
 *     channel slices of one buffer (Concat without a copy);
 *   - every function returns 0 on success, a negative value = -hipError_t, a positive
 *     value = argument-validation code (FCN_E_*); fcn_last_error_string() returns a
 *     thread-local message.  Nothing aborts, nothing prints;
 *   - everything is enqueued on the caller's stream (fcn_stream_t, may be NULL for the
 *     default stream) and is asynchronous unless the name ends in _sync;
 *   - the caller owns every buffer; the library keeps no pointer past a call except
 *     inside an fcn_graph_t it was asked to capture.
 */
#ifndef FCNHIP_H_
#define FCNHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FCN_ABI_VERSION 1

/* argument-validation codes (positive returns) */
#define FCN_E_ARG       1   /* null pointer / non-positive extent            */
#define FCN_E_ALIGN     2   /* channel count / stride / pointer not 16-B ok  */
#define FCN_E_UNSUPPORTED 3 /* geometry the kernels do not implement          */
#define FCN_E_STATE     4   /* call not valid in the current state            */
#define FCN_E_CAPACITY  5   /* output capacity too small                      */

typedef void* fcn_stream_t;
typedef void* fcn_event_t;
typedef void* fcn_graph_t;
typedef void* fcn_comm_t;

/* ---- runtime: replaces caffe.set_device / set_mode_gpu (fcn_object_detector.py:68-69)
 *      and Caffe's SyncedMemory allocation + H<->D copies behind blob.data ---- */
int  fcn_abi_version(void);
const char* fcn_last_error_string(void);
int  fcn_device_count(int* count);
int  fcn_init(int device);                 /* per-thread hipSetDevice; idempotent, thread-safe */
int  fcn_device_name(char* h_buf, int len);
int  fcn_device_sync(void);
int  fcn_malloc(void** p, size_t bytes);
int  fcn_free(void* p);
int  fcn_host_malloc(void** h_p, size_t bytes);   /* pinned host memory for blob.data views */
int  fcn_host_free(void* h_p);
int  fcn_memset_async(void* p, int value, size_t bytes, fcn_stream_t s);
int  fcn_memcpy_h2d_async(void* dst, const void* h_src, size_t bytes, fcn_stream_t s);
int  fcn_memcpy_d2h_async(void* h_dst, const void* src, size_t bytes, fcn_stream_t s);
int  fcn_memcpy_d2d_async(void* dst, const void* src, size_t bytes, fcn_stream_t s);
int  fcn_stream_create(fcn_stream_t* s);
int  fcn_stream_destroy(fcn_stream_t s);
int  fcn_stream_sync(fcn_stream_t s);
int  fcn_event_create(fcn_event_t* e);
int  fcn_event_destroy(fcn_event_t e);
int  fcn_event_record(fcn_event_t e, fcn_stream_t s);
int  fcn_event_sync(fcn_event_t e);
int  fcn_event_elapsed_ms(fcn_event_t start, fcn_event_t stop, float* h_ms);
int  fcn_stream_wait_event(fcn_stream_t s, fcn_event_t e);   /* later work on s waits for e (overlap of collectives) */
/* hipGraph capture of a whole Net::Forward / ForwardBackward launch sequence */
int  fcn_graph_begin(fcn_stream_t s);
int  fcn_graph_end(fcn_stream_t s, fcn_graph_t* g);
int  fcn_graph_launch(fcn_graph_t g, fcn_stream_t s);
int  fcn_graph_destroy(fcn_graph_t g);

/* ---- blob layout at the pycaffe boundary (blob.data is NCHW) ---- */
/* dst[n,h,w,dst_coffset + c] = src[n,c,h,w] + shift; dst channel stride dst_cstride.  `shift` lets the
 * upload of the net input absorb a following Power(shift) layer (models/deploy.prototxt:8-16). */
int  fcn_nchw_to_nhwc_f32(const float* src, float* dst, int N, int C, int H, int W,
                          int dst_cstride, int dst_coffset, float shift, fcn_stream_t s);
int  fcn_nhwc_to_nchw_f32(const float* src, float* dst, int N, int C, int H, int W,
                          int src_cstride, int src_coffset, fcn_stream_t s);
/* several (small) blobs in ONE launch - net.forward() hands back both head blobs of models/deploy.prototxt with it (up to 8 blobs) */
typedef struct fcn_layout_desc {
    const float* src;   /* NHWC, channel stride src_cstride, the blob's channels at src_coffset .. */
    float* dst;         /* NCHW, dense */
    int32_t N, C, H, W, src_cstride, src_coffset;
} fcn_layout_desc;
int  fcn_nhwc_to_nchw_multi_f32(const fcn_layout_desc* h_descs, int n, fcn_stream_t s);

/* ---- Convolution (+bias, fused in-place ReLU / Sigmoid):
 *      Caffe ConvolutionLayer::Forward_gpu, ReLULayer, SigmoidLayer as run by
 *      net.forward() (fcn_object_detector.py:87) over models/deploy.prototxt:8-2176 ---- */
#define FCN_CONV_RELU      1   /* y = max(y, 0)                                  */
#define FCN_CONV_SIGMOID2  2   /* y2 = sigmoid(y) is written as well (y2 != NULL) */
#define FCN_CONV_ACCUM     4   /* y += result (gradient fan-in when the kernel runs as a data-gradient pass) */
#define FCN_CONV_MASK     64   /* y = (y2 > 0) ? result : 0 with y2 read at the result's position (y2_cstride / y2_coffset): the ReLU
                                * backward of the layer below, applied by the LAST data-gradient pass that writes its gradient   */
#define FCN_CONV_IMAGE_ONES 128 /* with FCN_CONV_F16 on an 8-half pixel image (the first layer of an f16 net): the caller promises that channels 3
                                * and 4 of x hold the constant 1 at every pixel and channels 5..7 contribute nothing (zero pixels or zero
                                * weights) - the folded Power shift of models/deploy.prototxt:8-16.  Their products are then added
                                * as per-tap constants (f32) instead of being multiplied; pixels outside the image count as 0, as ever */
#define FCN_CONV_OUT_F32   8   /* with FCN_CONV_F16: y is float32 (the detection heads feed the f32 decode kernel)   */
#define FCN_CONV_OUT_F16  32   /* float32 x and w, y stored as half floats: the first layer of an f16 net keeps its input in
                                * float32 (models/deploy.prototxt shifts a [0,1] image by -127: 16 half-float levels)   */
#define FCN_CONV_F16      16   /* x, w and y hold IEEE half floats (v_mfma_f32_32x32x16_f16, f32 accumulate, f32 bias):
                                * BASELINE configs[4].  Cin and x_cstride must then be multiples of 8; the pointers of
                                * the descriptor are typed float* for both element types */
typedef struct fcn_conv_desc {
    const float* x;      /* NHWC input, channel stride x_cstride                           */
    const float* w;      /* weights [Cout][kh][kw][Cin]  (OHWI, Cin contiguous)            */
    const float* bias;   /* [Cout] or NULL                                                 */
    float*       y;      /* NHWC output; element (m, n) at y[m*y_cstride + y_coffset + n]  */
    float*       y2;     /* second output for FCN_CONV_SIGMOID2 (same indexing via y2_*)   */
    int32_t N, H, W, Cin, x_cstride;
    int32_t Cout, kh, kw, pad, stride, OH, OW;
    int32_t y_cstride, y_coffset, y2_cstride, y2_coffset;
    int32_t flags;
    float   in_shift;    /* reserved, must be 0 (a Power(shift) input transform is applied by the producer of x) */
} fcn_conv_desc;
/* one problem */
int  fcn_conv2d_fwd_f32(const fcn_conv_desc* h_desc, fcn_stream_t s);
/* n independent problems in ONE launch (the branches of an inception module).  prepare() validates
 * and uploads the problems into d_workspace (fcn_conv2d_group_workspace_bytes(n) bytes, owned by
 * the caller, alive as long as the group is used) with a synchronous copy - call it at plan time,
 * not inside a graph capture; the launch itself is a pure kernel launch and can be captured. */
typedef struct fcn_conv_group {
    void*   d_probs;
    int32_t n;
    int32_t cfg;          /* tile configuration chosen by prepare() */
    int32_t total_tiles;
} fcn_conv_group;
size_t fcn_conv2d_group_workspace_bytes(int n);
/* cfg_request: -1 = built-in heuristic, 0 .. fcn_conv2d_num_configs()-1 = that tile configuration (the engine
 * times every configuration once per launch at plan time and keeps the fastest) */
int  fcn_conv2d_num_configs(void);
/* The last two configurations are not tile shapes of the implicit-GEMM kernel but shape-specific kernels; prepare() returns
 * FCN_E_UNSUPPORTED when one is requested for a group it does not take (a tuner walking all configurations skips those):
 *   fcn_conv2d_first_layer_config()      conv_first7_kernel: a single 7x7 / stride 2 / pad 3 problem on 4-channel pixels with
 *                                        33..64 output channels (conv1/7x7_s2 of models/deploy.prototxt), ReLU optional; the
 *                                        built-in heuristic picks it for the problems it takes;
 *   fcn_conv2d_first_layer_config() + 1  conv_dot1x1_kernel: groups of 1x1 / stride 1 / unpadded float32 problems over the same
 *                                        pixels with at most 32 output channels in all, counted in slices of 8 per problem (the
 *                                        detection heads cvg/classifier + bbox/regressor); ReLU and FCN_CONV_SIGMOID2 allowed. */
int  fcn_conv2d_first_layer_config(void);
/* A "tail": narrow 1x1 problems (the detection heads cvg/classifier + bbox/regressor, deploy.prototxt:2363-2410 - 4 + 16 outputs over the
 * 1024 channels of inception_5b/output) evaluated BY THE LAUNCHES THAT PRODUCE THEIR INPUT instead of a launch of their own: every tile
 * that has written 32 channels of a block of pixels leaves their contribution to the narrow outputs in `scratch`, and the tile that
 * arrives last at a pixel block adds the contributions in channel order (a fixed order: bit-reproducible), applies bias / ReLU /
 * sigmoid and writes the narrow problems' outputs.  Announce the tail for a workspace BEFORE fcn_conv2d_group_prepare[_fused] on that
 * workspace; every problem of the group whose y is heads[0].x then contributes (it must be a float32 bias + ReLU problem over the same
 * pixels writing whole 32-channel groups of that blob).  finalize = 1 for the launch that completes the blob, 0 for an earlier launch
 * that writes part of it (partial sums only).  Tile configurations 23, 24, 25 and 27 have a tail variant; prepare() returns
 * FCN_E_UNSUPPORTED for the others.  heads: float32 1x1 / stride 1 problems reading the same blob, outputs in whole groups of four
 * channels, at most 24 in all; flags within FCN_CONV_RELU | FCN_CONV_SIGMOID2.  scratch / arrive: device memory of
 * fcn_conv2d_tail_scratch_bytes() / fcn_conv2d_tail_arrive_bytes() bytes, arrive zeroed once by the caller (launches leave it zero);
 * two launches that share them must not overlap in time.  fcn_conv2d_group_attach_tail(ws, NULL) and fcn_conv2d_group_release(ws) forget it. */
typedef struct fcn_conv_tail {
    int32_t n;                 /* 1..4 narrow problems */
    int32_t finalize;
    fcn_conv_desc heads[4];
    float*  scratch;
    void*   arrive;
} fcn_conv_tail;
size_t fcn_conv2d_tail_scratch_bytes(const fcn_conv_tail* t);
size_t fcn_conv2d_tail_arrive_bytes(const fcn_conv_tail* t);
int  fcn_conv2d_group_attach_tail(void* d_workspace, const fcn_conv_tail* t);
/* LDS bytes one workgroup of that configuration holds (a CU has 160 KiB: it bounds how many workgroups - of this or of a
 * concurrent launch on another stream - fit on a CU); -1 for an unknown index */
int  fcn_conv2d_config_lds_bytes(int cfg);
/* how many waves of a workgroup split K and reduce through LDS in that configuration (1 = no K split); -1 for an unknown index */
int  fcn_conv2d_config_waves_k(int cfg);
int  fcn_conv2d_group_prepare(const fcn_conv_desc* h_descs, int n, void* d_workspace, int cfg_request, fcn_conv_group* h_out);
int  fcn_conv2d_fwd_group_f32(const fcn_conv_group* h_group, fcn_stream_t s);
/* MAX poolings that read the same bottoms as the group's convolutions (an inception module's 3x3 stride-1 pool beside
 * its 1x1 convolutions) can ride in the group's launch as extra workgroups instead of a launch of their own.
 * Needs C, x_cstride, y_cstride, y_coffset multiples of 4 and 16-byte aligned pointers; idx may be NULL. */
typedef struct fcn_pool_desc {
    const float* x; float* y; int32_t* idx;
    int32_t N, H, W, C, x_cstride, k, stride, pad, OH, OW, y_cstride, y_coffset;
    int32_t f16;           /* 1: x / y hold half floats (C and the strides then multiples of 8), must match the group's convolutions */
} fcn_pool_desc;
int  fcn_conv2d_group_prepare_fused(const fcn_conv_desc* h_descs, int n, const fcn_pool_desc* h_pools, int npools, void* d_workspace,
                                    int cfg_request, fcn_conv_group* h_out);
/* The library keeps a host copy of every prepared group, keyed by its d_workspace.  Call this before freeing (or reusing for
 * something else) a workspace that was handed to a prepare call: the entry is erased, so that a later allocation that happens
 * to get the same device address can never pick up a stale plan.  Unknown pointers are ignored (returns 0). */
int  fcn_conv2d_group_release(void* d_workspace);

/* ---- Pooling / LRN / pointwise: Caffe PoolingLayer, LRNLayer, EltwiseLayer ---- */
/* MAX pool, ceil-mode output size computed by the caller (OH, OW), window clipped to the
 * image, first maximum in raster order wins; idx (may be NULL) receives iy*W+ix per output */
int  fcn_maxpool_fwd_f32(const float* x, float* y, int32_t* idx, int N, int H, int W, int C,
                         int x_cstride, int k, int stride, int pad, int OH, int OW,
                         int y_cstride, int y_coffset, fcn_stream_t s);
int  fcn_avepool_fwd_f32(const float* x, float* y, int N, int H, int W, int C, int x_cstride,
                         int k, int stride, int pad, int OH, int OW, int y_cstride, int y_coffset,
                         fcn_stream_t s);
/* LRN ACROSS_CHANNELS: y = x * (k + alpha/n * sum x^2)^-beta ; scale (may be NULL) keeps the base */
int  fcn_lrn_fwd_f32(const float* x, float* y, float* scale, int pixels, int C, int x_cstride,
                     int y_cstride, int local_size, float alpha, float beta, float k, fcn_stream_t s);
/* MAX pooling and LRN (ACROSS_CHANNELS, local_size 5) of one blob in a single pass, inference only: lrn_first 0 computes
 * LRN(maxpool(x)) (deploy.prototxt pool1/3x3_s2 -> pool1/norm1), 1 computes maxpool(LRN(x)) (conv2/norm2 -> pool2/3x3_s2);
 * the blob between the two layers is never written.  C, the strides and the pointers must allow 16-byte channel groups
 * (FCN_E_UNSUPPORTED otherwise: run fcn_maxpool_fwd_f32 and fcn_lrn_fwd_f32). */
int  fcn_maxpool_lrn5_fwd_f32(const float* x, float* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                              int OH, int OW, int y_cstride, int lrn_first, float alpha, float beta, float lrn_k, fcn_stream_t s);
/* pool -> LRN -> 1x1 convolution (+ bias, optional ReLU) in ONE launch: deploy.prototxt pool1/3x3_s2 -> pool1/norm1 ->
 * conv2/3x3_reduce (:54-104).  y[pixel][y_coffset + co] = act(bias[co] + sum_c w[co][c] * LRN(maxpool(x))[pixel][c]); w is
 * [Cout][C] row-major.  Neither the pooled nor the normalised blob is written.  3 x 3 windows and 64 -> 64 channels only
 * (FCN_E_UNSUPPORTED otherwise: fcn_maxpool_lrn5_fwd_f32 + fcn_conv2d_fwd_f32).  The pooling is exact; the LRN takes s^-0.75 from the
 * hardware reciprocal square root / square root (1 ulp each: within 1e-6 of fcn_maxpool_lrn5_fwd_f32(lrn_first = 0)); the convolution
 * sums K on the matrix cores in its own order (float32). */
int  fcn_maxpool_lrn5_conv1x1_fwd_f32(const float* x, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                                      int OH, int OW, float alpha, float beta, float lrn_k, const float* w, const float* bias,
                                      int Cout, int relu, float* y, int y_cstride, int y_coffset, fcn_stream_t s);
/* the half twin (x, w, y hold halves, bias float32; 3 x 3 / stride 2 / unpadded windows only): the LDS-patch form of
 * fcn_maxpool_lrn5_fwd_f16 whose last step multiplies the normalised tile by the filter bank (v_mfma_f32_16x16x16_f16, f32 accumulate,
 * one rounding after bias and ReLU) */
int  fcn_maxpool_lrn5_conv1x1_fwd_f16(const void* x, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                                      int OH, int OW, float alpha, float beta, float lrn_k, const void* w, const float* bias,
                                      int Cout, int relu, void* y, int y_cstride, int y_coffset, fcn_stream_t s);
int  fcn_relu_fwd_f32(const float* x, float* y, size_t count, float negative_slope, fcn_stream_t s);
int  fcn_sigmoid_fwd_f32(const float* x, float* y, size_t count, fcn_stream_t s);
int  fcn_power_fwd_f32(const float* x, float* y, size_t count, float power, float scale, float shift, fcn_stream_t s);
#define FCN_ELT_PROD 0
#define FCN_ELT_SUM  1
#define FCN_ELT_MAX  2
/* y = a (op) b over `count` contiguous floats; SUM uses coefficients ca, cb */
int  fcn_eltwise_fwd_f32(const float* a, const float* b, float* y, size_t count, int op,
                         float ca, float cb, fcn_stream_t s);
/* copies C channels of every pixel between strided NHWC buffers (Concat / Slice fallback) */
int  fcn_copy_channels_f32(const float* src, float* dst, int pixels, int C, int src_cstride,
                           int src_coffset, int dst_cstride, int dst_coffset, fcn_stream_t s);
/* grouped bilinear-style Deconvolution, group == channels, one filter [k][k] per channel:
 * Caffe DeconvolutionLayer with `group: C` as in train/fcn_bbox/train_val.prototxt:544-565 */
/* Softmax over the channels of every pixel (Caffe SoftmaxLayer, axis 1) */
int  fcn_softmax_fwd_f32(const float* x, float* y, int pixels, int C, int x_cstride, int y_cstride, fcn_stream_t s);
int  fcn_deconv_depthwise_fwd_f32(const float* x, const float* w, const float* bias, float* y,
                                  int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                                  int OH, int OW, int y_cstride, int y_coffset, fcn_stream_t s);

/* ---- inference pre-processing: demean_rgb_image + cv.resize + HWC->CHW
 *      (fcn_object_detector.py:79-82, 407-413) ---- */
/* h/w x 3 uint8 BGR frame -> NHWC float32 (C padded to dst_cstride) in [0,1]:
 * (px - mean[c] - min) / (max - min) over the whole frame, then bilinear resize to (H, W), then + shift
 * (the net's Power(shift) input transform, models/deploy.prototxt:8-16; pass 0 to get the blob itself).
 * d_minmax is a 32-byte device scratch (per-channel uint8 min and max, as int32). */
int  fcn_preprocess_bgr8(const uint8_t* frame, int h, int w, float* dst, int H, int W, int dst_cstride,
                         float shift, float* d_minmax, fcn_stream_t s);

/* ---- half-float activation path (BASELINE configs[4]: batched inference with f16 storage, f32 accumulation).
 *      Convolutions take FCN_CONV_F16 in fcn_conv_desc.flags; these are the layout converters and the other layers
 *      of models/deploy.prototxt in that element type.  Channel counts / strides are multiples of 8 (16 bytes). ---- */
int  fcn_nchw_f32_to_nhwc_f16(const float* src, void* dst, int N, int C, int H, int W, int dst_cstride, int dst_coffset, float shift,
                              fcn_stream_t s);
int  fcn_nhwc_f16_to_nchw_f32(const void* src, float* dst, int N, int C, int H, int W, int src_cstride, int src_coffset, fcn_stream_t s);
int  fcn_maxpool_fwd_f16(const void* x, void* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad, int OH, int OW,
                         int y_cstride, int y_coffset, fcn_stream_t s);
int  fcn_lrn_fwd_f16(const void* x, void* y, int pixels, int C, int x_cstride, int y_cstride, int local_size, float alpha, float beta,
                     float k, fcn_stream_t s);
/* the half twin of fcn_maxpool_lrn5_fwd_f32 (8-channel groups).  Against fcn_maxpool_fwd_f16 + fcn_lrn_fwd_f16 run one after the other
 * (either order): 3x3 / stride 2 / pad 0 poolings of at most 192 channels take an LDS-patch kernel whose LRN differs from the stand-alone
 * one in the last float32 bit of scale^-beta on a few elements - fewer than 1e-4 of the outputs differ, each by ONE f16 ulp
 * (tests/test_gpu_f16.py asserts exactly that bound); every other geometry is bit-identical to the two launches. */
int  fcn_maxpool_lrn5_fwd_f16(const void* x, void* y, int N, int H, int W, int C, int x_cstride, int k, int stride, int pad,
                              int OH, int OW, int y_cstride, int lrn_first, float alpha, float beta, float lrn_k, fcn_stream_t s);
/* n windows of ONE frame -> the n images of an N x H x W x dst_cstride blob: the node's multi-window path
 * (scripts/fcn_object_detector.py run_detector2 :198-211 with detection_window_roi :257-277) demeans and normalises the WHOLE frame
 * (min / max over the frame), crops stride x stride windows plus a central one, and resizes each to the net's input.  h_rois: n x
 * (x, y, w, h) int32 on the HOST, every window inside the frame, n <= 32; d_minmax: 32 bytes.  Same arithmetic as
 * fcn_preprocess_bgr8 (float64, float resize coefficients). */
int  fcn_preprocess_bgr8_rois(const uint8_t* frame, int h, int w, const int32_t* h_rois, int n, void* dst, int dst_f16, int H, int W,
                              int dst_cstride, float shift, float* d_minmax, fcn_stream_t s);
/* dst_f16 of the two calls below: 0 = float32 blob, 1 = half blob (channels 0..2 of every pixel are written, nothing else), 3 = the
 * half image of an f16 engine - 8-half pixels whose channels 3 and 4 hold the constant 1 and 5..7 zero (FCN_CONV_IMAGE_ONES): the whole
 * pixel (b, g, r, 1, 1, 0, 0, 0) leaves in ONE 16-byte store (dst_cstride must be 8, dst 16-byte aligned). */
/* n equally sized frames (h*w*3 bytes apart) -> the n images of an N x H x W x dst_cstride blob in three launches;
 * each frame is normalised with its own min / max.  d_minmax: 32 bytes per frame. */
int  fcn_preprocess_bgr8_batch(const uint8_t* frames, int n, int h, int w, void* dst, int dst_f16, int H, int W, int dst_cstride,
                               float shift, float* d_minmax, fcn_stream_t s);
int  fcn_preprocess_bgr8_f16(const uint8_t* frame, int h, int w, void* dst, int H, int W, int dst_cstride, float shift, float* d_minmax,
                             fcn_stream_t s);

/* ---- training-scene synthesis on the device: the pixel work of ArgumentationEngineMapping.argument
 *      (scripts/data_argumentation_layer/argumentation_engine.py:651-746) and the whole-image flip of
 *      random_argumentation (:143-188).  The random decisions stay on the host; images live in HBM. ---- */
typedef struct fcn_scene_obj {
    const uint8_t* img;    /* src_h x src_w x 3 BGR, as stored (unflipped)                                   */
    const uint8_t* mask;   /* src_h x src_w, nonzero = object                                                */
    int32_t src_h, src_w;
    int32_t flip;          /* cv.flip code applied to the object before cropping: 0, 1, -1; anything else: none */
    int32_t roi_x, roi_y, roi_w, roi_h;   /* crop in the flipped image                                          */
    int32_t out_w, out_h;  /* size after the optional bilinear rescale                                        */
    int32_t cx, cy;        /* paste position in the scene (may be negative / hang over the border)            */
    int32_t label1;        /* value written to the class mask (label + 1)                                     */
} fcn_scene_obj;
/* out_img (H x W x 3) = bilinear resize of the background crop, then the objects pasted in order where their (resized)
 * mask is nonzero, then the whole-image flip `final_flip` (0, 1, -1; else none); out_mask (H x W, may be NULL) = label1
 * of the last object covering the pixel, 0 elsewhere.  d_objs: nobj records in device memory. */
int  fcn_compose_scene_bgr8(const uint8_t* bg, int bg_h, int bg_w, int crop_x, int crop_y, int crop_w, int crop_h,
                            const fcn_scene_obj* d_objs, int nobj, int final_flip, uint8_t* out_img, uint8_t* out_mask,
                            int H, int W, fcn_stream_t s);
/* The same scene seen through the window (view_x, view_y, view_w, view_h) of the flipped H x W scene: out_img is
 * view_h x view_w x 3, out_mask view_h x view_w; either may be NULL.  This is the "zoom in" crop of random_argumentation
 * (argumentation_engine.py:156-172, crop_image_dimension :190-236), which crops the image but not the class mask. */
int  fcn_compose_scene_view_bgr8(const uint8_t* bg, int bg_h, int bg_w, int crop_x, int crop_y, int crop_w, int crop_h,
                                 const fcn_scene_obj* d_objs, int nobj, int final_flip, uint8_t* out_img, uint8_t* out_mask,
                                 int H, int W, int view_x, int view_y, int view_w, int view_h, fcn_stream_t s);

/* ---- colour augmentation of a composed scene: color_space_argumentation (argumentation_engine.py:308-322), an
 *      imgaug Sequential restated from the operators' documented definitions (imgaug itself is an un-vendored submodule).
 *      All images are h x w x 3 uint8, every stage rounds half-to-even and saturates to uint8 like the library's
 *      uint8 pipeline.  src and dst must not alias. ---- */
#define FCN_GAUSS_MAX_RADIUS 15
/* GaussianBlur: separable, half-kernel taps[0..radius] (centre first, float32, normalised by the caller), reflect-101
 * border; tmp: h*w*3 floats of scratch */
int  fcn_blur_gauss_bgr8(const uint8_t* src, uint8_t* dst, float* tmp, int h, int w, const float* h_taps, int radius, fcn_stream_t s);
/* AverageBlur (cv2.blur): k x k box, anchor k/2, reflect-101 border, round(sum * (1.0 / k^2)) in double; 1 <= k <= 15 */
int  fcn_blur_box_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, int k, fcn_stream_t s);
/* MedianBlur (cv2.medianBlur): per-channel median of the k x k window, replicated border; k in {3, 5, 7} */
int  fcn_blur_median_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, int k, fcn_stream_t s);
typedef struct fcn_color_params {
    float sharpen_centre;  /* (1 - a) + a * (8 + lightness): centre of the 3x3 Sharpen matrix            */
    float sharpen_off;     /* -a: its eight other entries (reflect-101 border)                           */
    int32_t add[3];        /* Add: per-channel integer offsets                                           */
    float mul[3];          /* Multiply: per-channel factors                                              */
    float gray_alpha;      /* Grayscale: out = gray_keep * v + gray_alpha * grey                         */
    float gray_keep;       /* 1 - gray_alpha, rounded to float32 by the caller                           */
} fcn_color_params;
/* Sharpen -> Add -> Multiply -> Grayscale in one pass.  grey = (4899 c0 + 9617 c1 + 1868 c2 + 8192) >> 14: OpenCV's
 * RGB2GRAY fixed point applied to the channels in storage order, as imgaug does to the reference's BGR image. */
int  fcn_color_augment_bgr8(const uint8_t* src, uint8_t* dst, int h, int w, const fcn_color_params* h_params, fcn_stream_t s);

/* class mask (h x w uint8) -> H x W label blob, one float per pixel at stride dst_cstride (nearest neighbour: top[1] of
 * the data layer in HEAD's mask mode, data_argumentation_layer.py:113-121) */
int  fcn_mask_to_label_f32(const uint8_t* mask, int h, int w, float* dst, int H, int W, int dst_cstride, fcn_stream_t s);

/* ---- DetectNet post-processing: gridbox_to_boxes + vote_boxes -> cv.groupRectangles
 *      (fcn_object_detector.py:337-394; OpenCV 3 objdetect groupRectangles/partition) ---- */
#define FCN_RECT_ROUND_NEAREST_EVEN 0  /* OpenCV vector<Rect> converter: saturate_cast<int>(double) = cvRound */
#define FCN_RECT_ROUND_TRUNCATE     1  /* C (int) cast                                                          */
typedef struct fcn_detect_params {
    int32_t num_classes;     /* C : coverage channels decoded                                  */
    int32_t gy, gx;          /* grid                                                            */
    int32_t cell_w, cell_h;  /* im_sz / grid (integer division, fcn_object_detector.py:368-369) */
    int32_t cvg_cstride, cvg_coffset;   /* NHWC coverage map: cvg[(y*gx+x)*cvg_cstride + cvg_coffset + c] */
    int32_t box_cstride, box_coffset;   /* NHWC bbox map, channels 4c..4c+3 of class c                     */
    float   prob_thresh;     /* ~detection_threshold (0.5)   */
    int32_t group_thresh;    /* ~min_boxes (3)               */
    double  eps;             /* ~nms_eps (0.2); OpenCV takes it as a double */
    int32_t min_height;      /* 20: keep if rect[3]-rect[1] >= min_height (fcn_object_detector.py:346) */
    int32_t round_mode;      /* FCN_RECT_ROUND_*             */
    int32_t max_out;         /* capacity of the output arrays per (image, class) slot */
} fcn_detect_params;
/* One to eight workgroups per (image, class) - the SimilarRects tests of a problem with many candidates are dealt to
 * several workgroups whose forests the last one to arrive merges (small batches only: a full batch fills the chip with one
 * per problem); slot = image * num_classes + class.  Zero-fill the workspace once after allocating it (the arrival words
 * carry a launch tag and are never reset; an uninitialised word matches a live tag with probability 2^-24) and do not
 * share it between launches that may run concurrently.  Outputs per slot:
 * out_rects[slot][max_out][4] int32 (x, y, w, h exactly as groupRectangles returns them, in its
 * cluster order), out_weights[slot][max_out] int32 (cluster size n; the reference's confidence is
 * log(n)), out_count[slot] int32 (may exceed max_out: then only max_out entries were stored).
 * Concatenating the slots of one image in class order reproduces the reference's nested loops
 * (fcn_object_detector.py:104-118).  Any grid; at most 5120 CANDIDATES (cells at or above prob_thresh) per (image, class) -
 * enough for every cell of a 640 x 480 frame at stride 8 - beyond which that slot's out_count is -1 and nothing else of the
 * slot is written.  d_workspace:
 * fcn_detect_workspace_bytes() bytes; image strides are in floats. */
size_t fcn_detect_workspace_bytes(const fcn_detect_params* h_p, int batch);
int  fcn_detect_decode_group(const float* cvg, const float* bbox, int batch,
                             size_t cvg_image_stride, size_t box_image_stride,
                             const fcn_detect_params* h_p, void* d_workspace,
                             int32_t* out_rects, int32_t* out_weights,
                             int32_t* out_count, fcn_stream_t s);

/* ---- run_detector2 after net.forward(): the node's score maps -> the frame-sized probability map and one box per (window, class)
 *      (scripts/fcn_object_detector.py:208-236 with create_mask_labels :279-303; OpenCV's cv.resize / findContours / contourArea /
 *      boundingRect restated in oracle/mask_ref.py).  score: NHWC float32 blob of N windows (the node's net.blobs['score']), classes at
 *      channels coffset .. coffset + C - 1 of cstride; class 0 is the background and is skipped, as in the reference.  h_rects: HOST
 *      array N x (x, y, w, h), the windows' places in the frame - all of one size (detection_window_roi :257-277), N <= 32.
 *      For every window n and class c = 1 .. C-1: values below prob_thresh become 0, x 255, bilinear resize to (w, h), truncating
 *      uint8 cast; the map is OR-ed into pmap (frame_h x frame_w bytes on the device, 4-byte aligned, zeroed by the caller) at the
 *      window's place, and out[(n * (C - 1) + c - 1) * 5 ..] = found, x, y, w, h: the bounding rectangle, in WINDOW coordinates, of the
 *      contour with the largest area (found = 0: no contour of positive area; the reference's 10-pixel padding and the window's origin
 *      are added by the caller).  d_workspace: fcn_score_masks_workspace_bytes() bytes. ---- */
size_t fcn_score_masks_workspace_bytes(int n_windows, int num_classes, int w, int h);
int  fcn_score_masks(const float* score, int N, int C, int H, int W, int cstride, int coffset, const int32_t* h_rects, float prob_thresh,
                     uint8_t* pmap, int frame_h, int frame_w, void* d_workspace, int32_t* out, fcn_stream_t s);

/* ---- DetectNet target generation: ArgumentationEngine.bounding_box_parameterized_labels
 *      (argumentation_engine.py:69-109, 26-55, 272-292) ---- */
/* rects: [total][4] int32 (x, y, w, h); labels: [total] int32; rect_offsets: [batch+1] int32 prefix.
 * Outputs are NCHW float32 exactly as the Python layer's tops (data_argumentation_layer.py:67-72):
 * foreground (batch, C, gy, gx); bbox/size/obj/cvg_block (batch, 4C, gy, gx). */
int  fcn_gen_targets(const int32_t* rects, const int32_t* labels, const int32_t* rect_offsets, int batch,
                     int num_classes, int gy, int gx, int stride, double iou_thresh,
                     float* foreground, float* bbox, float* size, float* obj, float* cvg_block,
                     fcn_stream_t s);
/* same arithmetic, written straight into the engine's NHWC blob buffers (element (img, cell, channel) at
 * [(img*gy*gx + cell) * cstride + channel]) so a training step needs no host round trip for its labels */
int  fcn_gen_targets_nhwc(const int32_t* rects, const int32_t* labels, const int32_t* rect_offsets, int batch,
                          int num_classes, int gy, int gx, int stride, double iou_thresh,
                          float* foreground, int fg_cstride, float* bbox, float* size, float* obj, float* cvg_block,
                          int blk_cstride, fcn_stream_t s);

/* ---- training: Net::Backward + losses + solver update as run by `caffe train` (train/train.sh:25-28) over the
 *      loss tail models/train_val.prototxt:53-72,2237-2281 with the settings of train/<net>/solver.prototxt ---- */
/* Weight / bias gradient of a Convolution layer.  `d` describes the FORWARD problem; d->y / y_cstride / y_coffset name the
 * gradient of the layer's output (d->w, d->bias, d->y2 are ignored).  dw is [Cout][kh][kw][Cin] like the forward weights,
 * db is [Cout] or NULL.  Bit-reproducible (pixel splits are summed in a fixed order).  Workspace size in floats: */
size_t fcn_conv2d_wgrad_workspace_floats(const fcn_conv_desc* h_d, int* h_splits);
int  fcn_conv2d_wgrad_f32(const fcn_conv_desc* h_d, float* dw, float* db, float* d_workspace, fcn_stream_t s);
/* Up to 4 layers in ONE launch (+ one grouped fixed-order reduction): the output convolutions of an inception module become
 * ready together and most of them are too small to fill the chip alone.  dbs[i] may be NULL. */
size_t fcn_conv2d_wgrad_group_workspace_floats(const fcn_conv_desc* h_ds, int n);
int  fcn_conv2d_wgrad_group_f32(const fcn_conv_desc* h_ds, float* const* h_dws, float* const* h_dbs, int n, float* d_workspace,
                                fcn_stream_t s);
/* The same four calls with the launch configuration named by the caller (the training engine times every configuration once per
 * launch at plan time and keeps the fastest, as the forward engine does): cfg_request -1 = the built-in heuristic (what the calls
 * above use), 0 .. fcn_conv2d_wgrad_num_configs() - 1 = that configuration.  The last one, fcn_conv2d_wgrad_split_config(), is
 * the role-split kernel (eight waves with fixed staging / multiplying roles, a region shape per problem); the others are tile
 * shapes of the 64-wide family.  The workspace of a launch depends on its configuration: size it with the SAME cfg_request (or
 * with the maximum over the ones that may be used).  Every configuration is bit-reproducible; two configurations differ in the
 * last bits (their pixel split counts differ). */
int  fcn_conv2d_wgrad_num_configs(void);
int  fcn_conv2d_wgrad_split_config(void);
size_t fcn_conv2d_wgrad_workspace_floats_cfg(const fcn_conv_desc* h_d, int cfg_request, int* h_splits);
int  fcn_conv2d_wgrad_cfg_f32(const fcn_conv_desc* h_d, float* dw, float* db, float* d_workspace, int cfg_request, fcn_stream_t s);
size_t fcn_conv2d_wgrad_group_workspace_floats_cfg(const fcn_conv_desc* h_ds, int n, int cfg_request);
int  fcn_conv2d_wgrad_group_cfg_f32(const fcn_conv_desc* h_ds, float* const* h_dws, float* const* h_dbs, int n, float* d_workspace,
                                    int cfg_request, fcn_stream_t s);
/* Filter bank of the data-gradient pass: wt[c][kh-1-r][kw-1-q][k] = w[k][r][q][c] (w: [Cout][kh][kw][Cin4],
 * wt: [Cin][kh][kw][Cout4], zero padded).  dX = fcn_conv2d_fwd_f32(dY, wt) with pad' = k-1-pad for stride-1 layers. */
int  fcn_conv_weights_flip_f32(const float* w, float* wt, int Cout, int kh, int kw, int Cin, int Cin4, int Cout4, fcn_stream_t s);
/* All filter banks of a net in ONE launch: segment i flips w_base + w_offset (floats) into wt_base + wt_offset. */
typedef struct fcn_flip_seg { uint64_t w_offset, wt_offset; int32_t Cout, kh, kw, Cin, Cin4, Cout4; } fcn_flip_seg;
int  fcn_conv_weights_flip_batch_f32(const float* w_base, float* wt_base, const fcn_flip_seg* d_segs, int nseg, fcn_stream_t s);
int  fcn_relu_bwd_f32(const float* dy, const float* y, float* dx, int pixels, int C, int cstride, fcn_stream_t s);
int  fcn_sigmoid_bwd_f32(const float* y, const float* dy, float* dx, size_t count, int accumulate, fcn_stream_t s);
int  fcn_maxpool_bwd_f32(const float* dy, const int32_t* idx, float* dx, int N, int H, int W, int C, int dx_cstride, int dx_coffset,
                         int k, int stride, int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate, fcn_stream_t s);
/* The same with the ReLU backward of the blob dx belongs to folded in (relu_y = that blob's activation, may be NULL): used when
 * the pooling backward is the LAST pass that writes the gradient (FCN_CONV_MASK is the convolution's counterpart). */
int  fcn_maxpool_bwd_mask_f32(const float* dy, const int32_t* idx, float* dx, int N, int H, int W, int C, int dx_cstride, int dx_coffset,
                              int k, int stride, int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate,
                              const float* relu_y, int relu_y_cstride, int relu_y_coffset, fcn_stream_t s);
int  fcn_lrn_bwd_f32(const float* x, const float* y, const float* scale, const float* dy, float* dx, int pixels, int C,
                     int x_cstride, int y_cstride, int local_size, float alpha, float beta, int accumulate, fcn_stream_t s);
/* Dropout (TRAIN): y = x * mask / (1 - ratio); mask of element (n,c,h,w) = hash(NCHW index, seed) >= ratio * 2^32.
 * The same call with the same seed applied to the gradient is the backward pass. */
int  fcn_dropout_f32(const float* x, float* y, int N, int C, int H, int W, int x_cstride, int x_coffset, int y_cstride,
                     int y_coffset, float ratio, unsigned seed, unsigned index_offset, fcn_stream_t s);
/* index_offset is added to the NCHW index: rank r of a data-parallel job passes r*N*C*H*W so that the masks of the
 * shards together equal the mask of the undivided batch */
/* kind 0 = L1Loss (NVIDIA Caffe): loss = sum|a-b|/num, da = sign(a-b) * weight/num;
 * kind 1 = EuclideanLoss: loss = sum(a-b)^2/(2 num), da = (a-b) * weight/num.  da may be NULL; d_loss is one device float. */
int  fcn_loss_f32(int kind, const float* a, const float* b, float* da, float* d_loss, int pixels, int C, int cstride, int num,
                  float weight, fcn_stream_t s);
/* SoftmaxWithLoss (train/fcn_bbox/train_val.prototxt:838-847): x NHWC scores, label one float per pixel (class id);
 * loss = -sum log p[label] / (valid pixels if normalize else N); dx (may be NULL) = (p - onehot) * weight / denom.
 * d_workspace: fcn_softmax_loss_workspace_bytes() bytes, 8-byte aligned.  The reduction order is fixed. */
size_t fcn_softmax_loss_workspace_bytes(void);
int  fcn_softmax_loss_f32(const float* x, const float* label, float* dx, float* d_loss, int N, int pixels, int C, int x_cstride,
                          int label_cstride, int normalize, int has_ignore, int ignore_label, float weight, void* d_workspace,
                          fcn_stream_t s);
/* Gradient w.r.t. the input of the depthwise deconvolution (fcn_deconv_depthwise_fwd_f32): H, W are the INPUT extents */
int  fcn_deconv_depthwise_bwd_f32(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int dx_cstride, int k,
                                  int stride, int pad, int OH, int OW, int dy_cstride, int dy_coffset, int accumulate, fcn_stream_t s);
/* Solver update over one flat parameter buffer cut into segments (one per learnable blob). */
typedef struct fcn_solver_seg { uint64_t offset, count; float lr_mult, decay_mult; } fcn_solver_seg;
/* SGD: g' = g*grad_scale + wd*decay_mult*w ; hist = momentum*hist + rate*lr_mult*g' ; w -= hist */
int  fcn_sgd_update_f32(float* w, const float* g, float* hist, const fcn_solver_seg* d_segs, int nseg, float rate, float momentum,
                        float weight_decay, float grad_scale, fcn_stream_t s);
/* Adam (Caffe AdamSolver): m,v moments, w -= rate*lr_mult*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+delta) */
int  fcn_adam_update_f32(float* w, const float* g, float* m, float* v, const fcn_solver_seg* d_segs, int nseg, float rate,
                         float beta1, float beta2, float delta, float weight_decay, int t, float grad_scale, fcn_stream_t s);

/* ---- data-parallel exchange (new capability; the reference trains with --gpu=0 only, train/train.sh:26):
 *      sum of the flat gradient buffer over all ranks with RCCL on the caller's stream ---- */
int  fcn_comm_unique_id(char* h_id128);                                    /* rank 0: ncclGetUniqueId (128 bytes)   */
int  fcn_comm_init(fcn_comm_t* comm, const char* h_id128, int world, int rank);
int  fcn_comm_allreduce_sum_f32(fcn_comm_t comm, float* buf, size_t count, fcn_stream_t s);
int  fcn_comm_destroy(fcn_comm_t comm);

#ifdef __cplusplus
}
#endif
#endif /* FCNHIP_H_ */
