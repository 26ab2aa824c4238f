"""ctypes binding of libfcnhip.so (include/fcnhip.h).

The shipped path has no CPU fallback: if the HIP library is missing or fails to
load, every entry point raises :class:`FcnLibraryError`.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# $FCN_LIB_PATH selects another BUILD of the same HIP library (the stamped diagnostic build of tools/conv_timeline.py);
# there is still no CPU path behind it.
LIB_PATH = os.environ.get("FCN_LIB_PATH") or os.path.join(_HERE, "libfcnhip.so")


class FcnLibraryError(RuntimeError):
    pass


class FcnError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("libfcnhip error %d: %s" % (code, msg))
        self.code = code


class ConvDesc(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("y", C.c_void_p), ("y2", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32), ("x_cstride", C.c_int32),
        ("Cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32), ("pad", C.c_int32), ("stride", C.c_int32),
        ("OH", C.c_int32), ("OW", C.c_int32),
        ("y_cstride", C.c_int32), ("y_coffset", C.c_int32), ("y2_cstride", C.c_int32), ("y2_coffset", C.c_int32),
        ("flags", C.c_int32), ("in_shift", C.c_float),
    ]


class ConvTail(C.Structure):
    _fields_ = [("n", C.c_int32), ("finalize", C.c_int32), ("heads", ConvDesc * 4), ("scratch", C.c_void_p), ("arrive", C.c_void_p)]


class ConvGroup(C.Structure):
    _fields_ = [("d_probs", C.c_void_p), ("n", C.c_int32), ("cfg", C.c_int32), ("total_tiles", C.c_int32)]


class SolverSeg(C.Structure):
    _fields_ = [("offset", C.c_uint64), ("count", C.c_uint64), ("lr_mult", C.c_float), ("decay_mult", C.c_float)]


class LayoutDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p)] + [(k, C.c_int32) for k in ("N", "C", "H", "W", "src_cstride", "src_coffset")]


class PoolDesc(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("idx", C.c_void_p)] + [(k, C.c_int32) for k in (
        "N", "H", "W", "C", "x_cstride", "k", "stride", "pad", "OH", "OW", "y_cstride", "y_coffset", "f16")]


class SceneObj(C.Structure):
    _fields_ = [("img", C.c_void_p), ("mask", C.c_void_p)] + [(k, C.c_int32) for k in (
        "src_h", "src_w", "flip", "roi_x", "roi_y", "roi_w", "roi_h", "out_w", "out_h", "cx", "cy", "label1")]


class ColorParams(C.Structure):
    _fields_ = [("sharpen_centre", C.c_float), ("sharpen_off", C.c_float), ("add", C.c_int32 * 3), ("mul", C.c_float * 3),
                ("gray_alpha", C.c_float), ("gray_keep", C.c_float)]


GAUSS_MAX_RADIUS = 15


class FlipSeg(C.Structure):
    _fields_ = [("w_offset", C.c_uint64), ("wt_offset", C.c_uint64), ("Cout", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("Cin", C.c_int32), ("Cin4", C.c_int32), ("Cout4", C.c_int32)]


class DetectParams(C.Structure):
    _fields_ = [
        ("num_classes", C.c_int32), ("gy", C.c_int32), ("gx", C.c_int32), ("cell_w", C.c_int32), ("cell_h", C.c_int32),
        ("cvg_cstride", C.c_int32), ("cvg_coffset", C.c_int32), ("box_cstride", C.c_int32), ("box_coffset", C.c_int32),
        ("prob_thresh", C.c_float), ("group_thresh", C.c_int32), ("eps", C.c_double), ("min_height", C.c_int32),
        ("round_mode", C.c_int32), ("max_out", C.c_int32),
    ]


CONV_RELU, CONV_SIGMOID2, CONV_ACCUM, CONV_OUT_F32, CONV_F16, CONV_OUT_F16, CONV_MASK = 1, 2, 4, 8, 16, 32, 64
CONV_IMAGE_ONES = 128      # half image whose channels 3 and 4 are the constant 1 (the folded Power shift): see include/fcnhip.h
ELT_PROD, ELT_SUM, ELT_MAX = 0, 1, 2
RECT_ROUND_NEAREST_EVEN, RECT_ROUND_TRUNCATE = 0, 1

_vp, _i, _f, _d, _sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t

# name -> (restype, argtypes).  Every symbol include/fcnhip.h declares is listed here.
PROTOTYPES = {
    "fcn_abi_version": (_i, []),
    "fcn_last_error_string": (C.c_char_p, []),
    "fcn_device_count": (_i, [C.POINTER(_i)]),
    "fcn_init": (_i, [_i]),
    "fcn_device_name": (_i, [C.c_char_p, _i]),
    "fcn_device_sync": (_i, []),
    "fcn_malloc": (_i, [C.POINTER(_vp), _sz]),
    "fcn_free": (_i, [_vp]),
    "fcn_host_malloc": (_i, [C.POINTER(_vp), _sz]),
    "fcn_host_free": (_i, [_vp]),
    "fcn_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "fcn_memcpy_h2d_async": (_i, [_vp, _vp, _sz, _vp]),
    "fcn_memcpy_d2h_async": (_i, [_vp, _vp, _sz, _vp]),
    "fcn_memcpy_d2d_async": (_i, [_vp, _vp, _sz, _vp]),
    "fcn_stream_create": (_i, [C.POINTER(_vp)]),
    "fcn_stream_destroy": (_i, [_vp]),
    "fcn_stream_sync": (_i, [_vp]),
    "fcn_event_create": (_i, [C.POINTER(_vp)]),
    "fcn_event_destroy": (_i, [_vp]),
    "fcn_event_record": (_i, [_vp, _vp]),
    "fcn_event_sync": (_i, [_vp]),
    "fcn_event_elapsed_ms": (_i, [_vp, _vp, C.POINTER(_f)]),
    "fcn_graph_begin": (_i, [_vp]),
    "fcn_graph_end": (_i, [_vp, C.POINTER(_vp)]),
    "fcn_graph_launch": (_i, [_vp, _vp]),
    "fcn_graph_destroy": (_i, [_vp]),
    "fcn_nchw_to_nhwc_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "fcn_nhwc_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcn_nhwc_to_nchw_multi_f32": (_i, [C.POINTER(LayoutDesc), _i, _vp]),
    "fcn_conv2d_fwd_f32": (_i, [C.POINTER(ConvDesc), _vp]),
    "fcn_conv2d_group_workspace_bytes": (_sz, [_i]),
    "fcn_conv2d_num_configs": (_i, []),
    "fcn_conv2d_first_layer_config": (_i, []),
    "fcn_conv2d_config_lds_bytes": (_i, [_i]),
    "fcn_conv2d_config_waves_k": (_i, [_i]),
    "fcn_conv2d_group_prepare": (_i, [C.POINTER(ConvDesc), _i, _vp, _i, C.POINTER(ConvGroup)]),
    "fcn_conv2d_group_prepare_fused": (_i, [C.POINTER(ConvDesc), _i, C.POINTER(PoolDesc), _i, _vp, _i, C.POINTER(ConvGroup)]),
    "fcn_conv2d_fwd_group_f32": (_i, [C.POINTER(ConvGroup), _vp]),
    "fcn_conv2d_group_release": (_i, [_vp]),
    "fcn_conv2d_tail_scratch_bytes": (C.c_size_t, [C.POINTER(ConvTail)]),
    "fcn_conv2d_tail_arrive_bytes": (C.c_size_t, [C.POINTER(ConvTail)]),
    "fcn_conv2d_group_attach_tail": (_i, [_vp, C.POINTER(ConvTail)]),
    "fcn_maxpool_fwd_f32": (_i, [_vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "fcn_avepool_fwd_f32": (_i, [_vp, _vp] + [_i] * 12 + [_vp]),
    "fcn_lrn_fwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "fcn_maxpool_lrn5_fwd_f32": (_i, [_vp, _vp] + [_i] * 12 + [_f, _f, _f, _vp]),
    "fcn_maxpool_lrn5_fwd_f16": (_i, [_vp, _vp] + [_i] * 12 + [_f, _f, _f, _vp]),
    "fcn_maxpool_lrn5_conv1x1_fwd_f32": (_i, [_vp] + [_i] * 10 + [_f, _f, _f, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "fcn_maxpool_lrn5_conv1x1_fwd_f16": (_i, [_vp] + [_i] * 10 + [_f, _f, _f, _vp, _vp, _i, _i, _vp, _i, _i, _vp]),
    "fcn_relu_fwd_f32": (_i, [_vp, _vp, _sz, _f, _vp]),
    "fcn_sigmoid_fwd_f32": (_i, [_vp, _vp, _sz, _vp]),
    "fcn_power_fwd_f32": (_i, [_vp, _vp, _sz, _f, _f, _f, _vp]),
    "fcn_eltwise_fwd_f32": (_i, [_vp, _vp, _vp, _sz, _i, _f, _f, _vp]),
    "fcn_copy_channels_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcn_deconv_depthwise_fwd_f32": (_i, [_vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "fcn_preprocess_bgr8": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp]),
    "fcn_detect_workspace_bytes": (_sz, [C.POINTER(DetectParams), _i]),
    "fcn_detect_decode_group": (_i, [_vp, _vp, _i, _sz, _sz, C.POINTER(DetectParams), _vp, _vp, _vp, _vp, _vp]),
    "fcn_score_masks_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "fcn_score_masks": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _f, _vp, _i, _i, _vp, _vp, _vp]),
    "fcn_gen_targets": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _vp, _vp, _vp, _vp, _vp]),
    "fcn_gen_targets_nhwc": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _d, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp]),
    "fcn_conv2d_wgrad_workspace_floats": (_sz, [C.POINTER(ConvDesc), C.POINTER(_i)]),
    "fcn_conv2d_wgrad_f32": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp]),
    "fcn_conv2d_wgrad_group_workspace_floats": (_sz, [C.POINTER(ConvDesc), _i]),
    "fcn_conv2d_wgrad_group_f32": (_i, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp, _vp]),
    "fcn_conv2d_wgrad_num_configs": (_i, []),
    "fcn_conv2d_wgrad_split_config": (_i, []),
    "fcn_conv2d_wgrad_workspace_floats_cfg": (_sz, [C.POINTER(ConvDesc), _i, C.POINTER(_i)]),
    "fcn_conv2d_wgrad_cfg_f32": (_i, [C.POINTER(ConvDesc), _vp, _vp, _vp, _i, _vp]),
    "fcn_conv2d_wgrad_group_workspace_floats_cfg": (_sz, [C.POINTER(ConvDesc), _i, _i]),
    "fcn_conv2d_wgrad_group_cfg_f32": (_i, [C.POINTER(ConvDesc), _vp, _vp, _i, _vp, _i, _vp]),
    "fcn_conv_weights_flip_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcn_nchw_f32_to_nhwc_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _f, _vp]),
    "fcn_nhwc_f16_to_nchw_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcn_maxpool_fwd_f16": (_i, [_vp, _vp] + [_i] * 12 + [_vp]),
    "fcn_lrn_fwd_f16": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _f, _f, _f, _vp]),
    "fcn_preprocess_bgr8_batch": (_i, [_vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "fcn_preprocess_bgr8_rois": (_i, [_vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _i, _f, _vp, _vp]),
    "fcn_preprocess_bgr8_f16": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _f, _vp, _vp]),
    "fcn_compose_scene_bgr8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _vp]),
    "fcn_mask_to_label_f32": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "fcn_compose_scene_view_bgr8": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "fcn_blur_gauss_bgr8": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "fcn_blur_box_bgr8": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "fcn_blur_median_bgr8": (_i, [_vp, _vp, _i, _i, _i, _vp]),
    "fcn_color_augment_bgr8": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "fcn_softmax_fwd_f32": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "fcn_softmax_loss_workspace_bytes": (_sz, []),
    "fcn_softmax_loss_f32": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _f, _vp, _vp]),
    "fcn_deconv_depthwise_bwd_f32": (_i, [_vp, _vp, _vp] + [_i] * 13 + [_vp]),
    "fcn_conv_weights_flip_batch_f32": (_i, [_vp, _vp, _vp, _i, _vp]),
    "fcn_relu_bwd_f32": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "fcn_sigmoid_bwd_f32": (_i, [_vp, _vp, _vp, _sz, _i, _vp]),
    "fcn_maxpool_bwd_f32": (_i, [_vp, _vp, _vp] + [_i] * 14 + [_vp]),
    "fcn_maxpool_bwd_mask_f32": (_i, [_vp, _vp, _vp] + [_i] * 14 + [_vp, _i, _i, _vp]),
    "fcn_lrn_bwd_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _f, _i, _vp]),
    "fcn_dropout_f32": (_i, [_vp, _vp] + [_i] * 8 + [_f, C.c_uint, C.c_uint, _vp]),
    "fcn_stream_wait_event": (_i, [_vp, _vp]),
    "fcn_comm_unique_id": (_i, [C.c_char_p]),
    "fcn_comm_init": (_i, [C.POINTER(_vp), C.c_char_p, _i, _i]),
    "fcn_comm_allreduce_sum_f32": (_i, [_vp, _vp, _sz, _vp]),
    "fcn_comm_destroy": (_i, [_vp]),
    "fcn_loss_f32": (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "fcn_sgd_update_f32": (_i, [_vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _vp]),
    "fcn_adam_update_f32": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _f, _f, _i, _f, _vp]),
}

HW_QUEUES: dict = {}      # what load() found / did about GPU_MAX_HW_QUEUES
_lib: Optional[C.CDLL] = None
_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libfcnhip.so (once) and attach prototypes.  Raises loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise FcnLibraryError(
                "libfcnhip.so not found at %s — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C fcn_object_detector_amd/csrc`. There is no CPU fallback." % LIB_PATH)
        # The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), in creation
        # order and counting idle streams; it reads the variable once, when it initialises.  Streams that share a queue do
        # not overlap: with 4 queues a fourth replica stream of a frame pipeline aliases another one and throughput DROPS
        # (3460 vs 4300 frames/s), the two-stream training step runs 6.60 vs 6.37 ms; with 8 the node pipeline lost a third
        # of its rate as soon as another pipeline's four idle streams existed (2590 vs 3600 frames/s).  16 covers the
        # engines a process of this package keeps alive at once; no measured cost against 8.
        # An embedding process should export GPU_MAX_HW_QUEUES itself; the variable is only filled in
        # here when nobody has set it and $FCN_SET_HW_QUEUES is not 0 - and that is said once, not done silently.
        global HW_QUEUES
        if "GPU_MAX_HW_QUEUES" in os.environ:
            HW_QUEUES = {"value": os.environ["GPU_MAX_HW_QUEUES"], "set_by": "environment"}
        elif os.environ.get("FCN_SET_HW_QUEUES", "1") != "0":
            os.environ["GPU_MAX_HW_QUEUES"] = "16"
            HW_QUEUES = {"value": "16", "set_by": "fcn_object_detector_amd.lib.load"}
            if os.environ.get("FCN_QUIET", "0") in ("", "0"):
                import sys
                sys.stderr.write("fcn_object_detector_amd: GPU_MAX_HW_QUEUES was unset; set to 16 for this process (export it yourself or "
                                 "FCN_SET_HW_QUEUES=0 to keep the runtime's default; FCN_QUIET=1 silences this line)\n")
        else:
            HW_QUEUES = {"value": None, "set_by": "nobody (FCN_SET_HW_QUEUES=0): the HIP runtime's default of 4 applies"}
        # RCCL's intra-node transport shares buffers between rank processes through dmabuf IPC; on hosts whose driver only
        # supports that mode the legacy handle path fails with `hipIpcGetMemHandle: invalid argument`.  Read by the HSA
        # runtime when it initialises, so it is filled in here for ranks that a foreign launcher (torch.distributed.run)
        # started without it - only when the job has more than one rank and nobody has set it.
        if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            lib = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
        except OSError as e:  # pragma: no cover - depends on the box
            raise FcnLibraryError("cannot load %s: %s" % (LIB_PATH, e)) from e
        for name, (res, args) in PROTOTYPES.items():
            try:
                fn = getattr(lib, name)
            except AttributeError as e:
                raise FcnLibraryError("libfcnhip.so lacks symbol %s (stale build?)" % name) from e
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = load().fcn_last_error_string()
        raise FcnError(rc, msg.decode("utf-8", "replace") if msg else "")


def call(name: str, *args) -> None:
    """Call an int-returning entry point and raise :class:`FcnError` on a non-zero code."""
    check(getattr(load(), name)(*args))
