"""ctypes binding of libodpl_hip.so (C ABI: include/odpl_hip.h).  Fails loudly when absent."""
import ctypes
import os
from ctypes import POINTER, c_float, c_int32, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libodpl_hip.so")

ABI_VERSION = 7            # include/odpl_hip.h: ODPL_ABI_VERSION
LAYOUT_PLANAR, LAYOUT_INTERLEAVED = 0, 1
MAX_SCALES = 4
MAX_ANCHORS = 8
COORD_MODES = {"mse_loss": 0, "smooth_l1_loss": 1}
CLS_MODES = {"bce_loss": 0, "ce_loss": 1}
CLS_SOFTMAX_FOCAL = 2


class OdplError(RuntimeError):
    pass


class YoloScale(ctypes.Structure):
    _fields_ = [("head", c_void_p), ("grad", c_void_p), ("G", c_int32), ("A", c_int32),
                ("anchor_w", c_float * MAX_ANCHORS), ("anchor_h", c_float * MAX_ANCHORS), ("stride", ctypes.c_int64 * 4)]


class SparseGrad(ctypes.Structure):
    """odpl_sparse_grad_t: a structured head gradient (host struct, device pointers)."""
    _fields_ = [("dense", c_void_p), ("rows", c_void_p), ("row_cell", c_void_p), ("n_rows_dev", c_void_p),
                ("n_rows", c_int32), ("B", c_int32), ("G2", c_int32), ("A", c_int32), ("no", c_int32), ("dense_ch", c_int32), ("layout", c_int32)]


class HeadConv(ctypes.Structure):
    """odpl_head_conv_t"""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("grad_x", c_void_p), ("grad_w", c_void_p), ("grad_b", c_void_p),
                ("upstream", c_void_p), ("Cin", c_int32), ("bf16", c_int32)]


class YoloLossDesc(ctypes.Structure):
    _fields_ = [("B", c_int32), ("C", c_int32), ("nT", c_int32), ("nscales", c_int32),
                ("coord_mode", c_int32), ("cls_mode", c_int32), ("average", c_int32), ("flags", c_int32),
                ("ignore_thres", c_float), ("obj_scale", c_float), ("noobj_scale", c_float),
                ("target", c_void_p), ("scale", YoloScale * MAX_SCALES)]


class RetinaLossDesc(ctypes.Structure):
    _fields_ = [("loc", c_void_p), ("cls", c_void_p), ("grad_loc", c_void_p), ("grad_cls", c_void_p),
                ("anchors", c_void_p), ("target", c_void_p),
                ("B", c_int32), ("An", c_int32), ("C", c_int32), ("nT", c_int32),
                ("coord_mode", c_int32), ("flags", c_int32), ("img_size", c_float), ("img_size_y", c_float),
                ("pos_anchor", c_void_p), ("num_pos", c_void_p)]


RETINA_LEVELS = 5


class RetinaHead(ctypes.Structure):
    """odpl_retina_head_t"""
    _fields_ = [("x", c_void_p * RETINA_LEVELS), ("grad_x", c_void_p * RETINA_LEVELS), ("w", c_void_p), ("grad_w", c_void_p), ("grad_b", c_void_p),
                ("grad_rows", c_void_p), ("w_stride", ctypes.c_int64 * 4), ("no", c_int32)]


class RetinaHeadsDesc(ctypes.Structure):
    """odpl_retina_heads_desc_t"""
    _fields_ = [("B", c_int32), ("A", c_int32), ("Cin", c_int32), ("nlevels", c_int32), ("H", c_int32 * RETINA_LEVELS), ("W", c_int32 * RETINA_LEVELS),
                ("pos_anchor", c_void_p), ("num_pos", c_void_p), ("upstream", c_void_p), ("rows_per_image", c_int32), ("flags", c_int32)]


HEADS_SHARED_INPUT = 1       # ODPL_HEADS_SHARED_INPUT
SSD_ROWS_ONLY = 1            # ODPL_SSD_ROWS_ONLY


class SparseConv3x3(ctypes.Structure):
    """odpl_sparse_conv3x3_t"""
    _fields_ = [("x", c_void_p), ("gy", c_void_p), ("active", c_void_p), ("w", c_void_p), ("grad_x", c_void_p), ("grad_w", c_void_p),
                ("w_stride", ctypes.c_int64 * 4), ("B", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32),
                ("ksize", c_int32), ("stats", c_void_p)]


class PixelSupport(ctypes.Structure):
    """odpl_pixel_support_t"""
    _fields_ = [("pos_anchor", c_void_p), ("num_pos", c_void_p), ("B", c_int32), ("An", c_int32), ("A", c_int32), ("n_levels", c_int32),
                ("H", c_int32 * 8), ("W", c_int32 * 8), ("depth", c_int32), ("level_mask", c_int32), ("lists", c_void_p), ("counts", c_void_p)]


class SparseConv3x3Fwd(ctypes.Structure):
    """odpl_sparse_conv3x3_fwd_t"""
    _fields_ = [("x", c_void_p), ("w", c_void_p), ("bias", c_void_p), ("w_stride", ctypes.c_int64 * 4), ("y", c_void_p),
                ("y_batch_stride", ctypes.c_int64), ("y_pixel_stride", ctypes.c_int64), ("list", c_void_p), ("count", c_void_p),
                ("B", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("Cout", c_int32), ("act", c_int32), ("ksize", c_int32)]


class YoloNmsDesc(ctypes.Structure):
    _fields_ = [("head", c_void_p * MAX_SCALES), ("G", c_int32 * MAX_SCALES),
                ("nscales", c_int32), ("B", c_int32), ("A", c_int32), ("no", c_int32),
                ("conf_thres", c_float), ("nms_thres", c_float), ("flags", c_int32)]


class AdamDesc(ctypes.Structure):
    _fields_ = [("table", c_void_p), ("numel", c_void_p), ("blk0", c_void_p), ("blk_tensor", c_void_p), ("host_blk0", c_void_p),
                ("n_tensors", c_int32), ("n_blocks", c_int32), ("maximize", c_int32),
                ("lr", ctypes.c_double), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double)]


ADAM_BLOCK, ADAM_MAX_TENSORS = 4096, 384


class GreedyNmsDesc(ctypes.Structure):
    _fields_ = [("loc", c_void_p), ("cls", c_void_p), ("anchors", c_void_p),
                ("B", c_int32), ("An", c_int32), ("C", c_int32), ("topk", c_int32),
                ("nms_thresh", c_float), ("class_thresh", c_float), ("flags", c_int32)]


class YoloDecodeScale(ctypes.Structure):
    _fields_ = [("head", c_void_p), ("H", c_int32), ("W", c_int32), ("A", c_int32),
                ("anchor_w", c_float * MAX_ANCHORS), ("anchor_h", c_float * MAX_ANCHORS)]


class YoloDecodeDesc(ctypes.Structure):
    _fields_ = [("nscales", c_int32), ("B", c_int32), ("C", c_int32), ("scale_x_y", c_float),
                ("scale_x_y_offset", c_float), ("scale", YoloDecodeScale * MAX_SCALES)]


class Image(ctypes.Structure):
    _fields_ = [("data", c_void_p), ("h", c_int32), ("w", c_int32), ("flip", c_int32), ("reserved", c_int32), ("lut", c_void_p)]


class YoloV5Layer(ctypes.Structure):
    _fields_ = [("p", c_void_p), ("grad", c_void_p), ("G", c_int32),
                ("anchor_w", c_float * MAX_ANCHORS), ("anchor_h", c_float * MAX_ANCHORS), ("layout", c_int32)]


class YoloV5Rows(ctypes.Structure):
    """odpl_yolov5_rows_t"""
    _fields_ = [("row_target", c_void_p), ("row_anchor_group", c_void_p), ("row_cell", c_void_p), ("row_tbox", c_void_p),
                ("n_rows", c_void_p), ("cap", c_int32)]


class YoloV5LossDesc(ctypes.Structure):
    _fields_ = [("nl", c_int32), ("B", c_int32), ("na", c_int32), ("nc", c_int32), ("nT", c_int32),
                ("iou_mode", c_int32), ("flags", c_int32), ("bf16", c_int32), ("target", c_void_p), ("layer", YoloV5Layer * MAX_SCALES)]


class SsdLossDesc(ctypes.Structure):
    _fields_ = [("off", c_void_p), ("cls", c_void_p), ("grad_off", c_void_p), ("grad_cls", c_void_p),
                ("priors", c_void_p), ("target", c_void_p),
                ("B", c_int32), ("P", c_int32), ("C", c_int32), ("nT", c_int32), ("cls_mode", c_int32), ("coord_mode", c_int32),
                ("flags", c_int32), ("active", c_void_p), ("active_count", c_void_p)]


_lib = None

# name -> (restype, argtypes); every symbol include/odpl_hip.h declares must be listed here
# (tests/test_abi.py parses the header and checks both directions).
SIGNATURES = {
    "odpl_abi_version": (c_int32, []),
    "odpl_yolo_region_loss_workspace_bytes": (c_size_t, [POINTER(YoloLossDesc)]),
    "odpl_yolo_region_loss": (c_int32, [POINTER(YoloLossDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "odpl_yolo_region_loss_compact_bytes": (c_size_t, [POINTER(YoloLossDesc)]),
    "odpl_yolo_region_loss_forward": (c_int32, [POINTER(YoloLossDesc), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p]),
    "odpl_yolo_region_loss_backward": (c_int32, [POINTER(YoloLossDesc), c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_yolo_compact_view": (c_int32, [POINTER(YoloLossDesc), c_void_p, c_int32, POINTER(SparseGrad)]),
    "odpl_head_conv_backward_workspace_bytes": (c_size_t, [POINTER(HeadConv), POINTER(SparseGrad)]),
    "odpl_head_conv_backward": (c_int32, [POINTER(HeadConv), POINTER(SparseGrad), c_void_p, c_size_t, c_void_p]),
    "odpl_head_conv_backward_multi_workspace_bytes": (c_size_t, [c_int32, c_void_p, c_void_p]),
    "odpl_head_conv_backward_multi": (c_int32, [c_int32, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_head_conv_tune": (c_int32, [c_int32]),
    "odpl_scale_inplace_if_not_one": (c_int32, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_retina_loss_workspace_bytes": (c_size_t, [POINTER(RetinaLossDesc)]),
    "odpl_retina_loss": (c_int32, [POINTER(RetinaLossDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_retina_heads_backward_workspace_bytes": (c_size_t, [POINTER(RetinaHeadsDesc), c_int32, c_void_p]),
    "odpl_retina_heads_backward": (c_int32, [POINTER(RetinaHeadsDesc), c_int32, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_retina_heads_tune": (c_int32, [c_int32]),
    "odpl_retina_anchors": (c_int32, [POINTER(c_int32), POINTER(c_float), POINTER(c_float), c_void_p, ctypes.c_int64, c_void_p]),
    "odpl_pair_iou": (c_int32, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, ctypes.c_int64, c_int32, c_int32, c_int32, c_void_p]),
    "odpl_bbox_wh_iou": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, ctypes.c_int64, c_void_p]),
    "odpl_box_iou_matrix": (c_int32, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_void_p]),
    "odpl_xywh2xyxy": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_void_p]),
    "odpl_ssd_priors": (c_int32, [POINTER(c_int32), POINTER(c_int32), POINTER(c_float), c_void_p, ctypes.c_int64, c_void_p]),
    "odpl_yolo_nms_workspace_bytes": (c_size_t, [POINTER(YoloNmsDesc)]),
    "odpl_adam_step": (c_int32, [POINTER(AdamDesc), c_void_p, c_void_p]),
    "odpl_yolo_nms": (c_int32, [POINTER(YoloNmsDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_greedy_nms_workspace_bytes": (c_size_t, [POINTER(GreedyNmsDesc)]),
    "odpl_greedy_nms": (c_int32, [POINTER(GreedyNmsDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_yolo_decode": (c_int32, [POINTER(YoloDecodeDesc), c_void_p, c_void_p, c_void_p]),
    "odpl_yolov5_loss_workspace_bytes": (c_size_t, [POINTER(YoloV5LossDesc)]),
    "odpl_yolov5_loss": (c_int32, [POINTER(YoloV5LossDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_yolov5_loss_compact_bytes": (c_size_t, [POINTER(YoloV5LossDesc)]),
    "odpl_yolov5_loss_forward": (c_int32, [POINTER(YoloV5LossDesc), c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_yolov5_loss_backward": (c_int32, [POINTER(YoloV5LossDesc), c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_yolov5_compact_view": (c_int32, [POINTER(YoloV5LossDesc), c_void_p, c_int32, POINTER(SparseGrad), POINTER(YoloV5Rows)]),
    "odpl_ssd_loss_workspace_bytes": (c_size_t, [POINTER(SsdLossDesc)]),
    "odpl_ssd_loss": (c_int32, [POINTER(SsdLossDesc), c_void_p, c_size_t, c_void_p, c_void_p, c_void_p]),
    "odpl_ssd_level_lists": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, POINTER(c_int32), c_void_p, c_void_p, c_void_p]),
    "odpl_debug_set_stamp_buffer": (c_int32, [c_void_p]),
    "odpl_batch_statistics": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    "odpl_ap_per_class_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "odpl_ap_per_class": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_size_t] + [c_void_p] * 6 + [c_void_p]),
    "odpl_yolo_train_decode": (c_int32, [c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(c_float), c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "odpl_yolo_stats_reduce": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_void_p]),
    "odpl_collate_images": (c_int32, [c_void_p, c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_void_p]),
    "odpl_collate_targets": (c_int32, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "odpl_collate_images_warp": (c_int32, [c_void_p, POINTER(ctypes.c_double), c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_void_p]),
    "odpl_collate_targets_warp": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "odpl_bias_grad_nchw": (c_int32, [c_void_p, c_int32, c_int32, c_int32, ctypes.c_int64, c_void_p, c_void_p]),
    "odpl_focus": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(ctypes.c_int64), POINTER(ctypes.c_int64), c_int32, c_void_p]),
    "odpl_reorg": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, POINTER(ctypes.c_int64), c_int32, c_void_p]),
    "odpl_bias_act_forward": (c_int32, [c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "odpl_bias_act_backward": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "odpl_bias_act_backward_flags": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_size_t, c_void_p]),
    "odpl_sparse_conv3x3_backward_workspace_bytes": (c_size_t, [c_void_p]),
    "odpl_sparse_conv3x3_backward": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_pixel_support_workspace_bytes": (c_size_t, [c_void_p]),
    "odpl_pixel_support": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_sparse_conv3x3_forward": (c_int32, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_pixel_union_lists_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "odpl_pixel_union_lists": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32,
                                         c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_sparse_upsample_add": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                           c_void_p]),
    "odpl_upsample2x_nearest": (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, ctypes.c_int64, ctypes.c_int64, c_int32, c_void_p]),
    "odpl_spp_forward": (c_int32, [c_void_p, POINTER(ctypes.c_int64), c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_void_p]),
    "odpl_spp_backward": (c_int32, [c_void_p, POINTER(ctypes.c_int64), c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_void_p, c_void_p]),
    "odpl_spp_forward_ex": (c_int32, [c_void_p, POINTER(ctypes.c_int64), c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_int32, c_int32, c_void_p]),
    "odpl_spp_backward_ex": (c_int32, [c_void_p, POINTER(ctypes.c_int64), c_int32, c_int32, c_int32, c_int32, c_void_p, POINTER(ctypes.c_int64), c_void_p, c_int32, c_int32, c_void_p]),
    "odpl_bn_act_workspace_bytes": (c_size_t, [ctypes.c_int64, c_int32]),
    "odpl_bn_act_forward": (c_int32, [c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float,
                                      c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_bn_act_forward_residual": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float,
                                               c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_bn_add_act_forward": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float,
                                          c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_bn_add_act_backward": (c_int32, [c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                           c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_bn_act_forward_strided": (c_int32, [c_void_p, c_void_p, c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float,
                                              c_int32, c_void_p, c_void_p, c_void_p, ctypes.c_int64, c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_bn_act_backward_strided": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32,
                                               c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_bn_act_backward": (c_int32, [c_void_p, c_void_p, ctypes.c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32,
                                       c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "odpl_tune_set": (c_int32, [c_int32, c_int32]),
    "odpl_scale_inplace_if_not_one_bf16": (c_int32, [c_void_p, c_size_t, c_void_p, c_void_p]),
    "odpl_profile_enable": (c_int32, [c_int32, c_int32]),
    "odpl_profile_collect": (c_int32, [c_int32, POINTER(c_float), POINTER(c_float), c_int32]),
    "odpl_yolo_build_targets": (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                          c_int32, c_float, c_void_p] + [c_void_p] * 10 + [c_void_p, c_void_p]),
}


def lib():
    """The loaded library; raises OdplError (never falls back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OdplError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C objectdetectionpl_amd/csrc` (hipcc --offload-arch=gfx950). "
                f"objectdetectionpl_amd has no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)      # global: the torch binding resolves its symbols against it
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(l, name)
            except AttributeError as e:
                raise OdplError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
            fn.restype = res
            fn.argtypes = args
        if l.odpl_abi_version() != ABI_VERSION:
            raise OdplError("libodpl_hip.so ABI version mismatch; rebuild")
        _lib = l
    return _lib


def check(code, what):
    if code != 0:
        raise OdplError(f"{what} failed with code {code}" + (" (bad argument)" if code == -1 else " (hipError_t)"))


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else c_void_p(t.data_ptr())


def current_stream(device):
    import torch
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise OdplError("objectdetectionpl_amd runs on the HIP device only (tensor is on "
                            f"{t.device}); there is no CPU fallback -- move inputs to cuda")


YOLO_STATE_CLEAN = 1
YOLO_NMS_ONE_WORKGROUP = 1
RETINA_STATE_CLEAN = 1
RETINA_ROWS_ONLY = 2
RETINA_ASSIGN_ONLY = 4
RETINA_ASSIGNED = 8
YOLOV5_STATE_CLEAN = 1
IOU_MODES = {"iou": 0, "giou": 1, "diou": 2, "ciou": 3}
TUNE_KEYS = {"yolo_fill_f4_per_block": 0, "yolo_fill_nt": 1, "retina_fill_nt": 2, "bn_target_blocks": 3, "bn_stream_max_blocks": 4, "bn_max_rows_per_tile": 5, "bn_slice_from_c": 6, "yolo_fused_max_targets": 7}


def tune_set(name, value):
    check(lib().odpl_tune_set(TUNE_KEYS[name], int(value)), "odpl_tune_set")


PROF_IDS = {"yolo_expand": 0, "yolo_assign": 1, "yolo_loss": 2, "yolo_finalize": 3, "retina_assign": 4, "retina_main": 5,
            "yolo_nms_prep": 6, "yolo_nms_image": 7, "greedy_nms_prep": 8, "greedy_nms_image": 9, "yolo_decode": 10, "yolov5_dense": 11, "headgrad_stream": 12, "headgrad_finish": 13, "yolo_patch": 14, "yolov5_obj": 15}


def profile_enable(name, on=True):
    check(lib().odpl_profile_enable(PROF_IDS[name], 1 if on else 0), "odpl_profile_enable")


def profile_collect(name, max_n=4096):
    """Durations (ms) of the named kernel's launches since the last collect (HIP events on the launch stream)."""
    buf = (c_float * max_n)()
    raw = (c_float * max_n)()
    n = lib().odpl_profile_collect(PROF_IDS[name], buf, raw, max_n)
    if n < 0:
        raise OdplError("odpl_profile_collect failed")
    n = min(n, max_n)
    return [buf[i] for i in range(n)], [raw[i] for i in range(n)]


_torch_ext = None


def torch_ext():
    """The C++ torch binding (csrc/torch_ext/odpl_torch_ext.so) for the per-layer epilogue ops, or None if it has
    not been built (callers then use the slower Python autograd path)."""
    global _torch_ext
    if _torch_ext is None:
        import importlib.util
        path = os.path.join(_HERE, "csrc", "torch_ext", "odpl_torch_ext.so")
        if not os.path.exists(path):
            _torch_ext = False
        else:
            lib()                                   # libodpl_hip.so first (RTLD_GLOBAL, soname libodpl_hip.so)
            import torch  # noqa: F401
            spec = importlib.util.spec_from_file_location("odpl_torch_ext", path)
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
            _torch_ext = mod
    return _torch_ext or None
