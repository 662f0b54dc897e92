"""Mirror of the inference-decode part of `LightningFunc/utils/YoloV4Utils.py:18-176` and of
`model/YOLOV4.py:18-49` (YoloLayer), on the HIP decode kernel."""
import ctypes

import torch
import torch.nn as nn

from ... import _lib


def _decode(heads, num_classes, per_scale_anchors, scale_x_y=1):
    _lib.require_cuda(*heads)
    dev = heads[0].device
    d = _lib.YoloDecodeDesc()
    d.nscales, d.B, d.C = len(heads), heads[0].shape[0], num_classes
    d.scale_x_y = scale_x_y
    d.scale_x_y_offset = 0.5 * (scale_x_y - 1)
    keep = []
    N = 0
    for i, (h, anc) in enumerate(zip(heads, per_scale_anchors)):
        h = h.float().contiguous()
        keep.append(h)
        A = len(anc) // 2
        if h.shape[1] != A * (5 + num_classes):
            raise ValueError(f"head {i}: expected {A * (5 + num_classes)} channels, got {h.shape[1]}")
        s = d.scale[i]
        s.head, s.H, s.W, s.A = h.data_ptr(), h.shape[2], h.shape[3], A
        for a in range(A):
            s.anchor_w[a], s.anchor_h[a] = anc[2 * a], anc[2 * a + 1]
        N += A * h.shape[2] * h.shape[3]
    boxes = torch.empty(d.B, N, 1, 4, dtype=torch.float32, device=dev)
    confs = torch.empty(d.B, N, num_classes, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().odpl_yolo_decode(ctypes.byref(d), _lib.ptr(boxes), _lib.ptr(confs), _lib.current_stream(dev)), "odpl_yolo_decode")
    return boxes, confs


def yolo_forward_dynamic(output, conf_thresh, num_classes, anchors, num_anchors, scale_x_y, only_objectness=1, validation=False):
    """YoloV4Utils.py:36-176 (conf_thresh / only_objectness / validation are unused there too)."""
    return _decode([output], num_classes, [list(anchors)[:2 * num_anchors]], scale_x_y)


def get_region_boxes(boxes_and_confs):
    """YoloV4Utils.py:18-34."""
    return [torch.cat([b for b, _ in boxes_and_confs], 1), torch.cat([c for _, c in boxes_and_confs], 1)]


def decode_heads(heads, num_classes, anchors18, masks, strides, scale_x_y=1):
    """All YoloLayers + get_region_boxes in ONE launch (what Yolov4Head does with inference=True, YOLOV4.py:126-131)."""
    per = []
    for m, s in zip(masks, strides):
        masked = []
        for k in m:
            masked += list(anchors18[k * 2:(k + 1) * 2])
        per.append([a / s for a in masked])          # YOLOV4.py:45-47: python division, then f32
    return list(_decode(list(heads), num_classes, per, scale_x_y))


class YoloLayer(nn.Module):
    """model/YOLOV4.py:18-49."""

    def __init__(self, anchor_mask=(), num_classes=0, anchors=(), num_anchors=1, stride=32, model_out=False):
        super().__init__()
        self.anchor_mask = list(anchor_mask)
        self.num_classes = num_classes
        self.anchors = list(anchors)
        self.num_anchors = num_anchors
        self.anchor_step = len(self.anchors) // num_anchors
        self.thresh = 0.6
        self.stride = stride
        self.scale_x_y = 1
        self.model_out = model_out

    def forward(self, output, target=None):
        if self.training:
            return output
        masked = []
        for m in self.anchor_mask:
            masked += self.anchors[m * self.anchor_step:(m + 1) * self.anchor_step]
        masked = [a / self.stride for a in masked]
        return yolo_forward_dynamic(output, self.thresh, self.num_classes, masked, len(self.anchor_mask), scale_x_y=self.scale_x_y)
