"""Oracle (test infrastructure): YOLOv4 inference decode, restated on torch-CPU from
`LightningFunc/utils/YoloV4Utils.py:18-176` (yolo_forward_dynamic / get_region_boxes) and
`model/YOLOV4.py:41-49` (YoloLayer: masked anchors / stride).  Pinned by tests/golden/decode.npz."""
import torch


def decode_scale(output, num_classes, anchors, num_anchors, scale_x_y=1):
    """YoloV4Utils.py:36-176.  output [B, A*(5+C), H, W]; anchors: flat python list (w0,h0,w1,h1,...) in grid units.
    -> boxes [B, A*H*W, 1, 4] (x1,y1,x2,y2 normalised), confs [B, A*H*W, C]."""
    B, _, H, W = output.shape
    C = num_classes
    o = output.view(B, num_anchors, 5 + C, H, W)
    bxy = torch.sigmoid(o[:, :, 0:2]) * scale_x_y - 0.5 * (scale_x_y - 1)
    bwh = torch.exp(o[:, :, 2:4])
    det = torch.sigmoid(o[:, :, 4])                                       # [B,A,H,W]
    cls = torch.sigmoid(o[:, :, 5:])                                      # [B,A,C,H,W]
    gx = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    gy = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    aw = torch.tensor([anchors[2 * i] for i in range(num_anchors)], dtype=torch.float32).view(1, num_anchors, 1, 1)
    ah = torch.tensor([anchors[2 * i + 1] for i in range(num_anchors)], dtype=torch.float32).view(1, num_anchors, 1, 1)
    bx = (bxy[:, :, 0] + gx) / W
    by = (bxy[:, :, 1] + gy) / H
    bw = (bwh[:, :, 0] * aw) / W
    bh = (bwh[:, :, 1] * ah) / H
    x1 = bx - bw * 0.5
    y1 = by - bh * 0.5
    boxes = torch.stack((x1, y1, x1 + bw, y1 + bh), -1).view(B, num_anchors * H * W, 1, 4)
    confs = cls.permute(0, 1, 3, 4, 2).reshape(B, num_anchors * H * W, C) * det.reshape(B, num_anchors * H * W, 1)
    return boxes, confs


def decode_heads(heads, num_classes, anchors18, masks, strides, scale_x_y=1):
    """YoloLayer.forward (YOLOV4.py:41-49) per scale + get_region_boxes (YoloV4Utils.py:18-34)."""
    bl, cl = [], []
    for h, m, s in zip(heads, masks, strides):
        masked = []
        for k in m:
            masked += anchors18[k * 2:(k + 1) * 2]
        masked = [a / s for a in masked]
        b, c = decode_scale(h, num_classes, masked, len(m), scale_x_y)
        bl.append(b); cl.append(c)
    return torch.cat(bl, 1), torch.cat(cl, 1)
