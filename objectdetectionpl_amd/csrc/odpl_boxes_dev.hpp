// IoU family as device functions; fp32 operation order follows the reference line by line so that
// index decisions taken from these values (argmax, > threshold) agree bit for bit (-ffp-contract=off).
#pragma once
#include "odpl_common.hpp"

namespace odpl {

struct Box4 { float x1, y1, x2, y2; };

// (cx,cy,w,h) -> corners, accuracy.py:43-48 / 289-295: c - w/2, c + w/2
__device__ __forceinline__ Box4 corners_of(float cx, float cy, float w, float h) {
    Box4 b;
    b.x1 = cx - w / 2.0f; b.x2 = cx + w / 2.0f;
    b.y1 = cy - h / 2.0f; b.y2 = cy + h / 2.0f;
    return b;
}

// bbox_iou, accuracy.py:39-69: pixel convention (+1), union + 1e-16
__device__ __forceinline__ float iou_plus1(const Box4& a, const Box4& b) {
    const float ix1 = fmaxf(a.x1, b.x1), iy1 = fmaxf(a.y1, b.y1);
    const float ix2 = fminf(a.x2, b.x2), iy2 = fminf(a.y2, b.y2);
    const float inter = fmaxf(ix2 - ix1 + 1.0f, 0.0f) * fmaxf(iy2 - iy1 + 1.0f, 0.0f);
    const float a1 = (a.x2 - a.x1 + 1.0f) * (a.y2 - a.y1 + 1.0f);
    const float a2 = (b.x2 - b.x1 + 1.0f) * (b.y2 - b.y1 + 1.0f);
    return inter / (a1 + a2 - inter + 1e-16f);
}

// torch.max/min propagate NaN; fmaxf/fminf drop it.  Boxes with NaN coordinates are outside the
// contract (the reference's own NMS loops forever on them), so the cheaper form is used.

// bbox_iou_v5, accuracy.py:71-114.  mode: 0 IoU, 1 GIoU, 2 DIoU, 3 CIoU
__device__ __forceinline__ float iou_v5(const Box4& a, const Box4& b, int mode) {
    const float inter = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f) *
                        fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
    const float w1 = a.x2 - a.x1, h1 = a.y2 - a.y1, w2 = b.x2 - b.x1, h2 = b.y2 - b.y1;
    const float uni = (w1 * h1 + 1e-16f) + w2 * h2 - inter;
    const float iou = inter / uni;
    if (mode == 0) return iou;
    const float cw = fmaxf(a.x2, b.x2) - fminf(a.x1, b.x1);
    const float ch = fmaxf(a.y2, b.y2) - fminf(a.y1, b.y1);
    if (mode == 1) {
        const float c_area = cw * ch + 1e-16f;
        return iou - (c_area - uni) / c_area;
    }
    const float c2 = cw * cw + ch * ch + 1e-16f;
    const float dx = (b.x1 + b.x2) - (a.x1 + a.x2), dy = (b.y1 + b.y2) - (a.y1 + a.y2);
    const float rho2 = dx * dx / 4.0f + dy * dy / 4.0f;
    if (mode == 2) return iou - rho2 / c2;
    const float da = atanf(w2 / h2) - atanf(w1 / h1);
    const float v = 0.40528473456935109f * (da * da);       // 4 / pi^2 (double) rounded to f32 by the multiply
    const float alpha = v / (1.0f - iou + v);
    return iou - (rho2 / c2 + v * alpha);
}

// iou, accuracy.py:6-37: corner boxes, no +1, no epsilon (SSD matching)
__device__ __forceinline__ float iou_corner(const Box4& a, const Box4& b) {
    const float dx = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1), 0.0f);
    const float dy = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1), 0.0f);
    const float inter = dx * dy;
    const float a1 = fmaxf(a.x2 - a.x1, 0.0f) * fmaxf(a.y2 - a.y1, 0.0f);
    const float a2 = fmaxf(b.x2 - b.x1, 0.0f) * fmaxf(b.y2 - b.y1, 0.0f);
    return inter / (a1 + a2 - inter);
}

// RetinaNetLoss.box_iou, losses.py:375-403: +1, no epsilon
__device__ __forceinline__ float iou_plus1_noeps(const Box4& a, const Box4& b) {
    const float w = fmaxf(fminf(a.x2, b.x2) - fmaxf(a.x1, b.x1) + 1.0f, 0.0f);
    const float h = fmaxf(fminf(a.y2, b.y2) - fmaxf(a.y1, b.y1) + 1.0f, 0.0f);
    const float inter = w * h;
    const float a1 = (a.x2 - a.x1 + 1.0f) * (a.y2 - a.y1 + 1.0f);
    const float a2 = (b.x2 - b.x1 + 1.0f) * (b.y2 - b.y1 + 1.0f);
    return inter / (a1 + a2 - inter);
}

}  // namespace odpl
