/*
 * odpl_hip.h -- C ABI of libodpl_hip.so: the MI355X (gfx950) detector-math library.
 *
 * The reference (Leyan529/ObjectDetectionPL) has no FFI: its boundary is the Python API
 * `LightningFunc.losses.configure_loss(...)` -> nn.Module.forward(output, target) -> dict
 * (losses.py:9-35) plus the free functions of LightningFunc/accuracy.py.  Each entry point
 * below replaces the eager-torch op sequence of one of those functions; the file:line it
 * replaces is cited per entry.  The Python host in objectdetectionpl_amd/ binds these with
 * ctypes (see INTEGRATION.md for the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain pointers + sizes only; every pointer is a DEVICE pointer unless named host_*;
 *   - all tensors are dense, contiguous, row-major, fp32 unless stated; 16-byte aligned bases;
 *   - `stream` is a hipStream_t passed as void*; every entry only ENQUEUES work on it
 *     (no allocation, no synchronisation, graph-capturable);
 *   - return 0 on success, ODPL_EINVAL (-1) on a bad argument, otherwise the hipError_t
 *     of the failing enqueue.  Data errors follow the reference: out-of-range labels and
 *     coordinates are not errors, they trip the reference's guards (accuracy.py:340-367).
 */
#ifndef ODPL_HIP_H
#define ODPL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ODPL_ABI_VERSION 7   /* 3: memory-format strides + compact (structured) gradients on the YOLO losses, head-conv backward from them;
                              * 4: odpl_retina_loss_desc_t grew pos_anchor / num_pos (ODPL_RETINA_ROWS_ONLY), odpl_retina_heads_backward
                              * 5: odpl_sparse_conv3x3_t grew stats; ODPL_RETINA_ASSIGN_ONLY / _ASSIGNED, odpl_pixel_support, odpl_sparse_conv3x3_forward,
                              *    ksize on both sparse-convolution descriptors, odpl_pixel_union_lists, odpl_sparse_upsample_add
                              * 6: the two probes left the library (tools/probes/: odpl_conv1x1_*, odpl_bn_fused_*, ODPL_TUNE_BN_FUSED*);
                              *    odpl_adam_step; ODPL_YOLO_NMS_ONE_WORKGROUP; ODPL_TUNE_YOLO_FUSED_MAX_TARGETS
                              * 7: odpl_sparse_grad_t grew n_rows_dev (the produced row count of a capacity-sized row list); odpl_retina_heads_desc_t
                              *    grew rows_per_image / flags (ODPL_HEADS_SHARED_INPUT); odpl_ssd_loss_desc_t grew flags / active / active_count
                              *    (ODPL_SSD_ROWS_ONLY), odpl_ssd_level_lists */
#define ODPL_EINVAL (-1)
#define ODPL_MAX_SCALES 4
#define ODPL_MAX_ANCHORS 8

/* coord_criterion (losses.py:17-18) and cls_criterion (losses.py:12-14) selectors */
#define ODPL_COORD_MSE 0
#define ODPL_COORD_SMOOTH_L1 1
#define ODPL_CLS_BCE 0
#define ODPL_CLS_CE 1

int odpl_abi_version(void);

/* ---- YOLO v2/v3/v4 region loss: losses.py:475-905 + accuracy.py:305-380 ------------------ */

/* memory formats of a [B, A*(5+C), G, G] head tensor (and of its gradient) */
#define ODPL_LAYOUT_PLANAR 0        /* NCHW: element (b,a,c,s) at ((b*A + a)*(5+C) + c)*G*G + s */
#define ODPL_LAYOUT_INTERLEAVED 1   /* channels-last: element (b,a,c,s) at ((b*G*G + s)*A + a)*(5+C) + c */

typedef struct {
    const float* head;              /* [B, A*(5+C), G, G] raw head tensor, in the memory format the producing conv wrote */
    float* grad;                    /* d loss / d head, same shape AND strides; NULL = forward only (validation) */
    int32_t G;                      /* grid size */
    int32_t A;                      /* anchors of this scale (<= ODPL_MAX_ANCHORS) */
    float anchor_w[ODPL_MAX_ANCHORS];   /* scaled anchors, grid units: f32(double(aw)/double(img/G)), */
    float anchor_h[ODPL_MAX_ANCHORS];   /* the host table of losses.py:499-503 */
    int64_t stride[4];              /* element strides (b, c, y, x) of head and grad as a dense tensor: NCHW-contiguous or
                                     * channels-last (torch.channels_last); all zero = NCHW-contiguous.  Anything else is
                                     * ODPL_EINVAL (the reference copies to [B,A,G,G,5+C] first, losses.py:517-522; here no
                                     * copy is made for either format) */
} odpl_yolo_scale_t;

typedef struct {
    int32_t B;                      /* images in the batch */
    int32_t C;                      /* classes */
    int32_t nT;                     /* rows of `target` */
    int32_t nscales;                /* 1 (v2) .. 3 (v3/v4) */
    int32_t coord_mode;             /* ODPL_COORD_* */
    int32_t cls_mode;               /* ODPL_CLS_* */
    int32_t average;                /* 1: divide every output by nscales (losses.py:616-623); 0: v2 */
    int32_t flags;                  /* ODPL_YOLO_* bits */
    float ignore_thres;             /* 0.5, losses.py:483 */
    float obj_scale;                /* 1,   losses.py:487 */
    float noobj_scale;              /* 100, losses.py:488 */
    const float* target;            /* [nT,6] (img, cls, x, y, w, h) normalised; may be NULL if nT==0 */
    odpl_yolo_scale_t scale[ODPL_MAX_SCALES];
} odpl_yolo_loss_desc_t;

/* flags.  STATE_CLEAN: the caller promises that the last use of this workspace was a successful
 * odpl_yolo_region_loss[_forward] call with the same B, A, G and memory format per scale (the call leaves its cell-state
 * maps, which sit first in the workspace, reset); the initial fill is then skipped.  A broken promise is detected where it
 * would index out of bounds (status bit 1 of out_counts[.][4]) and never faults, but the losses of that call are garbage. */
#define ODPL_YOLO_STATE_CLEAN 1

/* Bytes of scratch `odpl_yolo_region_loss` needs for this descriptor (host-side arithmetic only).  Includes the compact
 * gradient area when any scale has a `grad` pointer. */
size_t odpl_yolo_region_loss_workspace_bytes(const odpl_yolo_loss_desc_t* desc);

/*
 * Replaces MultiScaleRegionLoss_v4.forward (losses.py:583-634), MultiScaleRegionLoss_v3.forward
 * (744-795) and RegionLoss_v2.forward (830-905) INCLUDING their autograd backward:
 * target assignment (accuracy.py:305-380), the seven loss terms, and d loss/d head for a unit
 * upstream gradient, in one call (= odpl_yolo_region_loss_forward + _backward below).
 *   out7        [7]  loss, Localization, Size, Conf, Classification, Conf_obj, Conf_noobj
 *   out_scales  [nscales*7] the per-scale 7-tuples before accumulation (may be NULL)
 *   out_counts  [nscales*8] int32: nObj, nNoobj, guard1_tripped, guard2_tripped, status, 0,0,0 (may be NULL)
 *               status bit 0: an index below -size (the reference raises IndexError), bit 1: dirty state map
 */
int odpl_yolo_region_loss(const odpl_yolo_loss_desc_t* desc, void* workspace, size_t workspace_bytes,
                          float* out7, float* out_scales, int32_t* out_counts, void* stream);

/*
 * The same, split the way autograd calls it.  Given the assignment every loss term is local to a cell and all terms but
 * the confidence BCE live on the <= nT obj cells (losses.py:557-564), so d loss/d head is STRUCTURED: one value per cell
 * (its confidence logit), 4+C values on each obj cell, zeros elsewhere.  forward leaves it in that compact form in
 * `compact` (odpl_yolo_region_loss_compact_bytes; NULL: no gradient wanted -- validation_step, step.py:43-50):
 *     per scale: gconf f32 [cells] in the head's cell order, gobj f32 [nT, 5+C] (slot 4 unused), tcell int32 [nT] (-1: the
 *     target owns no cell).
 * backward expands it into scale[i].grad (the head's memory format), multiplied by *upstream (device scalar, NULL = 1):
 * the dense tensor a generic producer's backward expects.  `workspace` only needs odpl_yolo_region_loss_workspace_bytes of
 * the same descriptor with every grad NULL.  A 1x1-convolution producer skips backward altogether: odpl_head_conv_backward
 * consumes the compact form (odpl_yolo_compact_view) and the dense tensor is never written.
 */
size_t odpl_yolo_region_loss_compact_bytes(const odpl_yolo_loss_desc_t* desc);
int odpl_yolo_region_loss_forward(const odpl_yolo_loss_desc_t* desc, void* workspace, size_t workspace_bytes,
                                  void* compact, size_t compact_bytes, float* out7, float* out_scales,
                                  int32_t* out_counts, void* stream);
int odpl_yolo_region_loss_backward(const odpl_yolo_loss_desc_t* desc, const void* compact, size_t compact_bytes,
                                   const float* upstream, void* stream);

/* A structured output gradient of a detection head [B, A*no, G, G] (HOST struct holding DEVICE pointers):
 *   dense     f32 [B*A*G2]: gradient of channel a*no + dense_ch of every cell, in the layout's cell order
 *             (planar: (b*A + a)*G2 + s; interleaved: (b*G2 + s)*A + a)
 *   rows      f32 [n_rows, no]: gradient of the `no` channels of anchor a at one cell (slot dense_ch is ignored: it is in `dense`)
 *   row_cell  int32 [n_rows]: that cell (same order as `dense`), or -1 = no row
 *   n_rows_dev  optional device scalar <= n_rows: rows at and beyond it all carry row_cell -1 (a list sized for its capacity whose
 *             count never visits the host: the YOLOv5 loss); a hint that lets consumers stop early, NULL = none */
typedef struct {
    const float* dense;
    const float* rows;
    const int32_t* row_cell;
    const int32_t* n_rows_dev;
    int32_t n_rows, B, G2, A, no, dense_ch, layout;
} odpl_sparse_grad_t;
/* the compact gradient of scale `scale` as left by odpl_yolo_region_loss_forward (host-side pointer arithmetic only) */
int odpl_yolo_compact_view(const odpl_yolo_loss_desc_t* desc, const void* compact, int32_t scale, odpl_sparse_grad_t* out);

/* ---- backward of a head's 1x1 convolution from a structured gradient -------------------------------------------
 * The linear head convolutions (model/YOLOV4.py:89-127 conv2/conv10/conv18, YOLOV3 / YOLOV2 heads, YOLOV5.py:93-97
 * Detect.m: Conv2d(Cin, A*no, 1) with bias) feed the criterion directly.  Their backward from an odpl_sparse_grad_t:
 *     dX[n,:]  = sum_a dense[n,a] * W[a*no+dc,:] + sum_{rows t at n} sum_{c != dc} rows[t,c] * W[a_t*no+c,:]
 *     dW[ch,:] = sum_n (d loss/d head)[n,ch] * X[n,:]        db[ch] = sum_n (d loss/d head)[n,ch]
 * (what aten::convolution_backward computes from the dense tensor -- the implicit autograd of model/YOLOV4.py:216-218 --
 * without ever writing that tensor).  X: the convolution's input as [N = B*G2, Cin] row-major (a dense channels-last
 * [B, Cin, G, G] tensor), fp32 or bf16; W: [A*no, Cin] fp32.  Every output is fully written (not accumulated). */
typedef struct {
    const void* x;                  /* [N, Cin] */
    const float* w;                 /* [A*no, Cin] */
    void* grad_x;                   /* [N, Cin], dtype of x; NULL = not wanted */
    float* grad_w;                  /* [A*no, Cin]; NULL = not wanted */
    float* grad_b;                  /* [A*no]; NULL = not wanted */
    const float* upstream;          /* device scalar multiplied into every gradient; NULL = 1 */
    int32_t Cin;                    /* % 4 == 0 (% 8 for bf16) */
    int32_t bf16;                   /* x / grad_x are bfloat16 */
} odpl_head_conv_t;
size_t odpl_head_conv_backward_workspace_bytes(const odpl_head_conv_t* head, const odpl_sparse_grad_t* grad);
int odpl_head_conv_backward(const odpl_head_conv_t* head, const odpl_sparse_grad_t* grad, void* workspace,
                            size_t workspace_bytes, void* stream);
/* All heads of a model in ONE pair of launches (same A and dtype on every head, n <= ODPL_MAX_SCALES); `heads` / `grads`
 * are host arrays.  Limits: A <= 5 anchors per head, no <= 4096, N * Cin < 2^33; any number of rows (the finish launch walks
 * row lists longer than 1 792 rows in chunks).  The *_workspace_bytes queries return 0 for a descriptor outside them -- callers probe
 * with it BEFORE choosing this route (utils/headconv.py: structured_backward_supported) and otherwise take the dense route
 * (odpl_yolo_region_loss_backward / odpl_yolov5_loss_backward + a generic convolution backward).  The queries read no device
 * memory: pointers only need to be non-NULL. */
size_t odpl_head_conv_backward_multi_workspace_bytes(int32_t n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads);
int odpl_head_conv_backward_multi(int32_t n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads, void* workspace,
                                  size_t workspace_bytes, void* stream);
/* launch geometry of the streaming kernel: rows of X per workgroup, quoted for heads of >= 64 16-byte vectors per row (4 row lanes per
 * workgroup; narrower heads get proportionally taller tiles: every lane owns rows_per_tile / 4 rows); default 32;
 * 0 = the smallest height whose grid is one resident wave */
int odpl_head_conv_tune(int32_t rows_per_tile);

/*
 * autograd glue: grad[i] *= *scale unless *scale == 1.0f (decided on the device, no host sync).
 * Used by the backward of the fused loss, whose gradient was produced for a unit upstream gradient.
 */
int odpl_scale_inplace_if_not_one(float* grad, size_t n, const float* scale, void* stream);
int odpl_scale_inplace_if_not_one_bf16(void* grad_bf16, size_t n, const float* scale, void* stream);

/*
 * Drop-in for accuracy.build_targets (accuracy.py:305-380): the reference's ten dense outputs.
 *   pred_boxes [B,A,G,G,4], pred_cls [B,A,G,G,C], target [nT,6], anchors [A,2] (device, grid units)
 *   obj_mask/noobj_mask uint8 [B,A,G,G]; iou_scores,class_mask,tx,ty,tw,th,tconf f32 [B,A,G,G];
 *   tcls f32 [B,A,G,G,C].  `state` is int32 scratch [B*A*G*G].  status: int32[8] (device) as out_counts.
 */
int odpl_yolo_build_targets(const float* pred_boxes, const float* pred_cls, const float* target,
                            const float* anchors, int32_t B, int32_t A, int32_t G, int32_t C, int32_t nT,
                            float ignore_thres, int32_t* state,
                            float* iou_scores, float* class_mask, uint8_t* obj_mask, uint8_t* noobj_mask,
                            float* tx, float* ty, float* tw, float* th, float* tcls, float* tconf,
                            int32_t* status, void* stream);

/* ---- RetinaNet loss: losses.py:351-472 (+ focal_loss 317-349) ------------------------------ */

typedef struct {
    const float* loc;               /* [B, An, 4] predicted offsets */
    const float* cls;               /* [B, An, C] class logits (softmax focal) */
    float* grad_loc;                /* d loss/d loc, or NULL (then grad_cls must be NULL too): forward only */
    float* grad_cls;                /* d loss/d cls */
    const float* anchors;           /* [An, 4] (cx, cy, w, h) pixels: RetinaUtils.get_anchor_boxes */
    const float* target;            /* [nT, 6] (img, cls, cx, cy, w, h) normalised */
    int32_t B, An, C, nT;
    int32_t coord_mode;             /* ODPL_COORD_*: nn.SmoothL1Loss / nn.MSELoss, reduction='sum' */
    int32_t flags;                  /* ODPL_RETINA_STATE_CLEAN */
    float img_size;                 /* scalar target scale, losses.py:425: boxes = y[:, 2:] * img_size */
    float img_size_y;               /* EXTENSION for non-square inputs (BASELINE config 4 is 1333 x 800): when > 0 the y / h columns
                                     * are scaled by it and the x / w columns by img_size (= input width); 0 = the reference's one scalar */
    int32_t* pos_anchor;            /* ODPL_RETINA_ROWS_ONLY: int32 [B * An] (first num_pos entries written): b * An + a of the positive
                                     * anchors in ascending order -- the rows of grad_loc / grad_cls that carry a gradient; else NULL */
    int32_t* num_pos;               /* ODPL_RETINA_ROWS_ONLY: device int32 [1], the number of positive anchors; else NULL */
} odpl_retina_loss_desc_t;

/* the caller promises the last use of this workspace was a successful odpl_retina_loss call */
#define ODPL_RETINA_STATE_CLEAN 1
/* The focal one-hot drops the background column (losses.py:336-339) and SmoothL1 runs on positives only (:458-461): d loss/d (loc, cls)
 * is zero except on the num_pos positive anchors (~0.1 % of the rows at BASELINE config 4).  With this flag grad_loc / grad_cls are written
 * on those rows ONLY (no zero stream: the other rows are left untouched) and the rows are listed in pos_anchor / num_pos -- the
 * structured form odpl_retina_heads_backward consumes. */
#define ODPL_RETINA_ROWS_ONLY 2
/* Split call for a loss-driven step (odpl_pixel_support): ASSIGN_ONLY runs the target assignment (losses.py:423-446) and lists the
 * positive rows in pos_anchor / num_pos -- loc, cls and the gradients may be NULL, out3 is not written; a later call on the SAME
 * workspace, targets and anchors with ASSIGNED set skips the assignment and evaluates the loss on the predictions given then. */
#define ODPL_RETINA_ASSIGN_ONLY 4
#define ODPL_RETINA_ASSIGNED 8

size_t odpl_retina_loss_workspace_bytes(const odpl_retina_loss_desc_t* desc);

/*
 * Replaces RetinaNetLoss.forward (losses.py:405-472) including autograd backward for a unit upstream
 * gradient.  out3: loss, Localization, Classification.  out_counts int32[4]: num_pos, status
 * (bit0: an image has no box -- the reference raises there, we return NaN; bit1: label >= C), nGT, 0.
 */
int odpl_retina_loss(const odpl_retina_loss_desc_t* desc, void* workspace, size_t workspace_bytes,
                     float* out3, int32_t* out_counts, void* stream);

/*
 * RetinaUtils.get_anchor_boxes (RetinaUtils.py:6-31) as a device table.  Per level i (P3..P7) the host
 * passes fm_w, fm_h (int) and grid_w, grid_h (f32 = input/fm, computed in f32 as the reference does) and
 * the [5][9][2] f32 (w,h) table (double maths rounded once).  out: [sum fm_h*fm_w*9, 4] (cx,cy,w,h).
 */
int odpl_retina_anchors(const int32_t* host_fm_wh /*[5][2]*/, const float* host_grid_wh /*[5][2]*/,
                        const float* host_anchor_wh /*[5][9][2]*/, float* out, int64_t n_out, void* stream);

/* ---- IoU family: accuracy.py:6-114, 289-303; losses.py:375-403 ------------------------------- */

/* kind: 0 bbox_iou (accuracy.py:39-69, "+1" pixels, union+1e-16; GIoU/DIoU/CIoU kwargs ignored there),
 *       1 IoU / 2 GIoU / 3 DIoU / 4 CIoU of bbox_iou_v5 (accuracy.py:71-114), 5 iou (accuracy.py:6-37).
 * box1 is [n1,4] with n1 == n or 1 (broadcast), box2 [n,4]; layout_4xn: both stored transposed [4,n]
 * (bbox_iou_v5's convention).  x1y1x2y2 == 0: boxes are (cx,cy,w,h). */
int odpl_pair_iou(const float* box1, const float* box2, float* out, int64_t n, int64_t n1, int32_t kind,
                  int32_t x1y1x2y2, int32_t layout_4xn, void* stream);
/* bbox_wh_iou stacked over anchors (accuracy.py:297-303, 331): anchors [A,2], gwh [nT,2] -> out [A,nT] */
int odpl_bbox_wh_iou(const float* anchors, const float* gwh, float* out, int32_t A, int64_t nT, void* stream);
/* RetinaNetLoss.box_iou (losses.py:375-403): [N,4] x [M,4] -> [N,M]; xywh != 0: order='xywh' */
int odpl_box_iou_matrix(const float* box1, const float* box2, float* out, int64_t N, int32_t M, int32_t xywh, void* stream);
/* xywh2xyxy (accuracy.py:289-295) on [n,4] */
int odpl_xywh2xyxy(const float* in, float* out, int64_t n, void* stream);
/* get_dboxes (SSDUtils.py:5-27): host passes fks[6], boxes-per-cell[6] (= num_boxes+1) and the per-level
 * (w,h) table [6][6][2] (double maths rounded once); out [sum fk*fk*per, 4], clamped to <= 1. */
int odpl_ssd_priors(const int32_t* host_fk, const int32_t* host_per_cell, const float* host_wh, float* out, int64_t n_out, void* stream);

/* ---- YOLOv5 loss: accuracy.py:472-521, losses.py:37-156, accuracy.py:71-114 ------------------- */

typedef struct {
    const void* p;                        /* the [B, na, G, G, 5+nc] tensor of Yolo_Layers.forward (YOLOV5.py:93-97), fp32 or bf16, in the
                                           * memory format `layout` says */
    void* grad;                           /* same shape, dtype and memory format, or NULL on every layer: forward only */
    int32_t G;
    float anchor_w[ODPL_MAX_ANCHORS];     /* anchors / stride in f32 (losses.py:95-96) */
    float anchor_h[ODPL_MAX_ANCHORS];
    int32_t layout;                       /* ODPL_LAYOUT_INTERLEAVED (1): contiguous [B,na,G,G,no], the reference's
                                           * permute().contiguous(); ODPL_LAYOUT_PLANAR (0): the Detect convolution's NCHW output
                                           * viewed as [B,na,no,G,G] -- strides (na*no*G2, no*G2, G, 1, G2) -- i.e. the permute taken as a
                                           * VIEW (no copy; the objectness logits are contiguous planes).  Both use the cell index
                                           * (b*na + a)*G2 + s. */
} odpl_yolov5_layer_t;

typedef struct {
    int32_t nl, B, na, nc, nT;
    int32_t iou_mode;                     /* 0 IoU, 1 GIoU (what the reference calls, losses.py:118), 2 DIoU, 3 CIoU */
    int32_t flags;                        /* ODPL_YOLOV5_STATE_CLEAN */
    int32_t bf16;                         /* 1: p / grad are bfloat16 (config 5); arithmetic and sums stay fp32 */
    const float* target;                  /* [nT, 6] */
    odpl_yolov5_layer_t layer[ODPL_MAX_SCALES];
} odpl_yolov5_loss_desc_t;

#define ODPL_YOLOV5_STATE_CLEAN 1

/* bytes for the one-call entry (and for a forward with compact == NULL) */
size_t odpl_yolov5_loss_workspace_bytes(const odpl_yolov5_loss_desc_t* desc);
/* out4: loss, Localization (lbox*0.05), Classification (lcls*0.58), Conf_obj (lobj*1.0).
 * out_counts int32[ODPL_MAX_SCALES+1]: matched rows per layer, then status (bit0: a target indexes outside
 * the grid/batch -- the reference raises IndexError; the row is dropped here; bit2: dirty owner map). */
int odpl_yolov5_loss(const odpl_yolov5_loss_desc_t* desc, void* workspace, size_t workspace_bytes,
                     float* out4, int32_t* out_counts, void* stream);
/* Split the way autograd calls it (see odpl_yolo_region_loss_forward): forward leaves the STRUCTURED gradient -- the
 * objectness gradient of every cell + `no` values per matched row -- in `compact`; backward expands it into layer[i].grad
 * times *upstream; a 1x1-convolution producer (Detect.m, YOLOV5.py:93-97) takes odpl_yolov5_compact_view +
 * odpl_head_conv_backward instead and the dense tensor is never written. */
size_t odpl_yolov5_loss_compact_bytes(const odpl_yolov5_loss_desc_t* desc);
int odpl_yolov5_loss_forward(const odpl_yolov5_loss_desc_t* desc, void* workspace, size_t workspace_bytes, void* compact,
                             size_t compact_bytes, float* out4, int32_t* out_counts, void* stream);
int odpl_yolov5_loss_backward(const odpl_yolov5_loss_desc_t* desc, const void* compact, size_t compact_bytes,
                              const float* upstream, void* stream);
/* build_targets_v5's rows as the kernels produced them (accuracy.py:472-521: indices (b, a, gj, gi) = row_cell decoded with
 * cell = ((b*na + a)*G + gj)*G + gi, tbox, anchor index, target row -> tcls) -- HOST struct of DEVICE pointers into `compact` */
typedef struct {
    const int32_t* row_target;            /* [cap] row of `target` */
    const int32_t* row_anchor_group;      /* [cap] (anchor << 4) | group, group 0 = the cell itself, 1..4 = x-1, y-1, x+1, y+1 */
    const int32_t* row_cell;              /* [cap] linear cell, -1 = dropped / beyond n_rows */
    const float* row_tbox;                /* [cap, 4] (gx - gi, gy - gj, gw, gh) */
    const int32_t* n_rows;                /* [1] rows produced */
    int32_t cap;
} odpl_yolov5_rows_t;
int odpl_yolov5_compact_view(const odpl_yolov5_loss_desc_t* desc, const void* compact, int32_t layer, odpl_sparse_grad_t* out,
                             odpl_yolov5_rows_t* rows);

/* ---- backward of RetinaNet's two final head convolutions from the structured gradient (ODPL_RETINA_ROWS_ONLY) -------------------
 * cls_head[-1] / loc_head[-1] (model/RetinaNet.py:93-99: Conv2d(Cin, A * no, 3, 1, 1), shared by the FPN levels) produce cls_preds / loc_preds
 * directly (RetinaNet.py:101-113: permute(0, 2, 3, 1) + view(B, -1, no) + cat over the levels), so the rows of d loss/d preds map one to one
 * onto (level, pixel, anchor k) and only the num_pos rows of pos_anchor are non-zero.  What aten::convolution_backward computes from the
 * dense tensors, from those rows alone:  dW[k*no + c, :, ky, kx] += g[c] * X[b, y+ky-1, x+kx-1, :],  db[k*no + c] += g[c],
 * dX[b, y+ky-1, x+kx-1, :] += sum_c g[c] * W[k*no + c, :, ky, kx].  Every output is fully written (dX is zero away from the positives);
 * no float atomics.  X / dX: channels-last fp32 [B, H_l, W_l, Cin] per level, Cin % 4 == 0, min(no, 80) * Cin <= 37 000 (Cin <= 448 for
 * RetinaNet's 80-class head; SSD's 21-class heads on 1024 channels fit), A <= 10; weights through their element strides
 * (NCHW-contiguous or channels-last parameters).  The workspace query returns 0 for a descriptor outside these limits.
 * SSD (model/SSD.py:249-310: cl{i} / bbx{i} = Conv2d(Cin_i, n_i * C | n_i * 4, 3, padding=1) on six sources with their own weights,
 * channel counts and prior counts) is one call per source: nlevels = 1, the two heads hang on the SAME input (ODPL_HEADS_SHARED_INPUT:
 * one dX = both heads' sum), grad_rows point at the source's first prior inside the [B, 8732, no] tensors (rows_per_image = 8732) and
 * pos_anchor is the source's own list (odpl_ssd_level_lists). */
#define ODPL_RETINA_LEVELS 5
typedef struct {
    const float* x[ODPL_RETINA_LEVELS];      /* the convolution's input on each level */
    float* grad_x[ODPL_RETINA_LEVELS];       /* all NULL = not wanted */
    const float* w;                          /* [A * no, Cin, 3, 3] */
    float* grad_w;                           /* same strides as w; NULL = not wanted */
    float* grad_b;                           /* [A * no]; NULL = not wanted */
    const float* grad_rows;                  /* d loss/d preds in the dense [B, An, no] layout; only the rows listed in pos_anchor are read */
    int64_t w_stride[4];                     /* element strides of w / grad_w along (o, i, ky, kx) */
    int32_t no;                              /* outputs per anchor: C (class head) or 4 (box head) */
} odpl_retina_head_t;
typedef struct {
    int32_t B, A, Cin, nlevels;
    int32_t H[ODPL_RETINA_LEVELS], W[ODPL_RETINA_LEVELS];     /* feature-map sizes; An = A * sum H*W, rows ordered (level, y, x, k) */
    const int32_t* pos_anchor;               /* device int32 [>= num_pos]: b * An + a, ascending (odpl_retina_loss, ODPL_RETINA_ROWS_ONLY) */
    const int32_t* num_pos;                  /* device int32 [1] */
    const float* upstream;                   /* device scalar multiplied into every gradient; NULL = 1 */
    int32_t rows_per_image;                  /* rows per image of the grad_rows tensors when that is more than An (0 = An) */
    int32_t flags;                           /* ODPL_HEADS_* */
} odpl_retina_heads_desc_t;
#define ODPL_HEADS_SHARED_INPUT 1            /* n_heads == 2 with x / grad_x of head 1 == head 0's: grad_x receives the SUM of both heads' dX */
size_t odpl_retina_heads_backward_workspace_bytes(const odpl_retina_heads_desc_t* desc, int32_t n_heads, const odpl_retina_head_t* heads);
int odpl_retina_heads_backward(const odpl_retina_heads_desc_t* desc, int32_t n_heads, const odpl_retina_head_t* heads,
                               void* workspace, size_t workspace_bytes, void* stream);
/* positive rows that own staging slots in the workspace (default 4096: 9 x 2 x Cin floats each); rows beyond are summed in place by the
 * dx launch -- same arithmetic, slower.  Call before querying the workspace size. */
int odpl_retina_heads_tune(int32_t stage_rows);

/* ---- backward of a 3x3 (stride 1, pad 1, ungrouped) convolution over the pixels that carry a gradient (csrc/odpl_sparseconv.hip) ----
 * The head towers of model/RetinaNet.py:93-99 receive a gradient that is zero on most pixels (it spreads one pixel per layer from the
 * positive anchors).  `active` [B*H*W] bytes: non-zero where the row of grad_y may be non-zero (odpl_bias_act_backward_flags emits it);
 * rows flagged inactive MUST be zero in grad_y.  grad_x is fully written (zero away from the active pixels' neighbourhoods), grad_w has
 * the strides of w; either may be NULL.  What aten::convolution_backward computes, on the active pixels only; no float atomics.
 * x / grad_x: channels-last [B, H, W, Cin], grad_y: [B, H, W, Cout], fp32; Cin, Cout multiples of 64. */
typedef struct {
    const float* x; const float* gy; const uint8_t* active; const float* w;
    float* grad_x; float* grad_w;
    int64_t w_stride[4];                     /* element strides of w / grad_w along (o, i, ky, kx) */
    int32_t B, H, W, Cin, Cout;
    int32_t ksize;                           /* 0 or 3: 3 x 3, pad 1 (the tower units, the FPN smoothing convolutions); 1: 1 x 1, pad 0 (the FPN laterals) */
    int32_t* stats;                          /* optional device int32 [2]: active pixels, pixels within one pixel of an active one -- what the host
                                              * reads LATER (asynchronously) to decide whether this route still pays; NULL = not wanted */
} odpl_sparse_conv3x3_t;
size_t odpl_sparse_conv3x3_backward_workspace_bytes(const odpl_sparse_conv3x3_t* desc);
int odpl_sparse_conv3x3_backward(const odpl_sparse_conv3x3_t* desc, void* workspace, size_t workspace_bytes, void* stream);

/* ---- loss-driven forward of the RetinaNet heads (csrc/odpl_sparsefwd.hip) ---------------------------------------------------
 * step.py:14-25 is forward -> criterion and nothing else reads the predictions; RetinaNetLoss reads its POSITIVE rows only
 * (losses.py:336-339, 458-461).  Assignment needs anchors and boxes only, so a training step can run it first
 * (odpl_retina_loss, ODPL_RETINA_ASSIGN_ONLY), turn the positive rows into per-level pixel lists (odpl_pixel_support) and evaluate
 * the heads' 3x3 convolutions (model/RetinaNet.py:93-99) on those pixels only (odpl_sparse_conv3x3_forward): same values there as
 * the dense convolution, nothing computed elsewhere. */
typedef struct {
    const int32_t* pos_anchor;               /* device: b * An + a of the positive anchors (odpl_retina_loss, ODPL_RETINA_ASSIGN_ONLY) */
    const int32_t* num_pos;                  /* device int32 [1] */
    int32_t B, An, A, n_levels;              /* A anchors per pixel; An = A * sum_l H[l] * W[l] (checked) */
    int32_t H[8], W[8];                      /* the level maps, in the order of the anchor table (RetinaUtils.py:6-31) */
    int32_t depth;                           /* D <= 8: lists for reach 0 .. D-1 */
    int32_t level_mask;                      /* bit l set: lists wanted for level l */
    int32_t* lists;                          /* out, device int32 [D * B * sum_l H W]: level l owns D slots of B*H[l]*W[l] entries starting at
                                              * D * B * sum_{l' < l} H W; slot j = the pixels (b*H + y)*W + x of that level within Chebyshev
                                              * distance j of a pixel that owns a positive anchor -- in NO defined order (work queues) */
    int32_t* counts;                         /* out, device int32 [n_levels * D]: entries of slot (l, j) */
} odpl_pixel_support_t;
size_t odpl_pixel_support_workspace_bytes(const odpl_pixel_support_t* desc);
int odpl_pixel_support(const odpl_pixel_support_t* desc, void* workspace, size_t workspace_bytes, void* stream);

typedef struct {
    const float* x;                          /* [B, Cin, H, W] channels-last (dense), Cin % 32 == 0, 16-byte aligned */
    const float* w;                          /* [Cout, Cin, 3, 3] in any dense layout (w_stride) */
    const float* bias;                       /* [Cout] or NULL */
    int64_t w_stride[4];
    float* y;                                /* y[b * y_batch_stride + (yy * W + xx) * y_pixel_stride + co]: a channels-last map
                                              * (H*W*Cout, Cout) or a slice of the [B, An, no] prediction table (An*no, A*no) */
    int64_t y_batch_stride, y_pixel_stride;
    const int32_t* list;                     /* device: pixels (b*H + y)*W + x to evaluate */
    const int32_t* count;                    /* device int32 [1]: how many */
    int32_t B, H, W, Cin, Cout;
    int32_t act;                             /* ODPL_ACT_LINEAR / _RELU / _LEAKY on conv + bias */
    int32_t ksize;                           /* 0 or 3: 3 x 3, pad 1; 1: 1 x 1, pad 0 */
} odpl_sparse_conv3x3_fwd_t;
/* K is always summed as 8 consecutive ranges added left to right.  workspace (optional, >= 1 MiB, 16-byte aligned; NULL = none): a launch whose
 * list is short (<= 128 output tiles) gives each range its own workgroup and adds the partials in the same order in a second launch --
 * same bits either way; 32 MiB covers every such case at Cout <= 1024 */
int odpl_sparse_conv3x3_forward(const odpl_sparse_conv3x3_fwd_t* desc, void* workspace, size_t workspace_bytes, void* stream);

/* The FPN's top-down path under the same restriction (model/backbone/RetinaNetbone.py:55-75: p4 = toplayer1(up(p5) + latlayer2(c4)),
 * p3 = toplayer2(up(p4) + latlayer3(c3)); `F.upsample(..., mode='bilinear')`, align_corners = False).
 * odpl_pixel_union_lists: the pixels of an H x W map that are in list_a (may be NULL) or are a bilinear source of a pixel in list_b
 * (pixels of a finer Hf x Wf map; may be NULL) -> list_marked (ascending), list_near = those within one pixel of a marked one
 * (ascending; what a 3 x 3 convolution evaluated on list_marked reads), counts2 = their sizes.  Lists: device int32 [B*H*W].
 * odpl_sparse_upsample_add: out[q,:] = lateral[q,:] + bilinear(coarse)[q,:] on the listed pixels q of the fine map (channels-last fp32,
 * C % 4 == 0; ATen's source-index arithmetic and operation order); nothing is written elsewhere. */
size_t odpl_pixel_union_lists_workspace_bytes(int32_t B, int32_t H, int32_t W);
int odpl_pixel_union_lists(const int32_t* list_a, const int32_t* count_a, const int32_t* list_b, const int32_t* count_b,
                           int32_t B, int32_t H, int32_t W, int32_t Hf, int32_t Wf,
                           int32_t* list_marked, int32_t* list_near, int32_t* counts2,
                           void* workspace, size_t workspace_bytes, void* stream);
int odpl_sparse_upsample_add(const float* coarse, const float* lateral, float* out, const int32_t* list, const int32_t* count,
                             int32_t B, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W, void* stream);

/* ---- SSD loss: losses.py:159-315 ---------------------------------------------------------------- */

typedef struct {
    const float* off;                     /* [B, P, 4] predicted offsets */
    const float* cls;                     /* [B, P, C] class logits */
    float* grad_off;                      /* or NULL together with grad_cls: forward only */
    float* grad_cls;
    const float* priors;                  /* [P, 4] (cx,cy,w,h) in [0,1]: SSDUtils.get_dboxes */
    const float* target;                  /* [nT, 6], nT >= 1 */
    int32_t B, P, C, nT;
    int32_t cls_mode;                     /* ODPL_CLS_CE (nn.CrossEntropyLoss) or ODPL_CLS_SOFTMAX_FOCAL (focal_loss) */
    int32_t coord_mode;                   /* ODPL_COORD_* with reduction='none' then .sum() */
    int32_t flags;                        /* ODPL_SSD_ROWS_ONLY */
    int32_t* active;                      /* ODPL_SSD_ROWS_ONLY: out, int32 [B, P]: image b's priors that carry a gradient, ascending, in [b*P, b*P + count) */
    int32_t* active_count;                /* ODPL_SSD_ROWS_ONLY: out, int32 [B] */
} odpl_ssd_loss_desc_t;
/* grad_off / grad_cls are written on the rows of the positives and the selected hard negatives only (d loss/d preds is zero elsewhere:
 * losses.py:241-268 -- and those zeros are neither written nor read); `active` lists them.  For the structured backward of the heads. */
#define ODPL_SSD_ROWS_ONLY 1
/* `active` cut by source: list l receives b * An_l + (p - first[l]) for the active priors first[l] <= p < first[l+1] (ascending: the
 * pos_anchor of odpl_retina_heads_backward for source l), at lists + B * first[l]; counts[l] their number.  first: HOST int32 [nlevels + 1],
 * first[nlevels] = P; nlevels <= 8. */
int odpl_ssd_level_lists(const int32_t* active, const int32_t* active_count, int32_t B, int32_t P, int32_t nlevels, const int32_t* first,
                         int32_t* lists, int32_t* counts, void* stream);

#define ODPL_CLS_SOFTMAX_FOCAL 2

size_t odpl_ssd_loss_workspace_bytes(const odpl_ssd_loss_desc_t* desc);
/* out3: loss, Localization, Classification.  out_counts int32[2]: new_batch_size, status (bit0: class id
 * outside the criterion's range -- the reference raises). */
int odpl_ssd_loss(const odpl_ssd_loss_desc_t* desc, void* workspace, size_t workspace_bytes,
                  float* out3, int32_t* out_counts, void* stream);

/* ---- batched NMS ------------------------------------------------------------------------------ */

#define ODPL_NMS_STATE_CLEAN 1      /* last use of the workspace was a successful call of the same entry */

typedef struct {
    const float* head[ODPL_MAX_SCALES];   /* raw heads [B, A*no, G, G] exactly as forward() returns them */
    int32_t G[ODPL_MAX_SCALES];
    int32_t nscales, B, A, no;            /* no = 5 + C ("answers", YOLOV4.py:240) */
    float conf_thres;                     /* -0.0151: YOLOV4.py:228 overwrites the argument */
    float nms_thres;                      /* 0.4 */
    int32_t flags;                        /* ODPL_YOLO_NMS_ONE_WORKGROUP or 0 */
} odpl_yolo_nms_desc_t;
/* flags: keep the one-workgroup-per-image kernel also for networks with >= 8 classes (default there: class histogram -> one
 * workgroup per (image, class) -> rank order of the surviving rows; same output) */
#define ODPL_YOLO_NMS_ONE_WORKGROUP 1

size_t odpl_yolo_nms_workspace_bytes(const odpl_yolo_nms_desc_t* desc);
/*
 * YOLO confidence-weighted merge NMS, model/YOLOV4.py:221-283 (== YOLOV3.py:273-335, YOLOV2.py:159-222 with
 * A = 5, YOLOV5.py:157-218: pass its [B,3,G,G,no] buffer as-is with A = 3 -- the reference reinterprets that
 * memory as [B,3,no,G,G]).  out_dets [B, N, 7] (x1,y1,x2,y2,conf,cls_conf,cls_pred), N = sum A*G*G; the first
 * out_counts[b] rows of image b are valid; out_counts[b] == -1 <=> the reference leaves output[b] = None.
 */
int odpl_yolo_nms(const odpl_yolo_nms_desc_t* desc, void* workspace, size_t workspace_bytes,
                  float* out_dets, int32_t* out_counts, void* stream);

typedef struct {
    const float* loc;                     /* [B, An, 4] */
    const float* cls;                     /* [B, An, C] logits */
    const float* anchors;                 /* [An, 4] cx,cy,w,h */
    int32_t B, An, C, topk;               /* topk = 100 */
    float nms_thresh;                     /* 0.5 */
    float class_thresh;                   /* 0.45 */
    int32_t flags;
} odpl_greedy_nms_desc_t;

size_t odpl_greedy_nms_workspace_bytes(const odpl_greedy_nms_desc_t* desc);
/*
 * RetinaNet / SSD non_max_suppression, model/RetinaNet.py:117-178 == model/SSD.py:249-310 (mode='union').
 * out_dets [B, topk, 7] (x1,y1,x2,y2,0,score,label); out_counts[b*2] = kept rows, out_counts[b*2+1] = number of
 * candidates above class_thresh (the reference raises IndexError when that is exactly 1).
 */
int odpl_greedy_nms(const odpl_greedy_nms_desc_t* desc, void* workspace, size_t workspace_bytes,
                    float* out_dets, int32_t* out_counts, void* stream);

/* ---- YOLOv4 inference decode: utils/YoloV4Utils.py:18-176, model/YOLOV4.py:41-49 ------------ */

typedef struct {
    const float* head;                    /* [B, A*(5+C), H, W] */
    int32_t H, W, A;
    float anchor_w[ODPL_MAX_ANCHORS];     /* masked anchors / stride (python double division, rounded once) */
    float anchor_h[ODPL_MAX_ANCHORS];
} odpl_yolo_decode_scale_t;

typedef struct {
    int32_t nscales, B, C;
    float scale_x_y;                      /* 1 in the reference's YoloLayer */
    float scale_x_y_offset;               /* f32(0.5 * (scale_x_y - 1)) computed by the host in double */
    odpl_yolo_decode_scale_t scale[ODPL_MAX_SCALES];
} odpl_yolo_decode_desc_t;

/* boxes [B, N, 1, 4] (x1,y1,x2,y2 normalised), confs [B, N, C] = sigmoid(cls)*sigmoid(det), N = sum A*H*W,
 * scales concatenated in order (get_region_boxes). */
int odpl_yolo_decode(const odpl_yolo_decode_desc_t* desc, float* boxes, float* confs, void* stream);

/* ---- evaluation statistics: accuracy.py:116-154, 207-287 ---------------------------------------- */

/* get_batch_statistics: dets [B, Kmax, 7] with counts[b] valid rows (<= 0: image skipped, the reference's None),
 * targets [nT, 6] whose columns 2:6 are compared as corners (accuracy.py:132-148) -> tp [B, Kmax] (1.0 / 0.0).
 * status int32[1]: bit0 = more than 1024 annotations on one image (truncated). */
int odpl_batch_statistics(const float* dets, const int32_t* counts, const float* targets, int32_t B, int32_t Kmax,
                          int32_t nT, float iou_threshold, float* tp, int32_t* status, void* stream);

size_t odpl_ap_per_class_workspace_bytes(int32_t N, int32_t M);
/* ap_per_class + compute_ap: tp f64[N], conf f32[N], pred_cls f32[N], target_cls f32[M] (all device) ->
 * p, r, ap, f1 f64[<= M] and classes int32[<= M] for the *out_num_classes unique target classes (ascending). */
int odpl_ap_per_class(const double* tp, const float* conf, const float* pred_cls, const float* target_cls,
                      int32_t N, int32_t M, void* workspace, size_t workspace_bytes,
                      double* out_p, double* out_r, double* out_ap, double* out_f1, int32_t* out_classes,
                      int32_t* out_num_classes, void* stream);

/* get_yolo_statistics (accuracy.py:382-470), one scale: (1) training-style decode of the raw head
 * [B, A*(5+C), G, G] into pred_boxes [B,A,G,G,4] (grid units), pred_conf [B,A,G,G], pred_cls [B,A,G,G,C] and
 * output [B, A*G*G, 5+C] (boxes*stride, conf, cls); (2) odpl_yolo_build_targets on them; (3) the six metrics
 * out6 = cls_acc, recall50, recall75, precision, conf_obj, conf_noobj. */
int odpl_yolo_train_decode(const float* head, int32_t B, int32_t A, int32_t C, int32_t G, const float* host_anchor_wh,
                           float stride, float* pred_boxes, float* pred_conf, float* pred_cls, float* output, void* stream);
int odpl_yolo_stats_reduce(const float* pred_conf, const float* class_mask, const float* iou_scores, const uint8_t* obj_mask,
                           const uint8_t* noobj_mask, int64_t cells, float* out6, void* stream);

/* ---- SPP block of the YOLOv4 neck: utils/YoloV4Utils.py:400-420 ----------------- */
/* y[B, 4C, H, W] = cat(maxpool13(x), maxpool9(x), maxpool5(x), x) (stride 1, same padding); strides are ELEMENT
 * strides (b, c, h, w) of dense tensors in any memory format (NCHW or channels-last); H*W <= 1024.
 * grad_x has x's strides, grad_y has y's. */
int odpl_spp_forward(const float* x, const int64_t* host_x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                     float* y, const int64_t* host_y_strides, void* stream);
int odpl_spp_backward(const float* x, const int64_t* host_x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                      const float* grad_y, const int64_t* host_y_strides, float* grad_x, void* stream);
/* Same block with the channel order as a parameter -- ODPL_SPP_ORDER_V4: [m13 | m9 | m5 | x] (utils/YoloV4Utils.py:418),
 * ODPL_SPP_ORDER_V5: [x | m5 | m9 | m13] (utils/YoloV5Utils.py:110) -- and optionally bfloat16 tensors (bf16 != 0: dense
 * channels-last only, C % 16 == 0, H*W <= 400; max is exact, so only the backward sums round, once). */
#define ODPL_SPP_ORDER_V4 0
#define ODPL_SPP_ORDER_V5 1
int odpl_spp_forward_ex(const void* x, const int64_t* host_x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                        void* y, const int64_t* host_y_strides, int32_t bf16, int32_t order, void* stream);
int odpl_spp_backward_ex(const void* x, const int64_t* host_x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                         const void* grad_y, const int64_t* host_y_strides, void* grad_x, int32_t bf16, int32_t order, void* stream);

/* ---- device-side collate: dataset/Pascal.py:121-186 == Coco.py:146-220 ------------------------- */
/* One raw frame in HBM: uint8 [h, w, 3] interleaved (cv2.imread layout, BGR).  flip: bit 0 horizontal, bit 1 vertical
 * (the host draws them, p = 0.2 each in the reference). */
typedef struct {
    const uint8_t* data;
    int32_t h, w;
    int32_t flip;
    int32_t reserved;     /* bit 0: warp this image (odpl_collate_images_warp) */
    const uint8_t* lut;   /* NULL, or device uint8 [3][256]: per-channel point operation applied to the resized uint8 frame (the
                           * host composes RandomBrightnessContrast / RGBShift -- LUT operations in albumentations -- into it) */
} odpl_image_t;
/* x[b] = CHW( flip( resize_linear_u8(image b -> S x S) ) ) / 255, written with the given element strides (b, c, h, w). */
int odpl_collate_images(const odpl_image_t* host_images, int32_t B, int32_t S, float* x,
                        const int64_t* host_x_strides, void* stream);
/* objects [nT, 5] = (x, y, w, h, label) in source pixels (COCO), row_image [nT] = image of each row, image_dims [B, 3] =
 * (h, w, flip) (device arrays) -> y [nT, 6] = (img, label, x, y, w, h) normalised, after the flips. */
int odpl_collate_targets(const float* objects, const int32_t* row_image, const int32_t* image_dims, int32_t nT,
                         int32_t S, float* y, void* stream);
/* The same two entries with A.ShiftScaleRotate (dataset/Pascal.py:217, Coco.py:259: cv2.warpAffine, INTER_LINEAR,
 * BORDER_REFLECT_101, applied to the resized + flipped uint8 frame, before the photometric tables).  Images with bit 0 of
 * `reserved` set are warped by host_inverse_affine[b] = the 2x3 DESTINATION -> SOURCE map warpAffine inverts its argument
 * to (doubles, pixels of the S x S frame); OpenCV's fixed-point arithmetic (1/32-pixel positions, 2^15 bilinear weights).
 * Targets: image_affine [B][7] doubles (device) = flag, then the FORWARD 2x3 matrix; boxes become the axis-aligned hull of
 * their transformed corners, are clipped to the frame, and keep[t] = 0 marks rows albumentations drops (no area left).
 * keep may be NULL. */
int odpl_collate_images_warp(const odpl_image_t* host_images, const double* host_inverse_affine, int32_t B, int32_t S, float* x,
                             const int64_t* host_x_strides, void* stream);
int odpl_collate_targets_warp(const float* objects, const int32_t* row_image, const int32_t* image_dims, const double* image_affine,
                              int32_t nT, int32_t S, float* y, uint8_t* keep, void* stream);

/* Bias gradient of a convolution from an NCHW-dense output gradient [B, C, HW]: grad_bias[c] = sum over (b, hw)
 * (what ATen's convolution_backward computes with sum((0, 2, 3)); the linear head convs of model/YOLOV4.py:89-127). */
int odpl_bias_grad_nchw(const void* grad_y, int32_t bf16 /* grad_y is bfloat16; grad_bias stays fp32 */, int32_t B, int32_t C, int64_t HW,
                        float* grad_bias, void* stream);

/* ---- space-to-depth shuffles next to the stems/heads ------------------------------------------ */
/* Focus, utils/YoloV5Utils.py:117-125: y[b, g*C+c, i, j] = x[b, c, 2i+(g&1), 2j+(g>>1)]; x is [B,C,H,W], y is
 * [B,4C,H/2,W/2]; element strides (b,c,h,w) of dense tensors in any memory format.  inverse = 0: src = x, dst = y;
 * inverse = 1: src = grad_y, dst = grad_x (the same permutation walked backwards). */
int odpl_focus(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
               const int64_t* host_x_strides, const int64_t* host_y_strides, int32_t inverse, void* stream);
/* YOLOv2 passthrough "reorg", model/YOLOV2.py:147-151: x is the CONTIGUOUS NCHW block [B,C,H,W] (C % 4 == 0), reinterpreted
 * as [B,C/4,H,2,W,2] -> permute(0,3,5,1,2,4) -> [B,4C,H/2,W/2] with the given y strides. */
int odpl_reorg(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
               const int64_t* host_y_strides, int32_t inverse, void* stream);

/* Convolution + bias + activation units without a BatchNorm (RetinaNet's heads, model/RetinaNet.py:93-99: Conv2d(256,256,3) + ReLU):
 * the bias-free convolution stays on MIOpen; y = act(x + bias) on the channels-last fp32 [R = N*H*W, C] view, in place when y == x.
 * `ones` / `zeros`: device vectors of C ones / zeros (caller-owned constants). */
int odpl_bias_act_forward(const float* x, int64_t R, int32_t C, const float* bias, const float* ones, const float* zeros,
                          int32_t act, float* y, void* stream);
/* grad_x = grad_y * act'(y) and grad_bias = its sum over the rows, from the saved OUTPUT y in one pass + a finalize launch
 * (act: ODPL_ACT_LINEAR / _LEAKY / _RELU).  workspace: odpl_bn_act_workspace_bytes(R, C) + 4*C bytes; grad_x may alias grad_y. */
int odpl_bias_act_backward(const float* y, const float* grad_y, int64_t R, int32_t C, int32_t act, float* grad_x, float* grad_bias,
                           void* workspace, size_t workspace_bytes, void* stream);
/* the same + `active` [R] bytes: 1 where the row of grad_x holds a non-zero, 0 elsewhere (the pixel map odpl_sparse_conv3x3_backward walks) */
int odpl_bias_act_backward_flags(const float* y, const float* grad_y, int64_t R, int32_t C, int32_t act, float* grad_x, float* grad_bias,
                                 uint8_t* active, void* workspace, size_t workspace_bytes, void* stream);

/* Nearest-neighbour 2x upsampling of the necks on DENSE channels-last tensors: utils/YoloV4Utils.py:191-206 (`Upsample`,
 * F.interpolate(size, 'nearest') in training), nn.Upsample(scale_factor=2) of model/YOLOV3.py, utils/YoloV5Utils.py's neck.
 * The small tensor is [B,C,H,W] (channels-last, dense), the large one [B,*,2H,2W] channels-last with `big_pixel_stride` elements
 * per pixel and the C channels starting at `big_channel_offset` (stride == C and offset 0 for a plain result; a channel slice of a
 * `torch.cat` buffer otherwise).  fp32 (C % 4 == 0) or bf16 (C % 8 == 0), 16-byte aligned pointers, B*H <= 65535.
 * inverse = 0: src = x (small), dst = y (large).  inverse = 1: src = grad_y (large), dst = grad_x (small) = the sum of the four
 * replicas, accumulated in fp32 in the order (0,0) (0,1) (1,0) (1,1). */
int odpl_upsample2x_nearest(const void* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t bf16,
                            int64_t big_pixel_stride, int64_t big_channel_offset, int32_t inverse, void* stream);

/* ---- fused BatchNorm2d + activation on channels-last fp32: utils/YoloV4Utils.py:208-236 ---- */
#define ODPL_ACT_LINEAR 0
#define ODPL_ACT_MISH 1
#define ODPL_ACT_LEAKY 2             /* LeakyReLU(0.1) */
#define ODPL_ACT_RELU 3
size_t odpl_bn_act_workspace_bytes(int64_t R, int32_t C);
/* x, y: row-major [R = N*H*W, C] (a channels-last NCHW tensor), C % 4 == 0.  training != 0: batch statistics are
 * computed, written to save_mean / save_invstd and folded into running_mean / running_var (momentum, unbiased var);
 * training == 0: the caller pre-fills save_mean / save_invstd from the running statistics (needed when a backward will
 * follow), or passes both as NULL (inference) and the apply pass derives them from running_mean / running_var itself. */
int odpl_bn_act_forward(const float* x, int64_t R, int32_t C, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, int32_t training, float momentum, float eps,
                        int32_t act, float* save_mean, float* save_invstd, float* y, void* workspace,
                        size_t workspace_bytes, int64_t* num_batches_tracked /* device int64[1] or NULL: += 1 */, void* stream);
/* Same, with the residual add of the reference's ResBlock (utils/YoloV4Utils.py:239-262: `x = x + h`) folded into the
 * write: y = act(bn(x)) + residual (residual: same layout as y, or NULL).  Backward is unchanged: the residual's gradient
 * is grad_y itself. */
int odpl_bn_act_forward_residual(const float* x, const float* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, int32_t training, float momentum, float eps,
                                 int32_t act, float* save_mean, float* save_invstd, float* y, void* workspace,
                                 size_t workspace_bytes, int64_t* num_batches_tracked, void* stream);
/* The tail of a ResNet bottleneck (RetinaNet's backbone, model/backbone/RetinaNetbone.py: `F.relu(bn3(conv3(out)) + shortcut(x))`):
 * y = act(bn(x) + residual) in one write.  Backward: gu = grad_y * act'(y) (also the shortcut's gradient), then
 * odpl_bn_act_backward with ODPL_ACT_LINEAR on gu. */
int odpl_bn_add_act_forward(const float* x, const float* residual, int64_t R, int32_t C, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, int32_t training, float momentum, float eps,
                            int32_t act, float* save_mean, float* save_invstd, float* y, void* workspace,
                            size_t workspace_bytes, int64_t* num_batches_tracked, void* stream);
/* ... and its backward: the derivative of the activation is taken from the saved OUTPUT y (relu / leaky: sign of y == sign of the
 * pre-activation; linear: none); grad_residual = grad_y * act'(y) is d loss / d residual (written by the reduce pass, which sums it in the
 * same sweep); then finalize and the dx pass -- `aten::threshold_backward` + the BN backward of the reference's autograd graph in three
 * launches instead of four, seven passes over the map instead of eight.  fp32, channels-last [R, C]; workspace: odpl_bn_act_workspace_bytes. */
int odpl_bn_add_act_backward(const float* x, const float* y, const float* grad_y, int64_t R, int32_t C, const float* gamma,
                             const float* beta, const float* save_mean, const float* save_invstd, int32_t training, int32_t act,
                             float* grad_x, float* grad_residual, float* grad_gamma, float* grad_beta,
                             void* workspace, size_t workspace_bytes, void* stream);
int odpl_bn_act_backward(const float* x, const float* grad_y, int64_t R, int32_t C, const float* gamma, const float* beta,
                         const float* save_mean, const float* save_invstd, int32_t training, int32_t act,
                         float* grad_x, float* grad_gamma, float* grad_beta, void* workspace, size_t workspace_bytes,
                         void* stream);
/* Strided variants: y (forward) / grad_y (backward) are C-wide column slices of a wider row-major matrix with the given row
 * stride in elements (multiple of 4, slice start 16-byte aligned) -- the epilogue writes straight into its half of a
 * `torch.cat((a, b), 1)` buffer (the CSP merges of utils/YoloV4Utils.py:264-398) and the backward reads its half of the
 * concatenated gradient, so neither the cat copy nor the slice copies exist.  residual requires a dense y. */
/* bf16 != 0: x / residual / y (forward) and x / grad_y / grad_x (backward) are bfloat16 (the activations of a bf16-autocast
 * network, BASELINE config 5); gamma / beta / statistics / their gradients stay fp32, all arithmetic is fp32 and results are
 * rounded to bf16 once (after the activation), 8-byte alignment suffices. */
int odpl_bn_act_forward_strided(const void* x, const void* residual, int32_t bf16, int64_t R, int32_t C, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, int32_t training, float momentum, float eps,
                                int32_t act, float* save_mean, float* save_invstd, void* y, int64_t y_row_stride,
                                void* workspace, size_t workspace_bytes, int64_t* num_batches_tracked, void* stream);
int odpl_bn_act_backward_strided(const void* x, const void* grad_y, int64_t gy_row_stride, int32_t bf16, int64_t R, int32_t C,
                                 const float* gamma, const float* beta, const float* save_mean, const float* save_invstd,
                                 int32_t training, int32_t act, void* grad_x, float* grad_gamma, float* grad_beta,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- Adam over every parameter in one launch: LightningFunc/optimizer.py:6-38 (torch.optim.Adam, weight_decay as an L2 term) -------
 * The arithmetic of ATen's fused Adam kernel (fp32 parameters, double hyper-parameters, bias corrections from the device step count),
 * one flat launch instead of ten multi-tensor ones.  The caller builds the tables ONCE (they hold what never changes: where each
 * parameter, its exp_avg / exp_avg_sq and its step count live, and which tensor each ODPL_ADAM_BLOCK-element block belongs to); the
 * gradient pointers, which autograd renews every step, are a HOST array and travel as kernel arguments (ODPL_ADAM_MAX_TENSORS per
 * launch: more tensors take more launches).  The step counts are incremented by the caller BEFORE the call (torch._foreach_add_). */
#define ODPL_ADAM_BLOCK 4096
#define ODPL_ADAM_MAX_TENSORS 384
typedef struct {
    const uint64_t* table;          /* device [n_tensors][4]: parameter, exp_avg, exp_avg_sq, step (float32 scalar) device addresses */
    const uint32_t* numel;          /* device [n_tensors] */
    const uint32_t* blk0;           /* device [n_tensors + 1]: first block of each tensor, blocks of ODPL_ADAM_BLOCK elements */
    const uint32_t* blk_tensor;     /* device [n_blocks]: the tensor of each block */
    const uint32_t* host_blk0;      /* HOST copy of blk0 (launch geometry) */
    int32_t n_tensors, n_blocks;
    int32_t maximize;
    double lr, beta1, beta2, eps, weight_decay;
} odpl_adam_desc_t;
int odpl_adam_step(const odpl_adam_desc_t* desc, const float* const* host_grads, void* stream);

/* launch-geometry knobs (process-wide; defaults are the tuned values) */
#define ODPL_TUNE_YOLO_FILL_F4_PER_BLOCK 0
#define ODPL_TUNE_YOLO_FILL_NT 1
#define ODPL_TUNE_RETINA_FILL_NT 2
#define ODPL_TUNE_BN_TARGET_BLOCKS 3      /* row tiles of the BN reduce passes (default 1024) */
#define ODPL_TUNE_BN_STREAM_MAX_BLOCKS 4  /* grid cap of the BN apply / dx streams (0 = by size: 4096 / 16384) */
#define ODPL_TUNE_BN_MAX_ROWS_PER_TILE 5  /* rows per reduce tile at most (default 2048) */
#define ODPL_TUNE_BN_SLICE_FROM_C 6       /* BN reduce passes cut rows into 16-quad channel slices from this C on (default 256) */
#define ODPL_TUNE_YOLO_FUSED_MAX_TARGETS 7 /* region loss: assignment + loss are ONE launch up to this many target rows (default and at most 512);
                                           * 0: always the assign launch + the loss launch over the global state map */
int odpl_tune_set(int32_t key, int32_t value);
/* diagnostic: when non-NULL, every workgroup w of yolo_expand_kernel stores its start / end reading of the
 * constant 100 MHz counter into dev_u64[2w], dev_u64[2w+1] (buffer >= 2 * grid entries); NULL disables it. */
int odpl_debug_set_stamp_buffer(void* dev_u64);

/* ---- in-library kernel timing (bench.py roofline leg) ------------------------------------------
 * HIP events recorded on the launch stream immediately around the named kernel of every later call.
 * No counterpart in the reference (it has no benchmark); host_ms is a HOST buffer.               */
#define ODPL_PROF_YOLO_EXPAND 0      /* yolo_expand_kernel: the dense gradient stream of the generic backward */
#define ODPL_PROF_YOLO_ASSIGN 1
#define ODPL_PROF_YOLO_LOSS 2        /* yolo_loss_kernel: confidence blocks + obj-cell blocks */
#define ODPL_PROF_YOLO_FINALIZE 3
#define ODPL_PROF_RETINA_ASSIGN 4
#define ODPL_PROF_RETINA_MAIN 5
#define ODPL_PROF_YOLO_NMS_PREP 6
#define ODPL_PROF_YOLO_NMS_IMAGE 7
#define ODPL_PROF_GREEDY_NMS_PREP 8
#define ODPL_PROF_GREEDY_NMS_IMAGE 9
#define ODPL_PROF_YOLO_DECODE 10
#define ODPL_PROF_YOLOV5_DENSE 11
#define ODPL_PROF_HEADGRAD_STREAM 12  /* headgrad_stream_kernel: X read + dX write of the fused head backward */
#define ODPL_PROF_HEADGRAD_FINISH 13
#define ODPL_PROF_YOLO_PATCH 14
#define ODPL_PROF_YOLOV5_OBJ 15
#define ODPL_PROF_NUM_KERNELS 16
int odpl_profile_enable(int32_t kernel_id, int32_t on);
/* Synchronises the recorded events, writes up to max_n durations (ms) and returns how many launches
 * were recorded since the last collect (negative: ODPL_EINVAL).  Three events per launch: an empty
 * pre->start interval prices the event pair itself; host_ms = (stop-start) - (start-pre), raw_ms
 * (may be NULL) = stop-start uncorrected. */
int odpl_profile_collect(int32_t kernel_id, float* host_ms, float* raw_ms, int32_t max_n);

#ifdef __cplusplus
}
#endif
#endif /* ODPL_HIP_H */
