/* yolo_fastest_hip.h -- C ABI of libyolo_fastest_hip.so (MI355X / gfx950).
 *
 * The reference (JunFenngZhi/YOLO-Fastest-and-Embedded-deployment) has no FFI for its inference
 * path: the boundary is its Python surface.  Each entry point below replaces the body of one
 * Python-level call of the reference; the Python host code in
 * yolo-fastest-and-embedded-deployment_amd/ keeps the reference's names and argument meaning and
 * binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 *   reference interface (file:line under the reference repo)            entry point
 *   ------------------------------------------------------------------  -------------------
 *   YoloFastest(io_params).to(dev).eval() + load_state_dict(torch.load) yf_create
 *       src/model_training/model/yolo_fastest.py:70-148, src/detect.py:89-91
 *   model(img) -> (head_large, head_small)                               yf_forward
 *       src/model_training/model/yolo_fastest.py:150-218, src/detect.py:152
 *   YOLO_post_process.decode_box + class bucketing + sort + NMS          yf_decode_nms
 *       src/detect.py:41-84, :157-169   (+ __adjust_coord :131-139 when origin_* given)
 *   YOLO_post_process.non_maxium_supression(sorted_list)                 yf_nms_sorted
 *       src/detect.py:69-84
 *   Detect_YOLO.batch_detect's timed region (model + post-process)       yf_detect
 *       src/detect.py:151-171
 *   Detect_YOLO.__pre_process's arithmetic ((u8-128)/255, 2x2 area mean) yf_preprocess_u8
 *       src/detect.py:107-129
 *
 * Conventions: plain pointers and sizes only.  Every pointer named d_* is DEVICE memory owned by the
 * caller (a torch tensor's data_ptr()); `stream` is a hipStream_t passed as void*.  All launches are
 * stream-ordered and asynchronous; nothing here synchronises the device except yf_create /
 * yf_destroy.  Return value: 0 = ok, negative = error (see YF_E_*); yf_last_error_string() gives
 * the message of the calling thread's last failure.  A handle is bound to one device and is not
 * thread-safe (one handle per device per thread, as the reference's caller is single-threaded).
 */
#ifndef YOLO_FASTEST_HIP_H
#define YOLO_FASTEST_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct yf_engine *yf_handle;

enum {
    YF_OK = 0,
    YF_E_INVALID = -1,   /* bad argument (shape not multiple of 32, N > max_batch, null pointer, ...) */
    YF_E_BLOB = -2,      /* packed-weights blob does not describe YoloFastest (strict layout check)  */
    YF_E_HIP = -3,       /* a HIP runtime call failed                                                */
    YF_E_WORKSPACE = -4, /* caller's workspace too small                                             */
    YF_E_NOPROBE = -5    /* yf_forward_probe: tensor of that name never exists in device memory      */
};

/* ABI version of this header; yf_abi_version() must return the same number. */
#define YF_ABI_VERSION 1
int yf_abi_version(void);

/* Thread-local message of the last failing call ("" if none). */
const char *yf_last_error_string(void);

/* Build an engine for net-input size H x W (rows x cols, multiples of 32) from a packed-weights blob
 * (host memory; format: yolo-fastest-and-embedded-deployment_amd/packer.py -- BN-folded fp32, NHWC-friendly
 * layouts, self-describing layer table that is checked strictly against the YoloFastest graph).
 * The engine owns a device copy of the weights and nothing else.
 * The blob's header carries the io_params the reference's constructor reads (src/model_training/model/yolo_fastest.py:72-78):
 * input_channel (1 = gray, 3 = cv2's BGR frames; conv0's Cin, :78), num_anchors and num_cls (num_out = num_anchors * (5 + num_cls)
 * channels per head, :76, :138, :148).  Below, C_in = input_channel, A = num_anchors (<= 8), C = num_cls, num_out = A * (5 + C);
 * the shipped checkpoints are C_in = 1, A = 3, C = 3, num_out = 24.  yf_io_params() reads them back. */
int yf_create(const void *packed_weights, size_t nbytes, int H, int W, int max_batch, int device, yf_handle *out);
/* dtype 0 = fp32 (yf_create); 1 = BASELINE configs[2]: activations stored fp16 in HBM, the pointwise GEMMs on
 * v_mfma_f32_16x16x16_f16 with fp16 weights, fp32 accumulation everywhere; depthwise / small-channel kernels compute in fp32
 * on fp16 storage.  Input x and the two head tensors stay fp32.  (model.half() selects it from Python.) */
/* dtype 2 = fp32 storage everywhere (HBM and LDS: the residual trunk is never rounded), the pointwise GEMMs on the fp16 matrix
 * pipe with SPLIT operands: every MFMA operand a is carried as two fp16 halves hi = rne(a), lo = rne(a - hi) (22 bits) and a k-group
 * issues w_lo*a_hi + w_hi*a_lo + w_hi*a_hi into the fp32 accumulator.  As accurate as dtype 0 (same distance from the graph in fp64);
 * the accuracy answer to BASELINE configs[2]'s "2e-2 on logits", which single fp16 operands cannot meet (weights alone: 4.5e-2). */
int yf_create_ex(const void *packed_weights, size_t nbytes, int H, int W, int max_batch, int device, int dtype, yf_handle *out);
/* The fp32 -> fp16 rounding (nearest even) the weight packer uses on the host. */
uint16_t yf_f32_to_f16_bits(float f);
int yf_destroy(yf_handle h);
int yf_io_params(yf_handle h, int *input_channel, int *num_anchors, int *num_cls, int *num_out);   /* any pointer may be NULL */

/* Bytes of device scratch yf_forward / yf_detect need for a batch of N frames. */
int yf_workspace_bytes(yf_handle h, int N, size_t *out);

/* model(x): d_x float32 [N,C_in,H,W] contiguous (NCHW like the reference's input), values (u8-128)/255.
 * d_head_large float32 [N,num_out,H/16,W/16], d_head_small float32 [N,num_out,H/32,W/32], NCHW like the reference. */
int yf_forward(yf_handle h, const float *d_x, int N, float *d_head_large, float *d_head_small,
               void *d_workspace, size_t workspace_bytes, void *stream);

/* Test hook: run the forward pass and copy the named intermediate activation (the output of the
 * reference module attribute `name`, e.g. "conv1_9", "res3_3"), converted to NCHW float32
 * [N,C,h,w], into d_dst.  Returns YF_E_NOPROBE for tensors that a fused kernel keeps on chip. */
int yf_forward_probe(yf_handle h, const float *d_x, int N, const char *name, float *d_dst, size_t dst_bytes,
                     void *d_workspace, size_t workspace_bytes, void *stream);

/* Post-process of N frames (the reference handles batch element 0 only, detect.py:46; frame f here is
 * exactly what the reference computes for pred[f:f+1]).
 *   anchors: HOST double[2][A][2] = io_params["anchors"][head][anchor][w,h] in net-input pixels (the first A of each group,
 *       detect.py:51,63-64).  Classes: argmax over the C raw class logits, first maximum wins (detect.py:59).
 *   conf_thres / nms_thres: strict '>' as in detect.py:58,79.
 *   origin_h/origin_w: if both > 0 and different from H/W, corners are rescaled and re-rounded as
 *       __adjust_coord does (detect.py:131-139); pass 0,0 to keep net-input coordinates.
 * Outputs (device), fixed capacity K_max per frame, survivors in the reference's order
 * (class-major; within a class conf descending, ties in decode order):
 *   d_boxes   int32 [N,K_max,4]  x1,y1,x2,y2
 *   d_scores  float [N,K_max,2]  conf, cls_score  (computed in fp64, stored fp32)
 *   d_cls     int32 [N,K_max]
 *   d_src     int32 [N,K_max]    flat index over (head, anchor, row, col) of the surviving cell
 *   d_counts  int32 [N]          survivors of frame f; if more than K_max survive the first K_max are
 *                                stored and the count is the TRUE number (caller detects overflow);
 *                                -2 = the reference would raise ZeroDivisionError (two zero-area boxes
 *                                compared, detect.py:39)
 */
int yf_decode_nms(yf_handle h, const float *d_head_large, const float *d_head_small, int N, double conf_thres,
                  double nms_thres, const double *anchors, int origin_h, int origin_w, int K_max, int32_t *d_boxes,
                  float *d_scores, int32_t *d_cls, int32_t *d_src, int32_t *d_counts, void *stream);

/* The same with ONE output: d_records int32 [N, 1 + 8 K_max], one row per frame =
 *   count | boxes x1,y1,x2,y2 (4 K_max) | conf, cls_score as float32 bit patterns (2 K_max) | cls (K_max) | src (K_max)
 * -- the record block the multi-GPU exchange sends (one all-gather of fixed-capacity records per step): the collective
 * takes the kernel's own output buffer, nothing is packed or copied on the way.  Entries beyond `count` are not written. */
int yf_decode_nms_packed(yf_handle h, const float *d_head_large, const float *d_head_small, int N, double conf_thres, double nms_thres,
                         const double *anchors, int origin_h, int origin_w, int K_max, int32_t *d_records, void *stream);

/* YOLO_post_process.non_maxium_supression (detect.py:69-84) on its own: d_boxes int32 [n,4] is ONE class's
 * list already sorted by conf descending.  d_suppressor int32 [n] receives -1 for a kept box, else the index
 * of the kept box that removed it (-2 in every entry: the reference's ZeroDivisionError). */
int yf_nms_sorted(yf_handle h, const int32_t *d_boxes, int n, double nms_thres, int32_t *d_suppressor, void *stream);

/* Validation-time decode and NMS (the reference's OTHER convention, used by validate.py for mAP):
 *   yf_val_decode_head = YOLOLossV3.forward(input, targets=None)   src/model_training/loss/yolo_loss.py:48-68, :98-141
 *       d_head float32 [N,num_out,fh,fw] -> rows m_off .. m_off+A*fh*fw of d_out float32 [N,M_total,5+C] = (cx,cy,w,h,conf,cls0..),
 *       anchors: HOST double[A][2] of this head; calling it once per head with m_off = 0 / A*fh*fw reproduces
 *       validate.py:38-42's torch.cat over heads.  (The reference's own decode branch repeats its grid 3 times, yolo_loss.py:110-111,
 *       and therefore only runs with 3 anchors; this one takes any A.)
 *   yf_val_nms = utils.general.non_max_suppression                 src/model_training/utils/general.py:87-143 (+ bbox_iou :29-52)
 *       d_pred float32 [N,M,5+C] (yf_val_nms: C = the engine's num_cls; yf_val_nms_ex: C = num_classes, the reference's argument);
 *       conf >= conf_thres, per-class greedy NMS, IoU with the +1 convention, keep iou < nms_thres;
 *       d_det float32 [N,K_max,7] = (x1,y1,x2,y2,obj_conf,class_conf,class_pred), class-ascending then conf-descending;
 *       d_counts int32 [N] = true number of detections (0 <=> the reference's None). */
int yf_val_decode_head(yf_handle h, const float *d_head, int N, int fh, int fw, const double *anchors, int M_total, int m_off,
                       float *d_out, void *stream);
int yf_val_nms(yf_handle h, const float *d_pred, int N, int M, double conf_thres, double nms_thres, int K_max, float *d_det,
               int32_t *d_counts, void *stream);
int yf_val_nms_ex(yf_handle h, const float *d_pred, int N, int M, int num_classes, double conf_thres, double nms_thres, int K_max,
                  float *d_det, int32_t *d_counts, void *stream);

/* The loss end of the reference's training step (first slice; the layers' backward is not part of it):
 *   yf_train_loss = YOLOLossV3.forward(input, targets) for ONE head   src/model_training/loss/yolo_loss.py:48-97 (+ get_target :144-196)
 *                   and, if d_grad_head is not NULL, d(total loss)/d(input): what loss.backward() (train.py:131) leaves in input.grad.
 *   d_head float32 [N,A*(5+C),fh,fw]; anchors: HOST double[A][2] of this head (net-input pixels; yf_train_loss: A, C of the engine;
 *   yf_train_head_loss: 3, 3; yf_train_head_loss_ex: as given -- yolo_loss.py:28-33); d_targets float32 [N,T,6] =
 *   (x, y, w, h normalised to 0..1, class, marker >= 1; the first row with marker < 1 ends an image's list, yolo_loss.py:158);
 *   ignore_thres = config train_params.IOU_loss_thre.
 *   d_losses float32 [8] = total, x, y, w, h, conf, cls (the 7-tuple of yolo_loss.py:94-95) and, in [7], the number of targets whose
 *   cell lies outside the feature map (the reference raises IndexError for those; they are skipped here).
 *   d_work: yf_train_loss_workspace_bytes(N, fh, fw) bytes of device scratch, 8-byte aligned. */
int yf_train_loss_workspace_bytes(yf_handle h, int N, int fh, int fw, size_t *out);
int yf_train_loss(yf_handle h, const float *d_head, int N, int fh, int fw, const double *anchors, const float *d_targets, int T,
                  double ignore_thres, void *d_work, size_t work_bytes, float *d_losses, float *d_grad_head, void *stream);
/* The same without an engine: H x W = the net input (only the head's stride is taken from it), like the other yf_train_* entries. */
int yf_train_head_loss_workspace_bytes(int N, int fh, int fw, size_t *out);
int yf_train_head_loss(int device, int H, int W, const float *d_head, int N, int fh, int fw, const double *anchors, const float *d_targets,
                       int T, double ignore_thres, void *d_work, size_t work_bytes, float *d_losses, float *d_grad_head, void *stream);
int yf_train_head_loss_workspace_bytes_ex(int N, int fh, int fw, int num_anchors, int num_classes, size_t *out);
int yf_train_head_loss_ex(int device, int H, int W, const float *d_head, int N, int fh, int fw, const double *anchors, int num_anchors,
                          int num_classes, const float *d_targets, int T, double ignore_thres, void *d_work, size_t work_bytes,
                          float *d_losses, float *d_grad_head, void *stream);

/* Operators of the reference's TRAINING step (second slice): every layer type of YoloFastest in train mode, forward
 * and backward, on NCHW float32 device tensors like the reference's (src/model_training/model/yolo_fastest.py:16-66, train.py:98-160).
 * Per-layer kernels (MFMA GEMMs for the pointwise / dense convs, split reductions added in a fixed order: DESIGN.md section 4) -- not the
 * tuned inference engine, which folds BatchNorm and therefore cannot train.  `device` = HIP device index; everything is stream-ordered.
 *   conv: Conv2d(k in {1,3,5}, stride in {1,2}, pad (k-1)/2, groups 1 or C [depthwise = 1]); d_w [Cout, Cin/groups, k, k]; d_bias or NULL.
 *   deconv: ConvTranspose2d(k 2, stride 2, pad 0); d_w [Cin, Cout, 2, 2]; output 2H x 2W.
 *   bn: BatchNorm2d in train mode (eps 1e-5, momentum 0.1): d_stats float[2C] receives (mean, invstd) per channel for the backward;
 *       running_mean / running_var are updated in place (unbiased variance) unless NULL; relu = 1 fuses the following ReLU (the backward
 *       then masks by y > 0, with y recomputed from x bit for bit -- it does not read the forward's output).  bn_backward: d_x = the
 *       forward's input, d_dgamma, d_dbeta [C], d_dx like x.  C <= 256.
 *   d_scratch: yf_train_scratch_bytes() of device memory, one per stream, no initialisation needed: the per-channel reductions of bn
 *       and the weight gradients are split over many workgroups that store partial results there, and a second pass adds them in a
 *       fixed order (deterministic; device-scope float atomics are slow on a multi-XCD part).  conv_backward_weight accepts NULL / a
 *       smaller buffer (scratch_bytes) and then splits less.
 *   adam_step: torch.optim.Adam(lr, betas, eps), no weight decay (train.py:84); step counts from 1. */
int yf_train_conv_forward(int device, const float *d_x, const float *d_w, const float *d_bias, float *d_y, int N, int Cin, int H, int W, int Cout,
                          int k, int stride, int depthwise, void *stream);
int yf_train_conv_backward_data(int device, const float *d_dy, const float *d_w, float *d_dx, int N, int Cin, int H, int W, int Cout, int k,
                                int stride, int depthwise, void *stream);
int yf_train_conv_backward_weight(int device, const float *d_x, const float *d_dy, float *d_dw, int N, int Cin, int H, int W, int Cout, int k,
                                  int stride, int depthwise, void *d_scratch, size_t scratch_bytes, void *stream);
int yf_train_deconv_forward(int device, const float *d_x, const float *d_w, float *d_y, int N, int Cin, int H, int W, int Cout, void *stream);
int yf_train_deconv_backward_data(int device, const float *d_dy, const float *d_w, float *d_dx, int N, int Cin, int H, int W, int Cout, void *stream);
int yf_train_deconv_backward_weight(int device, const float *d_x, const float *d_dy, float *d_dw, int N, int Cin, int H, int W, int Cout,
                                    void *d_scratch, size_t scratch_bytes, void *stream);
int yf_train_scratch_bytes(size_t *bytes);
int yf_train_bn_forward(int device, const float *d_x, const float *d_gamma, const float *d_beta, float *d_running_mean, float *d_running_var,
                        float *d_stats, float *d_y, int N, int C, long HW, int relu, void *d_scratch, void *stream);
int yf_train_bn_backward(int device, const float *d_x, const float *d_dy, const float *d_stats, const float *d_gamma, const float *d_beta,
                         float *d_dgamma, float *d_dbeta, float *d_dx, int N, int C, long HW, int relu, void *d_scratch, void *stream);
/* One block of the network each way -- conv_norm_relu / conv_norm / deconv_norm_relu (yolo_fastest.py:16-48): conv (deconv = 1:
 * ConvTranspose2d) + BatchNorm (+ ReLU).  forward: d_z = conv output (kept for the backward), d_y = block output.  backward: d_gy =
 * gradient of the block output -> d_dgamma, d_dbeta, d_dw, d_dx (NULL: not needed, the first layer); d_gz: work tensor like z. */
int yf_train_unit_forward(int device, int deconv, const float *d_x, const float *d_w, const float *d_gamma, const float *d_beta,
                          float *d_running_mean, float *d_running_var, float *d_stats, float *d_z, float *d_y, int N, int Cin, int H, int W, int Cout,
                          int k, int stride, int depthwise, int relu, void *d_scratch, void *stream);
int yf_train_unit_backward(int device, int deconv, const float *d_x, const float *d_z, const float *d_gy, const float *d_stats, const float *d_w,
                           const float *d_gamma, const float *d_beta, float *d_dgamma, float *d_dbeta, float *d_gz, float *d_dw, float *d_dx, int N,
                           int Cin, int H, int W, int Cout, int k, int stride, int depthwise, int relu, void *d_scratch, size_t scratch_bytes,
                           void *stream);
int yf_train_channel_sum(int device, const float *d_dy, float *d_out, int N, int C, long HW, void *stream);                 /* bias gradient */
/* the same sum spread over the chip for large N HW (partial sums in d_scratch, yf_train_scratch_bytes(); a second launch adds them in order) */
int yf_train_channel_sum_split(int device, const float *d_dy, float *d_out, int N, int C, long HW, void *d_scratch, void *stream);
int yf_train_add(int device, const float *d_a, const float *d_b, float *d_out, long total, void *stream);                   /* residual / grad sum */
int yf_train_channel_slice(int device, const float *d_src, float *d_dst, int N, int C, long HW, int Cs, int sc0, int Cd, int dc0,
                           void *stream);                                                                                    /* torch.cat and back */
int yf_train_adam_step(int device, float *d_p, const float *d_g, float *d_m, float *d_v, long total, double lr, double beta1, double beta2,
                       double eps, int step, void *stream);
/* the same update for `ntensors` tensors in one launch: HOST arrays of device pointers and sizes; d_table: >= 48 * ntensors bytes of
 * device scratch (the pointer table is uploaded into it on `stream`). */
int yf_train_adam_multi(int device, int ntensors, void *const *d_p, const void *const *d_g, void *const *d_m, void *const *d_v, const long *sizes,
                        double lr, double beta1, double beta2, double eps, int step, void *d_table, size_t table_bytes, void *stream);
/* ... with a caller-owned PINNED host copy of the table (>= 48 * ntensors bytes, one per optimizer): the table is written there and
 * uploaded asynchronously, and only when `upload` != 0 -- set it when the pointer set changed; before the NEXT call with upload != 0 the
 * previous upload must have executed (wait on an event recorded behind this call).  h_table_pinned NULL = yf_train_adam_multi. */
int yf_train_adam_multi_pinned(int device, int ntensors, void *const *d_p, const void *const *d_g, void *const *d_m, void *const *d_v,
                               const long *sizes, double lr, double beta1, double beta2, double eps, int step, void *d_table, size_t table_bytes,
                               void *h_table_pinned, int upload, void *stream);

/* The same training forward / backward as ONE call each: the whole graph of yolo_fastest.py:150-218 in train mode (what
 * `pred = model(imgs)` and `loss.backward()` run, train.py:114, :131), launched from C++ into a caller-owned workspace.
 *   d_params / d_grads: HOST arrays of yf_trainer_num_params() device pointers in `model.parameters()` order (per conv block: conv
 *     weight, BN weight, BN bias; per head: weight, bias); d_grads are written, not accumulated.
 *   d_bn_buffers: HOST array of 2 * n_bn device pointers (running_mean, running_var per BatchNorm in module order) or NULL.
 *   d_ws: yf_trainer_workspace_bytes(t, N) bytes; forward leaves the activations there and backward reads them: the same buffer,
 *     untouched in between.  backward also needs the images again (d_x: the first conv's weight gradient).
 * Stream-ordered, no allocation, no synchronisation. */
typedef struct yf_trainer_s *yf_trainer;
int yf_trainer_create(int H, int W, int device, yf_trainer *out);                                       /* input_channel 1, num_out 24 */
int yf_trainer_create_ex(int H, int W, int device, int input_channel, int num_out, yf_trainer *out);   /* yolo_fastest.py:72-78 */
void yf_trainer_destroy(yf_trainer t);
int yf_trainer_num_params(yf_trainer t, int *n_params, int *n_bn);
int yf_trainer_workspace_bytes(yf_trainer t, int N, size_t *bytes);
int yf_trainer_forward(yf_trainer t, const float *d_x, int N, const void *const *d_params, void *const *d_bn_buffers, float *d_head_large,
                       float *d_head_small, void *d_ws, size_t ws_bytes, void *stream);
int yf_trainer_backward(yf_trainer t, const float *d_x, const float *d_grad_head_large, const float *d_grad_head_small, int N,
                        const void *const *d_params, void *const *d_grads, void *d_ws, size_t ws_bytes, void *stream);
/* how many passes ran as a HIP graph replay: a pass whose pointer arguments equal those of the call before it is captured once and
   replayed from then on (the steady state of a training loop); YF_TRAIN_GRAPH_OFF=1 in the environment disables it */
int yf_trainer_graph_replays(yf_trainer t, long *forward, long *backward);
/* out6 = replays (forward, backward), captures (forward, backward), evictions (forward, backward).  A trainer keeps up to four graphs per
   pass (one per remembered pointer set); a pointer pattern that keeps evicting graphs before they were replayed stops capturing. */
int yf_trainer_graph_stats(yf_trainer t, long *out6);
int yf_trainer_set_graphs(yf_trainer t, int on);   /* 0: this trainer issues every pass as plain launches (the environment variable YF_TRAIN_GRAPH_OFF
                                                      does the same for the whole process); default 1 */

/* yf_forward + yf_decode_nms back to back on one stream (heads also returned; may be NULL to use
 * workspace-internal buffers). */
int yf_detect(yf_handle h, const float *d_x, int N, double conf_thres, double nms_thres, const double *anchors,
              int origin_h, int origin_w, int K_max, int32_t *d_boxes, float *d_scores, int32_t *d_cls,
              int32_t *d_src, int32_t *d_counts, float *d_head_large, float *d_head_small, void *d_workspace,
              size_t workspace_bytes, void *stream);

/* yf_detect with the packed record block of yf_decode_nms_packed as its output. */
int yf_detect_packed(yf_handle h, const float *d_x, int N, double conf_thres, double nms_thres, const double *anchors, int origin_h,
                     int origin_w, int K_max, int32_t *d_records, float *d_head_large, float *d_head_small, void *d_workspace,
                     size_t workspace_bytes, void *stream);

/* Detect_YOLO.__pre_process arithmetic on device: d_u8 uint8 [N,src_h,src_w] gray frames ->
 * d_x float32 [N,1,H,W] = (v-128)/255 where v is the pixel itself (src == net size) or the 2x2
 * box mean (a+b+c+d+2)>>2 (src == 2x net size).  Other ratios: YF_E_INVALID.
 * C_in = 3: d_u8 uint8 [N,src_h,src_w,3] as cv2.imread returns a frame (HWC, BGR) -> d_x float32 [N,3,H,W] with the channel order
 * reversed, detect.py:119 `img[:, :, ::-1].transpose(2, 0, 1)`; the box mean is taken per channel. */
int yf_preprocess_u8(yf_handle h, const uint8_t *d_u8, int N, int src_h, int src_w, float *d_x, void *stream);

/* yf_forward on u8 frames: Detect_YOLO.__pre_process's arithmetic (src/detect.py:115-124) in front of the net.  d_u8 uint8 [N,src_h,src_w]
 * (C_in = 3: [N,src_h,src_w,3] HWC BGR) of ANY size (detect.py:115-116: `cv2.resize(img, (input_shape[1], input_shape[0]))`):
 *   src == net size, or exactly 2x (cv::resize turns INTER_LINEAR into INTER_AREA there: the 2x2 box mean (a+b+c+d+2)>>2): fused into the
 *     first kernel's loads -- bit-identical to yf_preprocess_u8 followed by yf_forward, one pass and 3-15 bytes per pixel less HBM traffic;
 *   any other size: one extra pass (yf_cv_preprocess_u8: cv::resize's 8-bit INTER_LINEAR) into net-sized u8 frames at the end of the
 *     workspace, then the fused entry.  The first call for a new source size builds cv::resize's coefficient tables with one small
 *     kernel on `stream` (round 6: no allocation or host synchronisation on this path -- the first resize of an engine allocates the table
 *     pool once; 32 distinct source sizes are kept, a 33rd evicts behind a device synchronisation). */
int yf_forward_u8(yf_handle h, const uint8_t *d_u8, int N, int src_h, int src_w, float *d_head_large, float *d_head_small,
                  void *d_workspace, size_t workspace_bytes, void *stream);

/* The two OpenCV calls of Detect_YOLO.__pre_process (src/detect.py:110-116) on the device, for any source size:
 *     img = cv2.cvtColor(ori_img, cv2.COLOR_BGR2GRAY)        1-channel net, src_c == 3 (cv2.imread's BGR frames)
 *     img = cv2.resize(img, (W, H))                           INTER_LINEAR; exactly 1/2 -> the 2x2 mean; same size -> copy
 * d_src uint8 [N,src_h,src_w(,3)] -> d_dst uint8 [N,H,W(,C_in)] (a 3-channel net keeps BGR order: detect.py:119's reversal is the next step).
 * OpenCV's published 8-bit arithmetic is restated: gray = (B*BY + G*GY + R*RY + half) >> shift with gray_bits 15 (9798/19235/3735: OpenCV 4.x;
 * 0 means 15) or 14 (4899/9617/1868: OpenCV 2.x / 3.x); resize with 11-bit coefficients, dst = (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
 * Bit-exact against oracle/cv_oracle.py; parity with an actual OpenCV build is UNPINNED (cv2 is not available where this was built, and an
 * IPP / vendor-HAL build may round differently). */
int yf_cv_preprocess_u8(yf_handle h, const uint8_t *d_src, int N, int src_h, int src_w, int src_c, int gray_bits, uint8_t *d_dst, void *stream);

/* yf_forward on cv2.imread's frames (uint8 [N,src_h,src_w,3], BGR) of any size: yf_cv_preprocess_u8 + the fused (v-128)/255 entry.  What
 * `Detect_YOLO.batch_detect` does per image (detect.py:108-127, 152), batched.  A 3-channel net: identical to yf_forward_u8. */
int yf_forward_bgr_u8(yf_handle h, const uint8_t *d_bgr, int N, int src_h, int src_w, int gray_bits, float *d_head_large, float *d_head_small,
                      void *d_workspace, size_t workspace_bytes, void *stream);

/* Introspection used by tests / bench. */
/* Name ("conv1_8+conv1_9+conv2_1"), layer-granular algorithmic bytes and flops per frame of launch `op` of the
 * current plan (each conv of the op reads its input and writes its output once, + residual read: SURVEY.md 8d). */
int yf_op_info(yf_handle h, int op, char *name, int name_len, double *algorithmic_bytes_per_frame, double *flops_per_frame);
/* The same with the flops split by the pipe they run on in this plan: matrix cores (pointwise / dense / deconv layers of the
 * MFMA kernels) and vector ALU (depthwise convs, the small-channel VALU block kernels).  bench.py prices each against its peak. */
int yf_op_info_ex(yf_handle h, int op, char *name, int name_len, double *algorithmic_bytes_per_frame, double *mfma_flops_per_frame,
                  double *valu_flops_per_frame);
/* Arithmetic the kernel of launch `op` runs in: 0 fp32, 1 fp16 storage + fp16 MFMA, 2 fp32 storage + split-operand fp16 MFMA (a
 * dtype-2 engine runs the launches that have no split-operand kernel in exact fp32: same storage, same accuracy class). */
int yf_op_dtype(yf_handle h, int op, int *kernel_dtype);
/* Kernel dispatches launch `op` issues at batch N: 1, except a chained residual launch at small batches, which is issued block by block
 * (DESIGN.md section 4 "Small batches").  Counter tools that match rocprofv3's dispatch rows to launches by order need it. */
int yf_op_dispatches(yf_handle h, int op, int N, int *dispatches);
/* One forward pass (whole batch in one pass) with a HIP event recorded on `stream` around every launch; blocks until
 * the pass is done and returns each launch's duration in ms in op_ms[yf_num_launches]. */
int yf_profile_forward(yf_handle h, const float *d_x, int N, void *d_workspace, size_t workspace_bytes, void *stream,
                       float *op_ms, int n_ops);
/* the same pass from u8 frames (yf_forward_u8's input conventions): the first launch then includes the fused pre-process */
int yf_profile_forward_u8(yf_handle h, const uint8_t *d_u8, int N, int src_h, int src_w, void *d_workspace, size_t workspace_bytes,
                          void *stream, float *op_ms, int n_ops);
int yf_streams_overlap(yf_handle h, void *stream_a, void *stream_b, int *overlap);
                                                   /* do two HIP streams run concurrently?  The runtime multiplexes streams onto a few
                                                      hardware queues; two on one queue execute strictly in issue order (measured: -30 %
                                                      for two lanes, no gain from two batches in flight).  Host-blocking probe (~0.2 ms:
                                                      a 100 us spin kernel on each); the engine uses it to pick its lane / branch streams
                                                      per caller stream, BatchPipeline to pick the streams of the batches in flight */
int yf_set_profile_repeats(yf_handle h, int repeats);
                                                   /* yf_profile_forward[_u8]: every launch is issued `repeats` (1 .. 64, default 1) times back to
                                                      back between its two events and the elapsed time divided by it -- an event pair around ONE
                                                      launch also times the event packets and the dispatch gap (5-7 us per launch, host dependent:
                                                      rocprofv3's kernel durations are that much shorter); every launch writes its whole output
                                                      from inputs it does not modify, so repeating it changes nothing */
int yf_profile_head_offsets(yf_handle h, int N, size_t *large_off, size_t *small_off);
                                                   /* byte offsets inside the workspace handed to yf_profile_forward[_u8] at which that pass left its
                                                      head logits (NCHW fp32, [N, num_out, H/16, W/16] and [N, num_out, H/32, W/32]): tests hold the
                                                      profiled (repeated-launch) pass itself to yf_forward's bits */
int yf_num_launches(yf_handle h, int *out);       /* kernel launches one yf_forward issues            */
int yf_set_chunk(yf_handle h, int frames);        /* frames per pass of the layer chain (0 = whole batch) */
int yf_set_split_sums(yf_handle h, int on);       /* 1 (default): at a handful of frames (<= 9, fp32 engines, 320x256 nets) the stride-32 residual chain and the
                                                      small head split their channel sums over several workgroups and add the partial sums in chunk order at
                                                      launch boundaries (batch-1 latency: DESIGN.md section 4 "Small batches").  Since round 6 the large-batch
                                                      kernels form the same per-chunk partial sums in the same order: THE SAME BITS either way, a frame's logits
                                                      never depend on how many frames travel with it.  0: the one-workgroup launches at every batch size. */
int yf_set_post_split(yf_handle h, int mode);      /* decode + NMS as one workgroup per frame AND class instead of one per frame (the reference runs NMS per class,
                                                      detect.py:158-169: independent work; a second small launch concatenates the classes in class order) --
                                                      the same records bit for bit, for DENSE frames: 64 frames of 1200 candidates use 192 CUs instead of 64
                                                      (0.127 -> 0.058 ms).  0 (default): automatically when the caller reserves K_max >= 256 survivors per frame,
                                                      the model has <= 8 classes and 2 N <= #CU; 1: always (<= 64 classes); 2: never.  The first dense call of a
                                                      size allocates the engine's scratch (not inside a stream capture). */
int yf_set_lanes(yf_handle h, int lanes);          /* 1..4: chunks of the batch (yf_set_chunk) run on this many concurrent
                                                     streams, forked from / joined to the caller's stream by events    */
int yf_set_branches(yf_handle h, int on);         /* 1 (default): the small head's launches (conv5_3 .. head_5) run on a side stream of
                                                     their lane, beside the large head's (deconv5_1 .. head_4); 0 = in line */
int yf_set_fusion(yf_handle h, int level);        /* 2 (default) = block-fused kernels + the per-frame deep stage's launch boundaries
                                                     removed (conv5_2 in the res5 launch, the small head one launch, deconv5_1 +
                                                     conv4_1_1 one launch); 1 = block-fused kernels (bitwise the same heads);
                                                     0 = one launch per layer, every named tensor probe-able (bring-up / parity) */

#ifdef __cplusplus
}
#endif
#endif
