"""ORACLE (test infrastructure, not product): CPU restatement of the reference post-process.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

Restates /root/reference/src/detect.py in plain Python doubles (the reference's own numeric
type there: fp32 logits promoted to Python float, math.exp, round() = round-half-even):
    YOLO_post_process.__sigmoid              :23-25
    YOLO_post_process.__cal_iou              :27-39
    YOLO_post_process.decode_box             :41-67
    YOLO_post_process.non_maxium_supression  :69-84
    Detect_YOLO.batch_detect glue            :157-169  (bucket by class, stable sort by conf
                                                        descending, NMS per class, class-major concat)
    Detect_YOLO.__adjust_coord               :131-139

Each candidate additionally carries its flat source index over the (head, anchor, row, col)
enumeration -- element [7] -- so that survivor ORDER can be compared bit-exactly.

Parity pin: tests/test_oracle_golden.py checks this file against the goldens that
tests/golden/make_golden.py produced by exec'ing the reference's own YOLO_post_process class
(20 + 6 real frames, 8 dense synthetic frames with up to 1228 candidates).
oracle/post_oracle.c is the same algorithm in C (for batch sizes Python loops cannot finish).
"""
import math


def sigmoid(x):  # detect.py:23-25
    return 1. / (1. + math.exp(-x))


def cal_iou(b1, b2):  # detect.py:27-39 -- integer corners, no +1; ZeroDivisionError if both areas are 0
    inter = 0
    iw = min(b1[2], b2[2]) - max(b1[0], b2[0])
    ih = min(b1[3], b2[3]) - max(b1[1], b2[1])
    if iw > 0 and ih > 0:
        inter = ih * iw
    union = (b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter
    return inter / union


def decode_box(heads, anchors, input_shape, conf_thres, num_anchors=3, num_cls=3):
    """detect.py:41-67.  heads: sequence of numpy float32 arrays [A*(5+C), h, w] for ONE frame
    (the reference takes batch element 0, :46).  Returns [[x1,y1,x2,y2,conf,cls_score,cls,src], ...]
    in (head, anchor, row, col) order."""
    out, base = [], 0
    attrs = 5 + num_cls
    for head, p in enumerate(heads):
        in_h, in_w = p.shape[1], p.shape[2]
        scale_h = input_shape[0] / in_h
        scale_w = input_shape[1] / in_w
        a = anchors[head]
        p = p.reshape(num_anchors, attrs, in_h, in_w)
        for pp in range(num_anchors):
            for i in range(in_h):
                for j in range(in_w):
                    conf = sigmoid(p[pp, 4, i, j])
                    if conf > conf_thres:
                        best, cls = p[pp, 5, i, j], 0
                        for c in range(1, num_cls):  # np.argmax: first maximum wins
                            if p[pp, 5 + c, i, j] > best:
                                best, cls = p[pp, 5 + c, i, j], c
                        cls_score = sigmoid(best)
                        x = (j + sigmoid(p[pp, 0, i, j])) * scale_w
                        y = (i + sigmoid(p[pp, 1, i, j])) * scale_h
                        w = math.exp(p[pp, 2, i, j]) * a[pp][0]
                        h = math.exp(p[pp, 3, i, j]) * a[pp][1]
                        out.append([round(x - w / 2), round(y - h / 2), round(x + w / 2), round(y + h / 2),
                                    conf, cls_score, cls, base + (pp * in_h + i) * in_w + j])
        base += num_anchors * in_h * in_w
    return out


def nms(bbox_list, nms_thres):  # detect.py:69-84 (consumes its argument, like the reference)
    results = []
    while len(bbox_list) != 0:
        results.append(bbox_list[0])
        if len(bbox_list) == 1:
            break
        bbox_list.pop(0)
        i = 0
        while i <= len(bbox_list) - 1:
            if cal_iou(bbox_list[i], results[-1]) > nms_thres:
                bbox_list.pop(i)
            else:
                i += 1
    return results


def detect_glue(cands, nms_thres, num_cls=3):  # detect.py:158-169
    buckets = [[] for _ in range(num_cls)]
    for b in cands:
        buckets[b[6]].append(b)
    out = []
    for cls in range(num_cls):
        if len(buckets[cls]) == 0:
            continue
        buckets[cls].sort(key=lambda it: it[4], reverse=True)  # stable
        out.extend(nms(buckets[cls], nms_thres))
    return out


def adjust_coord(boxes, origin_shape, input_shape):  # detect.py:131-139
    sh = origin_shape[0] / input_shape[0]
    sw = origin_shape[1] / input_shape[1]
    for b in boxes:
        b[0] = round(b[0] * sw)
        b[2] = round(b[2] * sw)
        b[1] = round(b[1] * sh)
        b[3] = round(b[3] * sh)
    return boxes


def post_process(heads, anchors, input_shape, conf_thres=0.5, nms_thres=0.2, num_cls=3, num_anchors=3):
    """decode + per-class NMS for one frame; returns the class-major survivor list."""
    return detect_glue(decode_box(heads, anchors, input_shape, conf_thres, num_anchors, num_cls), nms_thres, num_cls)
