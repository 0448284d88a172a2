/* ORACLE (test infrastructure, not product): C restatement of the reference post-process.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is the same algorithm as oracle/post_oracle.py, in C so that batches of dense frames
 * (thousands of candidates, O(n^2) IoU evaluations) finish in seconds.  Follows
 *   /root/reference/src/detect.py
 *     __sigmoid :23-25, __cal_iou :27-39, decode_box :41-67, non_maxium_supression :69-84,
 *     batch_detect glue :158-169 (bucket by class, STABLE sort by conf descending, class-major concat).
 * Numeric type: the reference promotes fp32 logits to Python floats (C double), uses math.exp
 * (this libm's exp) and round() (round-half-even == rint() in the default rounding mode).
 * The greedy list-pop NMS of :69-84 is restated with suppression flags, which visits the same
 * (kept, candidate) pairs in the same order.
 *
 * Parity pin: tests/test_oracle_golden.py checks it against oracle/post_oracle.py and the
 * goldens produced by the reference's own class (tests/golden/make_golden.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static double sigmoid_d(double x) { return 1. / (1. + exp(-x)); }

typedef struct {
    long long x1, y1, x2, y2;
    double conf, score;
    int cls, src;
} cand_t;

/* returns number of survivors written (<= kmax), -1 if kmax too small, -2 for the reference's
 * ZeroDivisionError (two zero-area boxes compared, detect.py:39). */
int yf_oracle_post(const float *head_large, int hl, int wl, const float *head_small, int hs, int ws,
                   const double *anchors /*[2][num_anchors][2]*/, int in_h, int in_w, double conf_thres, double nms_thres,
                   int num_cls, int num_anchors, int kmax, int32_t *out_box /*[kmax][4]*/, double *out_conf, double *out_score,
                   int32_t *out_cls, int32_t *out_src, int32_t *n_candidates)
{
    const int A = num_anchors, attrs = 5 + num_cls;
    int total = A * (hl * wl + hs * ws);
    cand_t *c = (cand_t *)malloc(sizeof(cand_t) * (size_t)total);
    int n = 0, base = 0;
    for (int head = 0; head < 2; ++head) {
        const float *p = head ? head_small : head_large;
        int h = head ? hs : hl, w = head ? ws : wl;
        double scale_h = (double)in_h / h, scale_w = (double)in_w / w;
        for (int pp = 0; pp < A; ++pp)
            for (int i = 0; i < h; ++i)
                for (int j = 0; j < w; ++j) {
#define T(k) ((double)p[(((size_t)pp * attrs + (k)) * h + i) * w + j])
                    double conf = sigmoid_d(T(4));
                    if (!(conf > conf_thres)) continue;
                    int cls = 0;
                    float best = p[(((size_t)pp * attrs + 5) * h + i) * w + j];
                    for (int k = 1; k < num_cls; ++k) {
                        float v = p[(((size_t)pp * attrs + 5 + k) * h + i) * w + j];
                        if (v > best) { best = v; cls = k; }
                    }
                    double x = (j + sigmoid_d(T(0))) * scale_w;
                    double y = (i + sigmoid_d(T(1))) * scale_h;
                    double bw = exp(T(2)) * anchors[(head * A + pp) * 2 + 0];
                    double bh = exp(T(3)) * anchors[(head * A + pp) * 2 + 1];
#undef T
                    cand_t *q = &c[n++];
                    q->x1 = (long long)rint(x - bw / 2); q->y1 = (long long)rint(y - bh / 2);
                    q->x2 = (long long)rint(x + bw / 2); q->y2 = (long long)rint(y + bh / 2);
                    q->conf = conf; q->score = sigmoid_d((double)best); q->cls = cls;
                    q->src = base + (pp * h + i) * w + j;
                }
        base += A * h * w;
    }
    if (n_candidates) *n_candidates = n;
    int nout = 0, rc = 0;
    int *idx = (int *)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    char *dead = (char *)malloc((size_t)(n > 0 ? n : 1));
    for (int cls = 0; cls < num_cls && rc == 0; ++cls) {
        int m = 0;
        for (int k = 0; k < n; ++k) if (c[k].cls == cls) idx[m++] = k;
        /* stable insertion sort, conf descending (list.sort(reverse=True) keeps ties in order) */
        for (int a = 1; a < m; ++a) {
            int v = idx[a], b = a - 1;
            while (b >= 0 && c[idx[b]].conf < c[v].conf) { idx[b + 1] = idx[b]; --b; }
            idx[b + 1] = v;
        }
        memset(dead, 0, (size_t)(m > 0 ? m : 1));
        for (int a = 0; a < m && rc == 0; ++a) {
            if (dead[a]) continue;
            const cand_t *ka = &c[idx[a]];
            if (nout >= kmax) { rc = -1; break; }
            out_box[nout * 4 + 0] = (int32_t)ka->x1; out_box[nout * 4 + 1] = (int32_t)ka->y1;
            out_box[nout * 4 + 2] = (int32_t)ka->x2; out_box[nout * 4 + 3] = (int32_t)ka->y2;
            out_conf[nout] = ka->conf; out_score[nout] = ka->score; out_cls[nout] = cls; out_src[nout] = ka->src;
            ++nout;
            for (int b = a + 1; b < m; ++b) {
                if (dead[b]) continue;
                const cand_t *kb = &c[idx[b]];
                long long iw = (kb->x2 < ka->x2 ? kb->x2 : ka->x2) - (kb->x1 > ka->x1 ? kb->x1 : ka->x1);
                long long ih = (kb->y2 < ka->y2 ? kb->y2 : ka->y2) - (kb->y1 > ka->y1 ? kb->y1 : ka->y1);
                long long inter = (iw > 0 && ih > 0) ? iw * ih : 0;
                long long uni = (kb->x2 - kb->x1) * (kb->y2 - kb->y1) + (ka->x2 - ka->x1) * (ka->y2 - ka->y1) - inter;
                if (uni == 0) { rc = -2; break; }
                if ((double)inter / (double)uni > nms_thres) dead[b] = 1;
            }
        }
    }
    free(c); free(idx); free(dead);
    return rc < 0 ? rc : nout;
}
