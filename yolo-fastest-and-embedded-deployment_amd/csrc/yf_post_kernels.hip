// yf_post_kernels.hip -- anchor/grid decode + per-class greedy NMS for gfx950, one workgroup per frame.
//
// Replaces the pure-Python post-process of the reference, src/detect.py:
//   YOLO_post_process.decode_box            :41-67   (triple loop anchors x rows x cols, Python doubles)
//   class bucketing + stable sort by conf   :158-167
//   YOLO_post_process.non_maxium_supression :69-84   (+ __cal_iou :27-39, integer corners, no +1)
//   Detect_YOLO.__adjust_coord              :131-139 (optional epilogue)
//
// Exactness plan (SURVEY.md "Hard parts"):
//   * threshold: the host finds, with this machine's libm, the smallest fp32 logit t for which the
//     reference's  1/(1+exp(-t)) > conf_thres  holds (sigmoid is monotone), so the device compares raw
//     fp32 logits -- bit-exact for every conf_thres;
//   * order: candidates are sorted ONCE on a unique 64-bit key (class asc | conf desc -- conf_order() -- | cell asc):
//     class-major output, conf descending, ties in decode order == the reference's bucket + stable sort
//     (sigmoid is monotone in the logit; two different logits share one double conf only above
//     logit ~ 22, conf > 1-3e-10, and there the key is built from the conf's own fp64 value: conf_order());
//   * boxes: fp64 arithmetic and rint() (round-half-even == Python round());
//   * IoU test: int64 areas, fp64 IEEE division, strict '>'.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "yf_kernels.h"

namespace yf {

static constexpr int POST_THREADS = 1024;  // one workgroup per frame owns the CU (LDS): 16 waves hide the LDS / fp64 latencies of the
                                           // sort, decode and NMS loops (with 4 waves the dense frames took 3x longer)
static constexpr int POST_PARTS = POST_THREADS / 64, POST_CPT = 64 / POST_PARTS;  // window row i = tid / PARTS tests CPT candidates

__device__ inline uint32_t orderable(float f)
{
    uint32_t u = __float_as_uint(f);
    return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ inline double sigmoid_d(double x) { return 1. / (1. + exp(-x)); }
// 32-bit sort field of a conf logit: ascending field == descending conf, and EQUAL fields exactly where the reference's fp64
// conf = 1 / (1 + exp(-t)) (detect.py:23-25, :58) is equal, so that its stable sort's ties (decode order, :167) are ties here too.
//   * t <= 20: distinct fp32 logits give distinct fp64 confs (an fp32 step moves conf by >= 17 ulp there) -> the logit's own order;
//   * t > 20: 1 + exp(-t) rounds to 1 + k 2^-52 with k = RN(exp(-t) 2^52) < 2^24, and 1 / (1 + k 2^-52) rounds to 1 - k 2^-52 exactly
//     (the next term, k^2 2^-104 < 5e-18, is below a quarter of the spacing of doubles under 1): conf is a function of k alone.
//     Different logits share one conf above t ~ 22 (all of them above 36.7, where conf == 1.0); the field is k.  A device exp() that is
//     one ulp off the host's changes k only if exp(-t) 2^52 lies within k 2^-52 (relative) of a half-integer: < 2e-9 per candidate.
// The two ranges join monotonically: k(20+) = 9.28e6 < ~orderable(20.0f) = 0x3e5fffff.
// (Not covered: logits within 4e-9 of zero also share conf == 0.5 +- 1 ulp in fp64; they only pass thresholds below 0.5.)
__device__ inline uint32_t conf_order(float t)
{
    if (t > 20.f) return (uint32_t)rint(exp(-(double)t) * 4503599627370496.0);
    return ~orderable(t);
}
__device__ inline int32_t clamp_i32(double v)
{
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int32_t)(-2147483647 - 1);
    return (int32_t)v;
}

struct CellRef {
    const float* p;  // frame's head tensor
    int h, w, head, pp, i, j;
    int attrs;       // 5 + num_cls
};
__device__ inline CellRef locate(const PostArgs& a, long frame, int cell)
{
    CellRef r;
    const int nl = a.na * a.hl * a.wl;
    const long nout = (long)a.na * (5 + a.nc);   // channels of a head tensor: num_anchors * (5 + num_cls), detect.py:53
    if (cell < nl) {
        r.head = 0; r.h = a.hl; r.w = a.wl; r.p = a.head_large + frame * nout * a.hl * a.wl;
    } else {
        cell -= nl; r.head = 1; r.h = a.hs; r.w = a.ws; r.p = a.head_small + frame * nout * a.hs * a.ws;
    }
    r.j = cell % r.w;
    int t = cell / r.w;
    r.i = t % r.h;
    r.pp = t / r.h;
    r.attrs = 5 + a.nc;
    return r;
}
__device__ inline long s_area(const int4& b) { return ((long)b.z - b.x) * ((long)b.w - b.y); }
__device__ inline float logit_at(const CellRef& r, int k) { return r.p[((r.pp * r.attrs + k) * r.h + r.i) * r.w + r.j]; }

// Shared scalars of one workgroup's post-process
struct PostShared {
    int wave_cnt[POST_THREADS / 64];
    int total, nkept, err, zero_area, wlen, wk;
    uint16_t pm[64][POST_PARTS], pz[64][POST_PARTS];
    int4 wb[64];   // the current window's survivors: boxes and areas
    long wa[64];
};

// ---- phase 1: threshold + order-preserving compaction of candidate keys (decode order).  only_cls >= 0: the candidates of that class only
// (post_split_kernel: one workgroup per frame and class).  A thread owns CPT CONSECUTIVE cells, so one block-wide prefix sum (wave scan +
// 16 wave totals: two barriers) orders everything -- rounds of 1024 cells with three barriers each before round 6.
constexpr int POST_MAX_CPT = 8;   // ncell <= 8191 = 8 x 1024 - 1
__device__ __forceinline__ int compact_keys(const PostArgs& a, long frame, int ncell, uint64_t* keys, int only_cls, PostShared& sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cpt = (ncell + POST_THREADS - 1) / POST_THREADS;
    uint64_t mine[POST_MAX_CPT];
    unsigned mask = 0;
#pragma unroll
    for (int u = 0; u < POST_MAX_CPT; ++u) {
        mine[u] = 0;
        const int cell = tid * cpt + u;
        if (u < cpt && cell < ncell) {
            CellRef r = locate(a, frame, cell);
            float t4 = logit_at(r, 4);
            if (t4 >= a.logit_min) {  // NaN fails, like `nan > thres`
                if (t4 == 0.f) t4 = 0.f;  // -0.0 and +0.0 have the same conf
                float best = logit_at(r, 5);
                int cls = 0;
                for (int k = 1; k < a.nc; ++k) {
                    float v = logit_at(r, 5 + k);
                    if (v > best) { best = v; cls = k; }  // np.argmax: first maximum wins (detect.py:59)
                }
                if (only_cls < 0 || cls == only_cls) {
                    mine[u] = ((uint64_t)cls << 45) | ((uint64_t)conf_order(t4) << 13) | (uint64_t)cell;
                    mask |= 1u << u;
                }
            }
        }
    }
    const int c = __popc(mask);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) sh.wave_cnt[wave] = incl;
    __syncthreads();
    int off = incl - c, total = 0;
#pragma unroll
    for (int wv = 0; wv < POST_THREADS / 64; ++wv) {
        const int n = sh.wave_cnt[wv];
        if (wv < wave) off += n;
        total += n;
    }
#pragma unroll
    for (int u = 0; u < POST_MAX_CPT; ++u)
        if ((mask >> u) & 1u) keys[off + __popc(mask & ((1u << u) - 1u))] = mine[u];
    __syncthreads();
    return total;
}

// ---- phase 2: bitonic sort of the unique keys (ascending); keys[M .. mpad) are padded with ~0 ----
// The compare-exchange stages whose partner distance j is below 64 stay inside one wave's 64 consecutive keys: they run in REGISTERS
// (one key per lane, partner by __shfl_xor) without a workgroup barrier between them -- all of levels k = 2 .. 64 in one pass, and the
// tail j = 32 .. 1 of every later level; only the stages with j >= 64 go through LDS with a barrier each.  2048 keys: 21 barriers
// instead of 66; 512 keys (one class of a dense frame): 10 instead of 45.  Unique keys: the result is THE sorted order either way.
__device__ __forceinline__ uint64_t bitonic_in_wave(uint64_t x, int i, int k, int jmax)
{
    const bool up = (i & k) == 0;
    for (int j = jmax; j > 0; j >>= 1) {
        const uint64_t y = __shfl_xor(x, j);
        const bool take_min = ((i & j) == 0) == up;
        x = take_min ? (x < y ? x : y) : (x < y ? y : x);
    }
    return x;
}
__device__ __forceinline__ void sort_keys(uint64_t* keys, int M)
{
    const int tid = threadIdx.x;
    int mpad = 64;
    while (mpad < M) mpad <<= 1;
    for (int i = M + tid; i < mpad; i += POST_THREADS) keys[i] = ~0ull;
    __syncthreads();
    for (int i = tid; i < mpad; i += POST_THREADS) {      // levels 2 .. 64 (a wave's iteration covers 64 consecutive keys: mpad and 1024 are multiples of 64)
        uint64_t x = keys[i];
        for (int k = 2; k <= 64; k <<= 1) x = bitonic_in_wave(x, i, k, k >> 1);
        keys[i] = x;
    }
    __syncthreads();
    for (int k = 128; k <= mpad; k <<= 1) {
        for (int j = k >> 1; j >= 64; j >>= 1) {
            for (int i = tid; i < mpad; i += POST_THREADS) {
                int l = i ^ j;
                if (l > i) {
                    uint64_t x = keys[i], y = keys[l];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                }
            }
            __syncthreads();
        }
        for (int i = tid; i < mpad; i += POST_THREADS) keys[i] = bitonic_in_wave(keys[i], i, k, 32);
        __syncthreads();
    }
}

// ---- phase 3: decode the boxes of the sorted candidates (fp64, round-half-even) ----
__device__ __forceinline__ void decode_boxes(const PostArgs& a, long frame, const uint64_t* keys, int4* boxes, unsigned char* alive, int M, PostShared& sh)
{
    for (int k = threadIdx.x; k < M; k += POST_THREADS) {
        int cell = (int)(keys[k] & 0x1fffu);
        CellRef r = locate(a, frame, cell);
        double scale_h = (double)a.in_h / r.h, scale_w = (double)a.in_w / r.w;
        double x = (r.j + sigmoid_d((double)logit_at(r, 0))) * scale_w;
        double y = (r.i + sigmoid_d((double)logit_at(r, 1))) * scale_h;
        double bw = exp((double)logit_at(r, 2)) * a.anchors[(r.head * a.na + r.pp) * 2 + 0];
        double bh = exp((double)logit_at(r, 3)) * a.anchors[(r.head * a.na + r.pp) * 2 + 1];
        boxes[k] = make_int4(clamp_i32(rint(x - bw / 2)), clamp_i32(rint(y - bh / 2)), clamp_i32(rint(x + bw / 2)),
                             clamp_i32(rint(y + bh / 2)));
        alive[k] = 1;
        if (s_area(boxes[k]) == 0) sh.zero_area = 1;
    }
    __syncthreads();
}

// ---- phase 4: greedy NMS, class segments are contiguous in the sorted list ----
// The reference's loop (detect.py:69-84) keeps the best remaining box of a class and drops every later box it overlaps; one
// barrier-separated sweep per kept box costs ~1.3 us each (dense frames: 260 survivors of 1200 candidates = 0.36 ms).  The same
// decisions in windows of 64 consecutive candidates of one class:
//   (1) all pairs inside the window at once: thread (row i = tid / PARTS, part q = tid % PARTS) tests i against the CPT candidates
//       of its part -> a piece of row i's suppression mask (and of its "union == 0" mask, the reference's
//       ZeroDivisionError, which must only count if the reference would have evaluated that pair);
//   (2) wave 0 resolves the window greedily on 64-bit masks: take the lowest remaining bit, keep it, clear what it suppresses;
//   (3) every later candidate of the class tests itself against the window's survivors, in order, until one drops it.
// Identical to the sequential procedure: a box is dropped iff an EARLIER KEPT box of its class overlaps it.
__device__ __forceinline__ void greedy_nms(const PostArgs& a, const uint64_t* keys, const int4* boxes, uint16_t* kept, unsigned char* alive, int M, PostShared& sh)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool skip0 = !(0.0 > a.nms_thres);  // with a non-negative threshold a zero intersection can never suppress
    for (int pos = 0; pos < M;) {
        const int cls = (int)(keys[pos] >> 45);
        if (wave == 0) {  // window = the run of this class starting at pos, at most 64 long (sorted: the class is a contiguous run)
            const bool in = pos + lane < M && (int)(keys[pos + lane] >> 45) == cls;
            const unsigned long long b = __ballot(in);
            if (lane == 0) sh.wlen = __popcll(b);   // the lanes of this class form a prefix
        }
        __syncthreads();
        const int wlen = sh.wlen;
        {   // (1)
            const int i = tid / POST_PARTS, q = tid % POST_PARTS;
            unsigned pm = 0, pz = 0;
            if (i < wlen && alive[pos + i]) {
                const int4 bi = boxes[pos + i];
                const long area_i = ((long)bi.z - bi.x) * ((long)bi.w - bi.y);
                for (int t = 0; t < POST_CPT; ++t) {
                    const int j = POST_CPT * q + t;
                    if (j <= i || j >= wlen || !alive[pos + j]) continue;
                    const int4 bj = boxes[pos + j];
                    const long iw = (long)min(bj.z, bi.z) - (long)max(bj.x, bi.x);
                    const long ih = (long)min(bj.w, bi.w) - (long)max(bj.y, bi.y);
                    if ((iw <= 0 || ih <= 0) && skip0) {  // disjoint (most pairs of a dense frame): IoU = 0 can never exceed a threshold >= 0;
                        if ((area_i | s_area(bj)) == 0) pz |= 1u << t;  // ... but 0 / 0 is the reference's ZeroDivisionError
                        continue;
                    }
                    const long inter = (iw > 0 && ih > 0) ? iw * ih : 0;
                    const long uni = s_area(bj) + area_i - inter;
                    if (uni == 0) { pz |= 1u << t; continue; }
                    if ((double)inter / (double)uni > a.nms_thres) pm |= 1u << t;
                }
            }
            sh.pm[i][q] = (uint16_t)pm; sh.pz[i][q] = (uint16_t)pz;
        }
        __syncthreads();
        if (wave == 0) {  // (2)
            unsigned long long row = 0, zrow = 0;
#pragma unroll
            for (int q = 0; q < POST_PARTS; ++q) {
                row |= (unsigned long long)sh.pm[lane][q] << (POST_CPT * q); zrow |= (unsigned long long)sh.pz[lane][q] << (POST_CPT * q);
            }
            const bool al = lane < wlen && alive[pos + lane];
            unsigned long long rem = __ballot(al);   // candidates not yet decided
            const bool zany = __ballot(zrow != 0) != 0;   // (almost never: two zero-area boxes in one window)
            unsigned long long keptbits = 0;
            int err = 0;
            while (rem) {
                const int i = __ffsll((long long)rem) - 1;   // wave-uniform
                const unsigned lo = __builtin_amdgcn_readlane((unsigned)row, i), hi = __builtin_amdgcn_readlane((unsigned)(row >> 32), i);
                rem &= ~(1ull << i);
                if (zany) {
                    const unsigned zlo = __builtin_amdgcn_readlane((unsigned)zrow, i), zhi = __builtin_amdgcn_readlane((unsigned)(zrow >> 32), i);
                    if ((((unsigned long long)zhi << 32) | zlo) & rem) err = 1;  // the reference divides by this pair's union while both are in its list
                }
                rem &= ~(((unsigned long long)hi << 32) | lo);
                keptbits |= 1ull << i;
            }
            // every lane publishes its own decision: alive := kept; the window's survivors (in order) with boxes and areas
            const int nk0 = sh.nkept, wk = __popcll(keptbits);
            if (lane < wlen) alive[pos + lane] = (keptbits >> lane) & 1;
            if ((keptbits >> lane) & 1) {
                const int rank = __popcll(keptbits & ((1ull << lane) - 1));
                const int4 b = boxes[pos + lane];
                kept[nk0 + rank] = (uint16_t)(pos + lane);
                sh.wb[rank] = b; sh.wa[rank] = s_area(b);
            }
            if (lane == 0) { sh.nkept = nk0 + wk; sh.wk = wk; if (err) sh.err = 1; }
        }
        __syncthreads();
        {   // (3)
            const int wk = sh.wk;
            if (!sh.zero_area && skip0) {
                // no zero-area box in the frame: a union can never be 0, so "dropped by the FIRST survivor that overlaps it" is just
                // "dropped if ANY survivor overlaps it" -- order-free: SPLIT threads share a candidate and take every SPLIT-th survivor
                constexpr int SPLIT = 4;
                const int sub = tid % SPLIT;
                for (int j = pos + wlen + tid / SPLIT; j < M; j += POST_THREADS / SPLIT) {
                    if ((int)(keys[j] >> 45) != cls) break;  // past this class
                    if (!alive[j]) continue;
                    const int4 bj = boxes[j];
                    const long area_j = s_area(bj);
                    for (int k = sub; k < wk; k += SPLIT) {
                        const int4 bi = sh.wb[k];
                        const long iw = (long)min(bj.z, bi.z) - (long)max(bj.x, bi.x);
                        const long ih = (long)min(bj.w, bi.w) - (long)max(bj.y, bi.y);
                        if (iw <= 0 || ih <= 0) continue;
                        const long inter = iw * ih;
                        if ((double)inter / (double)(area_j + sh.wa[k] - inter) > a.nms_thres) { alive[j] = 0; break; }
                    }
                }
            } else {
                for (int j = pos + wlen + tid; j < M; j += POST_THREADS) {
                    if ((int)(keys[j] >> 45) != cls) break;  // past this class
                    if (!alive[j]) continue;
                    const int4 bj = boxes[j];
                    const long area_j = s_area(bj);
                    for (int k = 0; k < wk; ++k) {
                        const int4 bi = sh.wb[k];           // the same address for every lane: an LDS broadcast
                        const long iw = (long)min(bj.z, bi.z) - (long)max(bj.x, bi.x);
                        const long ih = (long)min(bj.w, bi.w) - (long)max(bj.y, bi.y);
                        if ((iw <= 0 || ih <= 0) && skip0) {
                            if ((area_j | sh.wa[k]) == 0) sh.err = 1;
                            continue;
                        }
                        const long inter = (iw > 0 && ih > 0) ? iw * ih : 0;
                        const long uni = area_j + sh.wa[k] - inter;
                        if (uni == 0) { sh.err = 1; continue; }  // the reference raises ZeroDivisionError here
                        if ((double)inter / (double)uni > a.nms_thres) { alive[j] = 0; break; }
                    }
                }
            }
        }
        __syncthreads();
        pos += wlen;
    }
}

// ---- phase 5: one survivor into the caller's arrays / packed record row ----
__device__ __forceinline__ void emit_survivor(const PostArgs& a, int32_t* rec, long frame, int k, int4 b, int cls, int cell)
{
    CellRef r = locate(a, frame, cell);
    if (a.adj_w != 0.0) {  // __adjust_coord: int * float scale, round-half-even
        b.x = clamp_i32(rint((double)b.x * a.adj_w)); b.z = clamp_i32(rint((double)b.z * a.adj_w));
        b.y = clamp_i32(rint((double)b.y * a.adj_h)); b.w = clamp_i32(rint((double)b.w * a.adj_h));
    }
    const float conf = (float)sigmoid_d((double)logit_at(r, 4)), score = (float)sigmoid_d((double)logit_at(r, 5 + cls));
    if (rec) {   // (rows start at an odd int32 offset: no 16-byte stores here)
        int32_t* bx = rec + 1 + 4 * k;
        bx[0] = b.x; bx[1] = b.y; bx[2] = b.z; bx[3] = b.w;
        rec[1 + 4 * a.kmax + 2 * k] = __float_as_int(conf);
        rec[1 + 4 * a.kmax + 2 * k + 1] = __float_as_int(score);
        rec[1 + 6 * a.kmax + k] = cls;
        rec[1 + 7 * a.kmax + k] = cell;
        return;
    }
    long o = frame * a.kmax + k;
    reinterpret_cast<int4*>(a.boxes)[o] = b;
    a.scores[o * 2 + 0] = conf;
    a.scores[o * 2 + 1] = score;
    a.cls[o] = cls;
    a.src[o] = cell;
}

// LDS carve: keys u64[mpad] | boxes int4[ncell] | kept u16[ncell] | alive u8[ncell] | small scalars
// (worst case 512x640: 8192*8 + 4800*19 = 156.7 KB of the CU's 160 KB)
__global__ void __launch_bounds__(POST_THREADS) post_kernel(PostArgs a, int ncell, int mpad_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int4* boxes = reinterpret_cast<int4*>(smem + (size_t)mpad_max * 8);
    uint16_t* kept = reinterpret_cast<uint16_t*>(smem + (size_t)mpad_max * 8 + (size_t)ncell * 16);
    unsigned char* alive = smem + (size_t)mpad_max * 8 + (size_t)ncell * 18;
    __shared__ PostShared sh;

    const int tid = threadIdx.x;
    const long frame = blockIdx.x;
    if (tid == 0) { sh.nkept = 0; sh.err = 0; sh.zero_area = 0; }
    const int M = compact_keys(a, frame, ncell, keys, -1, sh);    // (its barriers also publish the three scalars)
    int32_t* const rec = a.records ? a.records + frame * (1 + 8L * a.kmax) : nullptr;   // packed record row of this frame
    if (M == 0) {
        if (tid == 0) { if (rec) rec[0] = 0; else a.counts[frame] = 0; }
        return;
    }
    sort_keys(keys, M);
    decode_boxes(a, frame, keys, boxes, alive, M, sh);
    greedy_nms(a, keys, boxes, kept, alive, M, sh);

    const int nk = sh.nkept;
    if (tid == 0) { if (rec) rec[0] = sh.err ? -2 : nk; else a.counts[frame] = sh.err ? -2 : nk; }
    const int nw = nk < a.kmax ? nk : a.kmax;
    for (int k = tid; k < nw; k += POST_THREADS) {
        const int i = kept[k];
        const uint64_t key = keys[i];
        emit_survivor(a, rec, frame, k, boxes[i], (int)(key >> 45), (int)(key & 0x1fffu));
    }
}

// class-major concatenation of a frame's per-class scratch rows into the caller's arrays (detect.py:169); any number of threads
__device__ __forceinline__ void assemble_frame(const PostArgs& a, long frame)
{
    const long row_ints = 2 + 5L * a.kmax;
    const int32_t* rows = a.split_tmp + frame * a.nc * row_ints;
    int total = 0, err = 0;
    for (int c = 0; c < a.nc; ++c) { total += rows[c * row_ints]; err |= rows[c * row_ints + 1]; }
    int32_t* const rec = a.records ? a.records + frame * (1 + 8L * a.kmax) : nullptr;
    if (threadIdx.x == 0) { if (rec) rec[0] = err ? -2 : total; else a.counts[frame] = err ? -2 : total; }
    const int nw = total < a.kmax ? total : a.kmax;
    for (int k = threadIdx.x; k < nw; k += blockDim.x) {
        int c = 0, kk = k;
        for (;; ++c) {                        // class of output slot k
            const int n = rows[c * row_ints];
            if (kk < n) break;
            kk -= n;
        }
        const int32_t* e = rows + c * row_ints + 2 + 5 * kk;   // kk < kmax: k < kmax and kk <= k
        emit_survivor(a, rec, frame, k, make_int4(e[0], e[1], e[2], e[3]), c, e[4]);
    }
}

// ------------------------------------------------------------------------------------------------
// post_split_kernel (round 6): dense frames.  The reference runs NMS per class (detect.py:158-169: bucket, sort, NMS,
// concatenate in class order): the classes of a frame are independent work.  One workgroup per (frame, class): it compacts the
// candidates whose argmax class is its own (decode order), sorts them by conf, decodes them and runs the same windowed greedy NMS on
// them -- the arithmetic, and therefore the survivors and their order inside the class, of post_kernel -- and leaves its survivors
// (box, cell) with their count and error flag in a scratch row; post_assemble_kernel, a second small launch, concatenates the classes in
// class order into the caller's arrays.  BASELINE configs[4]'s per-GPU share (64 dense 640x512 frames: 1210 candidates, 260 survivors per
// frame) then uses 192 of the 256 CUs instead of 64: 0.127 -> 0.058 ms per 64 frames (tools/dense_post_bench.py).  Measured and dropped: the
// assembly inside this launch by the frame's LAST workgroup to finish (device-scope ticket: release fence -> atomic increment -> acquire
// fence) -- a device-scope release is an L2 write-back on this part (the class workgroups sit on different XCDs): 0.076 ms with one fence
// per workgroup, 0.136 ms with one per thread, and 0.23 ms against 0.075 ms for 768 workgroups; the kernel boundary is the cheaper fence.
// scratch row of (frame, class): int32 [2 + 5 kmax] = count | err | kmax x (x1, y1, x2, y2, cell)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(POST_THREADS) post_split_kernel(PostArgs a, int ncell, int mpad_max)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    int4* boxes = reinterpret_cast<int4*>(smem + (size_t)mpad_max * 8);
    uint16_t* kept = reinterpret_cast<uint16_t*>(smem + (size_t)mpad_max * 8 + (size_t)ncell * 16);
    unsigned char* alive = smem + (size_t)mpad_max * 8 + (size_t)ncell * 18;
    __shared__ PostShared sh;

    const int tid = threadIdx.x;
    const long frame = blockIdx.x / a.nc;
    const int mycls = blockIdx.x - (int)frame * a.nc;
    const long row_ints = 2 + 5L * a.kmax;
    int32_t* const row = a.split_tmp + (frame * a.nc + mycls) * row_ints;
    if (tid == 0) { sh.nkept = 0; sh.err = 0; sh.zero_area = 0; }
    const int M = compact_keys(a, frame, ncell, keys, mycls, sh);
    if (M > 0) {
        sort_keys(keys, M);
        decode_boxes(a, frame, keys, boxes, alive, M, sh);
        greedy_nms(a, keys, boxes, kept, alive, M, sh);
    }
    const int nk = M > 0 ? sh.nkept : 0;
    if (tid == 0) { row[0] = nk; row[1] = M > 0 ? sh.err : 0; }
    for (int k = tid; k < (nk < a.kmax ? nk : a.kmax); k += POST_THREADS) {
        const int i = kept[k];
        const int4 b = boxes[i];
        int32_t* e = row + 2 + 5 * k;
        e[0] = b.x; e[1] = b.y; e[2] = b.z; e[3] = b.w; e[4] = (int)(keys[i] & 0x1fffu);
    }
}

// ... and the second launch: one small workgroup per frame concatenates the frame's class rows in class order.  The kernel boundary is the
// release / acquire between the class workgroups (which run on different XCDs, each behind its own L2) and this one.
__global__ void __launch_bounds__(256) post_assemble_kernel(PostArgs a)
{
    assemble_frame(a, blockIdx.x);
}

// Stand-alone greedy NMS over one class's conf-sorted list (detect.py:69-84); one workgroup.
__global__ void __launch_bounds__(POST_THREADS) nms_sorted_kernel(const int4* __restrict__ boxes, int n, double thres,
                                                                  int32_t* sup)
{
    __shared__ int s_err;
    const int tid = threadIdx.x;
    if (tid == 0) s_err = 0;
    for (int i = tid; i < n; i += POST_THREADS) sup[i] = -1;
    __syncthreads();
    for (int i = 0; i < n; ++i) {
        if (sup[i] != -1) continue;
        const int4 bi = boxes[i];
        const long area_i = ((long)bi.z - bi.x) * ((long)bi.w - bi.y);
        for (int j = i + 1 + tid; j < n; j += POST_THREADS) {
            if (sup[j] != -1) continue;
            const int4 bj = boxes[j];
            long iw = (long)min(bj.z, bi.z) - (long)max(bj.x, bi.x);
            long ih = (long)min(bj.w, bi.w) - (long)max(bj.y, bi.y);
            long inter = (iw > 0 && ih > 0) ? iw * ih : 0;
            long uni = ((long)bj.z - bj.x) * ((long)bj.w - bj.y) + area_i - inter;
            if (uni == 0) { s_err = 1; continue; }
            if ((double)inter / (double)uni > thres) sup[j] = i;
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_err)
        for (int i = tid; i < n; i += POST_THREADS) sup[i] = -2;
}

void launch_nms_sorted(const int32_t* boxes, int n, double nms_thres, int32_t* suppressor, hipStream_t s)
{
    hipLaunchKernelGGL(nms_sorted_kernel, dim3(1), dim3(POST_THREADS), 0, s, reinterpret_cast<const int4*>(boxes), n,
                       nms_thres, suppressor);
}

// ================================================================================================
// Validation-time decode + NMS -- the OTHER convention of the reference:
//   src/model_training/loss/yolo_loss.py:48-68,98-141   fp32 decode of one head to (cx, cy, w, h, conf, cls...)
//   src/model_training/utils/general.py:29-52, 87-143   corners, conf >= thres, per-class greedy NMS, IoU with +1, keep iou < thres
// ================================================================================================
struct ValAnchors { float wh[2 * POST_MAX_ANCHORS]; };
__global__ void __launch_bounds__(256) val_decode_kernel(const float* __restrict__ in, float* __restrict__ out, long total, int h, int w,
                                                          int M_total, int m_off, ValAnchors anc, int na, int nc, float stride_w, float stride_h)
{
    long idx = (long)blockIdx.x * 256 + threadIdx.x;  // (frame, anchor, i, j)
    if (idx >= total) return;
    const int j = (int)(idx % w);
    long t = idx / w;
    const int i = (int)(t % h);
    t /= h;
    const int a = (int)(t % na);
    const long n = t / na;
    const int attrs = 5 + nc;
    const float* p = in + ((n * na + a) * attrs) * (long)h * w + (long)i * w + j;
    const long hw = (long)h * w;
    auto sg = [](float v) { return 1.f / (1.f + expf(-v)); };
    const float aw = anc.wh[2 * a], ah = anc.wh[2 * a + 1];
    float* o = out + (n * M_total + m_off + ((long)a * h + i) * w + j) * attrs;
    o[0] = (sg(p[0]) + (float)j) * stride_w;          // (x + grid_x) * stride_w        yolo_loss.py:133,139
    o[1] = (sg(p[hw]) + (float)i) * stride_h;
    o[2] = (expf(p[2 * hw]) * aw) * stride_w;         // exp(w) * anchor_w (feature-map units), then * stride
    o[3] = (expf(p[3 * hw]) * ah) * stride_h;
    for (int k = 4; k < attrs; ++k) o[k] = sg(p[k * hw]);   // conf, class scores
}

void launch_val_decode(const float* in, float* out, int N, int h, int w, int M_total, int m_off, const float* anc, int na, int nc, float stride_w,
                       float stride_h, hipStream_t s)
{
    long total = (long)N * na * h * w;
    ValAnchors va{};
    for (int i = 0; i < 2 * na && i < 2 * POST_MAX_ANCHORS; ++i) va.wh[i] = anc[i];
    hipLaunchKernelGGL(val_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total, h, w, M_total, m_off,
                       va, na, nc, stride_w, stride_h);
}

// One workgroup per image.  All arithmetic after the decode is fp32 add/mul/div, so given the same prediction tensor the
// result is bit-identical to the reference's torch code (ties in conf: index order; the reference's sort is unstable).
__global__ void __launch_bounds__(POST_THREADS) val_nms_kernel(const float* __restrict__ pred, int M, int mpad_max, int nc, float conf_thres,
                                                               float nms_thres, int kmax, float* __restrict__ det, int32_t* counts)
{
    const int attrs = 5 + nc;   // floats per prediction row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    uint16_t* kept = reinterpret_cast<uint16_t*>(smem + (size_t)mpad_max * 8);
    unsigned char* alive = smem + (size_t)mpad_max * 8 + (size_t)M * 2;
    __shared__ int s_wave_cnt[POST_THREADS / 64];
    __shared__ int s_total, s_nkept;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* P = pred + (long)blockIdx.x * M * attrs;
    if (tid == 0) { s_total = 0; s_nkept = 0; }
    __syncthreads();
    for (int base = 0; base < M; base += POST_THREADS) {
        const int m = base + tid;
        bool pass = false;
        uint64_t key = 0;
        if (m < M) {
            const float conf = P[(long)m * attrs + 4];
            pass = conf >= conf_thres;
            if (pass) {
                float best = P[(long)m * attrs + 5];
                int cls = 0;
                for (int k = 1; k < nc; ++k) { float v = P[(long)m * attrs + 5 + k]; if (v > best) { best = v; cls = k; } }   // torch.max: first maximum
                key = ((uint64_t)cls << 45) | ((uint64_t)(~__float_as_uint(conf)) << 13) | (uint64_t)m;
            }
        }
        unsigned long long bm = __ballot(pass);
        int before = __popcll(bm & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave_cnt[wave] = __popcll(bm);
        __syncthreads();
        int off = s_total;
        for (int wv = 0; wv < wave; ++wv) off += s_wave_cnt[wv];
        if (pass) keys[off + before] = key;
        __syncthreads();
        if (tid == 0) { int tt = s_total; for (int wv = 0; wv < POST_THREADS / 64; ++wv) tt += s_wave_cnt[wv]; s_total = tt; }
        __syncthreads();
    }
    const int K = s_total;
    if (K == 0) { if (tid == 0) counts[blockIdx.x] = 0; return; }
    int mpad = 64;
    while (mpad < K) mpad <<= 1;
    for (int i = K + tid; i < mpad; i += POST_THREADS) keys[i] = ~0ull;
    __syncthreads();
    for (int k = 2; k <= mpad; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < mpad; i += POST_THREADS) {
                int l = i ^ j;
                if (l > i) {
                    uint64_t x = keys[i], y = keys[l];
                    bool up = (i & k) == 0;
                    if ((x > y) == up) { keys[i] = y; keys[l] = x; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += POST_THREADS) alive[i] = 1;
    __syncthreads();
    auto corners = [&](int k) {  // general.py:90-95, fp32
        const float* q = P + (long)(keys[k] & 0x1fffu) * attrs;
        const float cx = q[0], cy = q[1], w = q[2], h = q[3];
        return make_float4(cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2);
    };
    for (int i = 0; i < K; ++i) {
        if (!alive[i]) continue;
        if (tid == 0) { kept[s_nkept] = (uint16_t)i; s_nkept = s_nkept + 1; }
        const float4 bi = corners(i);
        const int ci = (int)(keys[i] >> 45);
        const float a1 = (bi.z - bi.x + 1) * (bi.w - bi.y + 1);
        for (int j = i + 1 + tid; j < K; j += POST_THREADS) {
            if ((int)(keys[j] >> 45) != ci) break;
            if (!alive[j]) continue;
            const float4 bj = corners(j);
            const float iw = fmaxf(fminf(bi.z, bj.z) - fmaxf(bi.x, bj.x) + 1, 0.f);
            const float ih = fmaxf(fminf(bi.w, bj.w) - fmaxf(bi.y, bj.y) + 1, 0.f);
            const float inter = iw * ih;
            const float a2 = (bj.z - bj.x + 1) * (bj.w - bj.y + 1);
            const float iou = inter / (a1 + a2 - inter + 1e-16f);
            if (!(iou < nms_thres)) alive[j] = 0;
        }
        __syncthreads();
    }
    __syncthreads();
    const int nk = s_nkept;
    if (tid == 0) counts[blockIdx.x] = nk;
    for (int k = tid; k < (nk < kmax ? nk : kmax); k += POST_THREADS) {
        const int i = kept[k];
        const float* q = P + (long)(keys[i] & 0x1fffu) * attrs;
        const float4 c = corners(i);
        const int cls = (int)(keys[i] >> 45);
        float* o = det + ((long)blockIdx.x * kmax + k) * 7;
        o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w; o[4] = q[4]; o[5] = q[5 + cls]; o[6] = (float)cls;
    }
}

static int pow2_at_least(int n);
int launch_val_nms(const float* pred, int N, int M, int nc, float conf_thres, float nms_thres, int kmax, float* det, int32_t* counts, hipStream_t s)
{
    if (M > 8191 || nc < 1) return -1;
    int mp = pow2_at_least(M);
    size_t lds = (size_t)mp * 8 + (size_t)M * 3 + 16;
    static size_t attr_set[YF_MAX_DEVICES] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    if (lds > 48 * 1024 && lds > attr_set[dev]) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(val_nms_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_set[dev] = lds;
    }
    hipLaunchKernelGGL(val_nms_kernel, dim3(N), dim3(POST_THREADS), lds, s, pred, M, mp, nc, conf_thres, nms_thres, kmax, det, counts);
    return 0;
}

static int pow2_at_least(int n)
{
    int p = 64;
    while (p < n) p <<= 1;
    return p;
}

size_t post_lds_bytes(int ncell)
{
    size_t b = (size_t)pow2_at_least(ncell) * 8 + (size_t)ncell * 16 + (size_t)ncell * 2 + (size_t)ncell;
    return (b + 15) & ~(size_t)15;
}

size_t post_split_tmp_ints(int N, int nc, int kmax) { return (size_t)N * nc * (2 + 5 * (size_t)kmax); }

int launch_post(const PostArgs& a, int N, hipStream_t s)
{
    if (a.na < 1 || a.na > POST_MAX_ANCHORS || a.nc < 1 || a.nc >= (1 << 18)) return -1;   // (the class sits above bit 45 of the sort key)
    int ncell = a.na * (a.hl * a.wl + a.hs * a.ws);
    if (ncell > 8191) return -1;  // 13-bit cell field of the sort key
    size_t lds = post_lds_bytes(ncell);
    if (lds + sizeof(PostShared) + 64 > 160 * 1024) return -1;  // dynamic + static LDS of one CU
    static size_t attr_set[YF_MAX_DEVICES][2] = {};
    const int dev = current_device();
    if (dev < 0) return -2;
    const bool split = a.split_tmp != nullptr;
    if (lds > attr_set[dev][split]) {
        if (hipFuncSetAttribute(split ? reinterpret_cast<const void*>(post_split_kernel) : reinterpret_cast<const void*>(post_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -2;
        attr_set[dev][split] = lds;
    }
    if (split) {
        hipLaunchKernelGGL(post_split_kernel, dim3((unsigned)(N * a.nc)), dim3(POST_THREADS), lds, s, a, ncell, pow2_at_least(ncell));
        hipLaunchKernelGGL(post_assemble_kernel, dim3((unsigned)N), dim3(256), 0, s, a);
    }
    else hipLaunchKernelGGL(post_kernel, dim3(N), dim3(POST_THREADS), lds, s, a, ncell, pow2_at_least(ncell));
    return 0;
}

}  // namespace yf
