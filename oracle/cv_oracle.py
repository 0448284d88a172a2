"""TEST INFRASTRUCTURE (never imported by the product; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/).

CPU restatement of the two OpenCV calls of `Detect_YOLO.__pre_process` (src/detect.py:107-116):

    ori_img = cv2.imread(img_path)                                   # u8 [h, w, 3], BGR
    img = cv2.cvtColor(ori_img, cv2.COLOR_BGR2GRAY)                  # :110-111, 1-channel nets on 3-channel originals
    img = cv2.resize(img, (input_shape[1], input_shape[0]))          # :115-116, INTER_LINEAR (the default), when the sizes differ

**PARITY UNPINNED vs OpenCV.**  OpenCV is a third-party dependency of the reference (un-vendored, no version pinned in its README;
`cv2` is absent from this image and cannot be installed), the reference has no test or golden vector at this boundary, and its
`test_data` frames only exercise the exact-2x case (640x512 -> 320x256: the reproduced has-target flags of its logs support, but do not
prove, byte identity).  What is restated here is OpenCV's published 8-bit arithmetic (modules/imgproc/src/color_rgb.simd.hpp `RGB2Gray<uchar>`,
modules/imgproc/src/resize.cpp `resizeGeneric_` / `HResizeLinear` / `VResizeLinear<uchar, int, short, FixedPtCast<...>>` /
`ResizeAreaFastVec`); an IPP- or vendor-HAL-accelerated OpenCV build may round differently.  Independent checks in tests/: agreement
with torch's float bilinear (align_corners=False == OpenCV's half-pixel centres) to 1 LSB, constants and ramps, the 2x case against the
committed golden input frames.

Numerics restated:
  * BGR2GRAY, 8-bit: gray = (B*BY + G*GY + R*RY + (1 << (shift-1))) >> shift with (RY, GY, BY, shift) = (4899, 9617, 1868, 14) -- OpenCV
    up to 4.x's `yuv_shift` form -- or (9798, 19235, 3735, 15), the `gray_shift = 15` form of newer 4.x builds (`bits=15`).
  * resize, INTER_LINEAR, 8-bit: per destination column dx: fx = float((dx + 0.5) * scale_x - 0.5), sx = floor(fx), fx -= sx; sx < 0 -> sx = 0,
    fx = 0; sx >= w - 1 -> sx = w - 1, fx = 0; coefficients cvRound((1 - fx) * 2048), cvRound(fx * 2048) as int16 (INTER_RESIZE_COEF_BITS = 11);
    rows likewise without the fx reset (row indices clamped instead).  Horizontal pass in int32: S[sx] * a0 + S[sx + 1] * a1; vertical pass
    dst = (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
  * resize by EXACTLY 1/2 in both directions: cv::resize turns INTER_LINEAR into INTER_AREA (`is_area_fast && iscale_x == 2 && iscale_y == 2`),
    whose 8-bit fast path is (a + b + c + d + 2) >> 2 over the 2x2 block (the rule the fused pre-process of rounds 1-4 implements).
"""
import numpy as np

INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS
GRAY_COEFFS = {14: (4899, 9617, 1868), 15: (9798, 19235, 3735)}      # (R, G, B)


def cvt_bgr2gray(bgr, bits=15):
    """cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) for uint8 [..., 3] (B, G, R) -> uint8 [...]."""
    ry, gy, by = GRAY_COEFFS[bits]
    a = np.asarray(bgr).astype(np.int64)
    return ((a[..., 0] * by + a[..., 1] * gy + a[..., 2] * ry + (1 << (bits - 1))) >> bits).astype(np.uint8)


def _cv_round(x):
    """cvRound of a float32 array: round half to even (lrintf / cvtss2si)."""
    return np.rint(x.astype(np.float32)).astype(np.int64)


def linear_tables(src, dst):
    """The per-destination-index tables cv::resize builds for INTER_LINEAR along one axis.
    -> (ofs int32 [dst], coef int16 [dst, 2]) WITH the horizontal pass's edge resets (reset=True semantics are applied by the caller)."""
    scale = np.float64(1.0) / (np.float64(dst) / np.float64(src))      # `inv_scale_x = (double)dsize.width / ssize.width; scale_x = 1. / inv_scale_x`
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)        # `fx = (float)((dx+0.5)*scale_x - 0.5)`
    s = np.floor(f).astype(np.int64)                         # cvFloor
    f = (f - s.astype(np.float32)).astype(np.float32)        # `fx -= sx` in float
    return s, f


def resize_linear_u8(img, dsize_wh, _force_linear=False):
    """cv2.resize(img, (w, h)) with the default interpolation for uint8 [h, w] or [h, w, c]."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    sh, sw = img.shape[:2]
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if (sh, sw) == (dh, dw):
        return img.copy()
    if not _force_linear and sw == 2 * dw and sh == 2 * dh:  # INTER_LINEAR -> INTER_AREA, 2x2 fast path
        a = img.astype(np.int64)
        return ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    sx, fx = linear_tables(sw, dw)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    a1 = _cv_round(fx * np.float32(INTER_RESIZE_COEF_SCALE))
    a0 = _cv_round((np.float32(1) - fx) * np.float32(INTER_RESIZE_COEF_SCALE))
    sy, fy = linear_tables(sh, dh)
    b1 = _cv_round(fy * np.float32(INTER_RESIZE_COEF_SCALE))
    b0 = _cv_round((np.float32(1) - fy) * np.float32(INTER_RESIZE_COEF_SCALE))
    y0, y1 = np.clip(sy, 0, sh - 1), np.clip(sy + 1, 0, sh - 1)
    x1 = np.minimum(sx + 1, sw - 1)
    a = img.astype(np.int64)
    if a.ndim == 2:
        a = a[:, :, None]
    rows = a[:, sx, :] * a0[None, :, None] + a[:, x1, :] * a1[None, :, None]          # horizontal pass, int32 in OpenCV (max 255 * 2048)
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    out = out.astype(np.uint8)
    return out[:, :, 0] if img.ndim == 2 else out


def cv_pre_process_u8(ori_bgr, input_shape, origin_img_shape, gray_bits=15):
    """detect.py:107-118 up to (not including) the float arithmetic: cv2.imread's frame -> the uint8 image `img` that is normalised next.
    ori_bgr uint8 [h, w, 3]; returns uint8 [H, W] for a 1-channel net on 3-channel originals, else [H, W, 3] (still BGR: the channel
    reversal of :119 belongs to backbone_oracle.preprocess)."""
    img = np.asarray(ori_bgr)
    if input_shape[2] == 1 and origin_img_shape[2] != 1:
        img = cvt_bgr2gray(img, gray_bits)
    if list(input_shape[0:2]) != list(origin_img_shape[0:2]):
        img = resize_linear_u8(img, (input_shape[1], input_shape[0]))
    return img
