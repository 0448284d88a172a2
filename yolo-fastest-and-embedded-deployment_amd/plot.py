"""`plot_one_box` -- the reference's box + label drawing (src/model_training/utils/general.py:56-67) without OpenCV.

Same signature and geometry: rectangle c1-c2 with `line_thickness` (default round(0.002 * (h + w) / 2) + 1), a FILLED label
box from c1 to (c1.x + text_w, c1.y - text_h - 3) and the label text at (c1.x, c1.y - 2) in [225, 255, 255].
The reference hands cv2 a BGR image, BGR colours, and cv2.imwrite stores what a viewer sees as RGB; here `img` is an RGB numpy
array (PIL decode), so a colour given in the reference's order is reversed before drawing -- the saved file shows the same
colours as the reference's result images (tests/golden/golden_results.npz holds samples of those).  What cannot be
reproduced without cv2 is the glyph rasterisation (Hershey simplex at scale tl/5): PIL's built-in font is used at the height
cv2.getTextSize reports for that scale (~22 * scale px)."""
import numpy as np


def plot_one_box(xyxy, img, color=None, label=None, line_thickness=None):
    """Draws in place on `img` (uint8 RGB [h,w,3], C-contiguous) and returns it."""
    from PIL import Image, ImageDraw, ImageFont
    tl = line_thickness or round(0.002 * (img.shape[0] + img.shape[1]) / 2) + 1
    if color is None:
        import random
        color = [random.randint(0, 255) for _ in range(3)]
    rgb = tuple(int(c) for c in reversed(color))          # the reference's colours are BGR
    c1, c2 = (int(xyxy[0]), int(xyxy[1])), (int(xyxy[2]), int(xyxy[3]))
    im = Image.fromarray(img)
    d = ImageDraw.Draw(im)
    # cv2.rectangle(thickness=tl) strokes a line of width tl CENTRED on the rectangle's edges
    lo, hi = tl // 2, tl - 1 - tl // 2
    x1, x2 = min(c1[0], c2[0]), max(c1[0], c2[0])
    y1, y2 = min(c1[1], c2[1]), max(c1[1], c2[1])
    for (a, b, c, e) in ((x1 - lo, y1 - lo, x2 + hi, y1 + hi), (x1 - lo, y2 - lo, x2 + hi, y2 + hi),
                         (x1 - lo, y1 - lo, x1 + hi, y2 + hi), (x2 - lo, y1 - lo, x2 + hi, y2 + hi)):
        d.rectangle([a, b, c, e], fill=rgb)
    if label:
        scale = tl / 5.0
        th = max(int(round(22 * scale)), 6)                # cv2.getTextSize(FONT_HERSHEY_SIMPLEX, scale)[0][1] ~ 22 * scale
        try:
            font = ImageFont.load_default(size=th + 2)
        except TypeError:                                  # older Pillow: fixed-size bitmap font
            font = ImageFont.load_default()
        l, t, r, b = d.textbbox((0, 0), label, font=font)
        tw = r - l
        lc2 = (c1[0] + tw, c1[1] - th - 3)
        d.rectangle([c1[0], lc2[1], lc2[0], c1[1]], fill=rgb)
        d.text((c1[0], c1[1] - 2 - th - t), label, fill=(255, 255, 225), font=font)   # [225, 255, 255] BGR
    img[...] = np.asarray(im)
    return img
