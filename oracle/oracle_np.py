"""Second, independent restatement of the index-math ops as explicit numpy /
pure-Python loops (TEST INFRASTRUCTURE ONLY; PARITY UNPINNED, see
oracle/__init__.py).  Written from the reference formulas, sharing no code
with oracle_torch.py, so the two can pin each other on small cases.

float32 arithmetic is done with numpy float32 scalars/arrays in the same
operation order as the reference so results are bit-comparable.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def same_pad(in_size, k, s, d=1):
    """TF SAME padding: (before, after, out)."""
    out = (in_size + s - 1) // s
    total = max((out - 1) * s + (k - 1) * d + 1 - in_size, 0)
    return total // 2, total - total // 2, out


def conv2d_same(x, w, b=None, stride=1, dilation=1):
    """Direct-loop SAME conv, NHWC x HWIO (tf.nn.conv2d semantics), float64
    accumulation for use as a numerically clean reference on tiny cases."""
    n, h, wd, ci = x.shape
    kh, kw, _, co = w.shape
    pt, _, oh = same_pad(h, kh, stride, dilation)
    pl, _, ow = same_pad(wd, kw, stride, dilation)
    y = np.zeros((n, oh, ow, co), np.float64)
    for oy in range(oh):
        for ox in range(ow):
            for ky in range(kh):
                iy = oy * stride + ky * dilation - pt
                if iy < 0 or iy >= h:
                    continue
                for kx in range(kw):
                    ix = ox * stride + kx * dilation - pl
                    if ix < 0 or ix >= wd:
                        continue
                    y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64)
    if b is not None:
        y += b.astype(np.float64)
    return y


def conv2d_transpose_k4s2_same(x, w, b=None):
    """tf.layers.conv2d_transpose(k=4, s=2, 'same'); w [KH,KW,Cout,Cin];
    scatter form: y[2*iy+ky-1, 2*ix+kx-1, co] += x[iy,ix,ci]*w[ky,kx,co,ci]."""
    n, h, wd, ci = x.shape
    co = w.shape[2]
    y = np.zeros((n, 2 * h, 2 * wd, co), np.float64)
    for iy in range(h):
        for ix in range(wd):
            for ky in range(4):
                oy = 2 * iy + ky - 1
                if oy < 0 or oy >= 2 * h:
                    continue
                for kx in range(4):
                    ox = 2 * ix + kx - 1
                    if ox < 0 or ox >= 2 * wd:
                        continue
                    y[:, oy, ox, :] += x[:, iy, ix, :].astype(np.float64) @ w[ky, kx].astype(np.float64).T
    if b is not None:
        y += b.astype(np.float64)
    return y


def resize_bilinear_legacy(x, oh, ow):
    """TF-1.13 ResizeBilinear, align_corners=False (float32, reference op order:
    top=tl+(tr-tl)*xl; bot=bl+(br-bl)*xl; out=top+(bot-top)*yl)."""
    n, h, w, c = x.shape
    if (h, w) == (oh, ow):
        return x
    x = x.astype(f32)
    y = np.empty((n, oh, ow, c), f32)
    sy, sx = f32(h) / f32(oh), f32(w) / f32(ow)
    for i in range(oh):
        src = f32(i) * sy
        y0 = int(src)
        y1 = min(y0 + 1, h - 1)
        yl = f32(src - f32(y0))
        for j in range(ow):
            srx = f32(j) * sx
            x0 = int(srx)
            x1 = min(x0 + 1, w - 1)
            xl = f32(srx - f32(x0))
            tl, tr, bl, br = x[:, y0, x0], x[:, y0, x1], x[:, y1, x0], x[:, y1, x1]
            top = tl + (tr - tl) * xl
            bot = bl + (br - bl) * xl
            y[:, i, j] = top + (bot - top) * yl
    return y


def resize_nearest_align_corners(x, oh, ow):
    n, h, w, c = x.shape
    y = np.empty((n, oh, ow, c), x.dtype)
    sy = f32(h - 1) / f32(oh - 1) if oh > 1 else f32(0)
    sx = f32(w - 1) / f32(ow - 1) if ow > 1 else f32(0)
    for i in range(oh):
        yi = min(int(np.floor(f32(i) * sy + f32(0.5))), h - 1)
        for j in range(ow):
            xi = min(int(np.floor(f32(j) * sx + f32(0.5))), w - 1)
            y[:, i, j] = x[:, yi, xi]
    return y


def warp_indices(flow):
    """core_warp.py:99-115,189-194: float32 query, clamped floor, clamped alpha."""
    n, h, w, _ = flow.shape
    fy = np.empty((n, h, w), np.int32)
    fx = np.empty((n, h, w), np.int32)
    ay = np.empty((n, h, w), f32)
    ax = np.empty((n, h, w), f32)
    fl = flow.astype(f32)
    for b in range(n):
        for y in range(h):
            for x in range(w):
                qy = f32(y) - fl[b, y, x, 0]
                qx = f32(x) - fl[b, y, x, 1]
                flo_y = min(max(f32(0), np.floor(qy)), f32(h - 2))
                flo_x = min(max(f32(0), np.floor(qx)), f32(w - 2))
                fy[b, y, x] = int(flo_y)
                fx[b, y, x] = int(flo_x)
                ay[b, y, x] = min(max(f32(0), f32(qy - flo_y)), f32(1))
                ax[b, y, x] = min(max(f32(0), f32(qx - flo_x)), f32(1))
    return fy, fx, ay, ax


def dense_image_warp(image, flow):
    """core_warp.py:139-148 with the float32 op order of the reference:
    top=ax*(tr-tl)+tl; bot=ax*(br-bl)+bl; out=ay*(bot-top)+top."""
    n, h, w, c = image.shape
    img = image.astype(f32)
    fy, fx, ay, ax = warp_indices(flow)
    out = np.empty_like(img)
    for b in range(n):
        for y in range(h):
            for x in range(w):
                y0, x0 = fy[b, y, x], fx[b, y, x]
                tl, tr = img[b, y0, x0], img[b, y0, x0 + 1]
                bl, br = img[b, y0 + 1, x0], img[b, y0 + 1, x0 + 1]
                a_x, a_y = ax[b, y, x], ay[b, y, x]
                top = a_x * (tr - tl) + tl
                bot = a_x * (br - bl) + bl
                out[b, y, x] = a_y * (bot - top) + top
    return out


def cost_volume(c1, warp, r=4):
    """core_costvol.py:20-40 with float64 accumulation (clean reference)."""
    n, h, w, c = c1.shape
    d = 2 * r + 1
    out = np.zeros((n, h, w, d * d), np.float64)
    for y in range(h):
        for x in range(w):
            for dy in range(d):
                yy = y + dy - r
                if yy < 0 or yy >= h:
                    continue
                for dx in range(d):
                    xx = x + dx - r
                    if xx < 0 or xx >= w:
                        continue
                    out[:, y, x, dy * d + dx] = (c1[:, y, x].astype(np.float64) *
                                                 warp[:, yy, xx].astype(np.float64)).sum(-1) / c
    return np.where(out > 0, out, 0.1 * out)


# --------------------------------------------------------------------------
# evaluation tail
# --------------------------------------------------------------------------
def compute_boundary_score(segmentation):
    """models/utils/general_utils.py:122-138: fraction of the four 2-pixel image borders covered by the mask (the
    corner pixels are counted by both the horizontal and the vertical strips, numerator and denominator alike)."""
    seg = np.asarray(segmentation).astype(np.float64)
    h, w = seg.shape[0], seg.shape[1]
    up, bottom, left, right = seg[0:2, :], seg[h - 2:h, :], seg[:, 0:2], seg[:, w - 2:w]
    return (up.sum() + bottom.sum() + left.sum() + right.sum()) / float(up.size + bottom.size + left.size + right.size)


def compute_IoU(gt_mask, pred_mask_f, threshold=0.1, mask_threshold=0.6):
    """test_generator.py:19-35: threshold the soft mask, flip it when it covers the image borders (>= mask_threshold),
    IoU against the boolean ground truth; 1 when both are empty.  Returns (iou, annotation) like the reference (a
    bare 1 in the empty/empty case)."""
    gt = np.asarray(gt_mask).astype(bool)
    pred = np.asarray(pred_mask_f) > threshold
    annotation = pred if compute_boundary_score(pred) < mask_threshold else np.logical_not(pred)
    if np.isclose(np.sum(annotation), 0) and np.isclose(np.sum(gt), 0):
        return 1
    return np.sum(annotation & gt) / np.sum(annotation | gt, dtype=np.float32), annotation


def compute_mae(gt_mask, pred_mask_f):
    """test_generator.py:38-40."""
    return np.mean(np.abs(np.asarray(gt_mask, dtype=np.float64) - np.asarray(pred_mask_f, dtype=np.float64)))


def compute_all_IoU(pred_masks, gt_masks, threshold=0.1, border_th=0.6, epsilon=1e-8):
    """general_utils.py:89-120,140-159 (the TF graph used for the validation IoU, adversarial_learner.py:135-139):
    gt > 0.01, mask > threshold, flipped when its border score >= 0.6, IoU = |and| / (|or| + 1e-8) per sample."""
    pred = (np.asarray(pred_masks) > threshold).astype(np.float32)
    gt = np.asarray(gt_masks) > 0.01
    out = np.zeros(pred.shape[0], np.float64)
    for b in range(pred.shape[0]):
        fg = compute_boundary_score(pred[b]) < border_th
        obj = pred[b] if fg else 1.0 - pred[b]
        ob = obj.astype(bool)
        out[b] = np.logical_and(gt[b], ob).sum() / (np.logical_or(gt[b], ob).sum() + epsilon)
    return out


# --------------------------------------------------------------------------
# input stage
# --------------------------------------------------------------------------
def resize_nearest_legacy(x, oh, ow):
    """TF-1.13 ResizeNearestNeighbor, align_corners=False: in = min(floor(out * in/out), in-1)
    (preprocess_mask: data/davis2016_data_utils.py:93-99)."""
    n, h, w, c = x.shape
    y = np.empty((n, oh, ow, c), x.dtype)
    sy, sx = f32(h) / f32(oh), f32(w) / f32(ow)
    for i in range(oh):
        yi = min(int(np.floor(f32(i) * sy)), h - 1)
        for j in range(ow):
            xi = min(int(np.floor(f32(j) * sx)), w - 1)
            y[:, i, j] = x[:, yi, xi]
    return y


def central_crop_box(h, w, frac):
    """tf.image.central_crop (TF 1.13): start = int((size - size*frac)/2), extent = size - 2*start
    (data/davis2016_data_utils.py:129-133)."""
    if frac >= 1.0:
        return 0, 0, h, w
    y0, x0 = int((h - h * frac) / 2), int((w - w * frac) / 2)
    return y0, x0, h - 2 * y0, w - 2 * x0


def preprocess_image(img_u8, oh=384, ow=640):
    """data/davis2016_data_utils.py:86-91: cast / 255 - 0.5, then legacy bilinear resize."""
    x = img_u8.astype(f32) / f32(255.0) - f32(0.5)
    return resize_bilinear_legacy(x, oh, ow)


def preprocess_mask(mask_u8, oh=384, ow=640):
    """data/davis2016_data_utils.py:93-99: cast / 255, nearest-neighbour resize."""
    return resize_nearest_legacy(mask_u8.astype(f32) / f32(255.0), oh, ow)


def flip_crop_resize(x, y0, x0, ch, cw, flip_lr, flip_td, nearest=False):
    """flip (data/aug_flips.py:3-16), crop window, resize back to the input size
    (random_crop_image_pair :101-127 / central_cropping :129-133)."""
    n, h, w, c = x.shape
    if flip_td:
        x = x[:, ::-1]
    if flip_lr:
        x = x[:, :, ::-1]
    crop = np.ascontiguousarray(x[:, y0:y0 + ch, x0:x0 + cw])
    return resize_nearest_legacy(crop, h, w) if nearest else resize_bilinear_legacy(crop, h, w)


def pair_table(seq_lengths, t_len, training):
    """The (frame index, direction) table of image_inputs (:196-214) / test_inputs (:252-276): forward pairs from the first
    frames, backward pairs from the last ones, over the concatenated sequences."""
    first, last, n = [], [], 0
    for ln in seq_lengths:
        if training:
            last.append(np.arange(n + t_len, n + ln))
            first.append(np.arange(n, n + ln - t_len))
        elif t_len < 0:
            last.append(np.arange(n + abs(t_len), n + ln))
            first.append(np.arange(n, n + abs(t_len)))
        else:
            first.append(np.arange(n, n + ln - t_len))
            last.append(np.arange(n + ln - t_len, n + ln))
        n += ln
    first, last = np.concatenate(first), np.concatenate(last)
    return np.vstack([np.stack([first, np.ones_like(first)], 1), np.stack([last, -np.ones_like(last)], 1)]).astype(np.float32)
