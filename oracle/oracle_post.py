"""CPU restatement (numpy, float64 where the reference is) of the offline post-processing stage, SURVEY.md section 8f row N4:

  post_processing/generate_soft_score_from_buffer.py   sanity_check :116-125, rectify_pred_mask :98-114,
                                                        buffer_to_soft_score :16-96 (score accumulation + min-max), propagate :127-231
  post_processing/crf_refine.py                         candidate selection :40-50, refine :110-138

TEST INFRASTRUCTURE ONLY (imported by tests/ -- never by the product path).

The stage rests on third-party code that is absent from /root/reference and not installable here; each is restated from its
published algorithm and the reference's call sites fix the arguments:
  * scipy.misc.imresize (scipy <= 1.2, removed in 1.3; the reference is a py2.7 / scipy-1.x script): bytescale to uint8 with the
    array's own min / max, PIL `Image.resize(..., BILINEAR)`, back to a uint8 array.  Pillow IS installed: `imresize` below calls
    the real `Image.resize`, and `pil_bilinear_u8` (the restatement of Pillow's 8-bit resampler, Resample.c: double-precision
    triangle-filter coefficients with the support widened by the down-scaling factor, 22-bit fixed point, horizontal pass then
    vertical pass) is held to it bit for bit in tests/test_post_processing.py -- that restatement is what the HIP kernels follow.
  * cv2.remap(src, map, None, INTER_LINEAR) (OpenCV 3.x/4.x, imgwarp.cpp remapBilinear): coordinates rounded to 1/32 pixel
    (cvRound(x * 32), round-half-even), 4-tap float weights from the 32x32 bilinear table, BORDER_CONSTANT 0 outside.  OpenCV is
    absent: PARITY UNPINNED against the library (the restatement is checked against hand-computed cases).
  * pyflow.coarse2fine_flow (Ce Liu's variational flow, shipped as a py2.7 binary): NOT restated -- SURVEY 8f replaces it by the
    path's own PWC-Net flow; `propagate` takes the flow as an argument.
  * pydensecrf DenseCRF2D (Kraehenbuehl & Koltun 2011): mean-field inference with one bilateral Potts term, symmetric kernel
    normalisation.  The library evaluates the Gaussian kernel approximately on a permutohedral lattice; here it is evaluated
    exactly (dense, truncated at `radius`): PARITY UNPINNED against the library, the algorithm is the paper's.
"""
from __future__ import annotations

import math

import numpy as np


# ------------------------------------------------------------------------------------------------ sanity / rectify ----
def sanity_check(s):
    """generate_soft_score_from_buffer.py:116-125: mean over the four two-pixel border strips (corners counted twice)."""
    H, W = s.shape
    a, b, c, d = s[0:2, :], s[H - 2:H, :], s[:, 0:2], s[:, W - 2:W]
    return (np.sum(a) + np.sum(b) + np.sum(c) + np.sum(d)) / (1.0 * (a.size + b.size + c.size + d.size))


def bytescale(data):
    """scipy.misc.bytescale(data) as toimage() calls it for a float array (scipy 1.2 pilutil.py:33-102): cmin / cmax = the
    array's min / max, low = 0, high = 255."""
    data = np.asarray(data)
    if data.dtype == np.uint8:
        return data
    cmin, cmax = data.min(), data.max()
    cscale = cmax - cmin
    if cscale == 0:
        cscale = 1
    scale = 255.0 / cscale
    bytedata = (data - cmin) * scale + 0
    return (bytedata.clip(0, 255) + 0.5).astype(np.uint8)


def _coeffs(in_size, out_size):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1.0) over the whole axis."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((out_size, ksize), np.int64)
    bounds = np.zeros((out_size, 2), np.int64)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        ss = 1.0 / filterscale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize)
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            v = -v if v < 0 else v
            w[x] = 1.0 - v if v < 1.0 else 0.0
        ww = w[:xmax].sum() if xmax else 0.0
        # Pillow accumulates ww in loop order; np.sum may pair differently -- redo it sequentially
        ww = 0.0
        for x in range(xmax):
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << 22)) if w[x] < 0 else int(0.5 + w[x] * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return kk, bounds, ksize


def pil_bilinear_u8(img_u8, out_h, out_w):
    """Pillow's Image.resize((out_w, out_h), BILINEAR) on a uint8 'L' image: horizontal pass, then vertical pass, each
    ss = 1 << 21; ss += pixel * k; out = clip8(ss >> 22)."""
    img = np.asarray(img_u8, np.int64)
    H, W = img.shape

    def resample_rows(src, n_out, kk, bounds):  # along axis 1
        out = np.zeros((src.shape[0], n_out), np.int64)
        for xx in range(n_out):
            xmin, xmax = bounds[xx]
            ss = np.full(src.shape[0], 1 << 21, np.int64)
            for x in range(xmax):
                ss += src[:, xmin + x] * kk[xx, x]
            out[:, xx] = np.clip(ss >> 22, 0, 255)
        return out
    tmp = img
    if out_w != W:
        kk, bounds, _ = _coeffs(W, out_w)
        tmp = resample_rows(img, out_w, kk, bounds)
    if out_h != H:
        kk, bounds, _ = _coeffs(H, out_h)
        tmp = resample_rows(tmp.T, out_h, kk, bounds).T
    return tmp.astype(np.uint8)


def imresize(arr, size):
    """scipy.misc.imresize(arr, (H, W)) (interp='bilinear', mode=None) with the REAL Pillow resize."""
    from PIL import Image
    im = Image.fromarray(bytescale(arr), mode="L") if False else Image.fromarray(bytescale(arr))
    return np.asarray(im.resize((size[1], size[0]), resample=Image.BILINEAR))


def rectify_pred_mask(pred_mask, crop, H, W, resize=imresize):
    """generate_soft_score_from_buffer.py:98-114: bring a prediction made on another central crop back to the base crop."""
    if crop > 1:
        crop = 1.0 / crop
        hh, ww = int(H * crop), int(W * crop)
        h, w = int((H - hh) / 2), int((W - ww) / 2)
        rec = resize(pred_mask[h:h + hh, w:w + ww], (H, W)).astype(np.float64)
    else:
        rec = np.zeros((H, W))
        hh, ww = int(H * crop), int(W * crop)
        pc = resize(pred_mask, (hh, ww))
        h, w = max(int((H - hh) / 2), 0), max(int((W - ww) / 2), 0)
        rec[h:h + hh, w:w + ww] = pc
    return rec / (np.amax(rec) + 1e-6)


def soft_score(preds_b, preds_f, crops=(85, 90, 95, 100), base_crop=90.0, base_hw=(192, 384), san_t=0.6, resize=imresize):
    """The per-frame body of buffer_to_soft_score (:38-93).  preds_b / preds_f: [shift-1][crop index] soft masks [H, W] of the
    backward (-shift) and forward (+shift) ensemble runs.  Returns pred_mask = min-max normalised score."""
    H, W = base_hw
    score = None
    for si in range(len(preds_b)):
        shift = si + 1
        for ci, crop in enumerate(crops):
            s_b, s_f = np.squeeze(preds_b[si][ci]).astype(np.float64), np.squeeze(preds_f[si][ci]).astype(np.float64)
            sani_b, sani_f = sanity_check(s_b), sanity_check(s_f)
            if sani_b >= san_t and sani_f >= san_t:
                s_b, s_f = s_b * 0.0, s_f * 0.0
            elif sani_b >= san_t and sani_f < san_t:
                s_b = s_f
            elif sani_b < san_t and sani_f >= san_t:
                s_f = s_b
            if shift == 1 and crop == base_crop:
                term = s_b + s_f
            else:
                ratio = crop / base_crop
                term = rectify_pred_mask(s_b, ratio, H, W, resize) + rectify_pred_mask(s_f, ratio, H, W, resize)
            score = term if score is None else score + term
    mn, mx = np.amin(score), np.amax(score)
    return (score - mn) / (mx - mn + 1e-6)


# ------------------------------------------------------------------------------------------------------- remap ----
def remap_bilinear(src, map_xy):
    """cv2.remap(src, map_xy, None, cv2.INTER_LINEAR) for a float32 single-channel src and an absolute float32 map [H,W,2]
    (x, y): fixed-point coordinates (1/32 px, round-half-even), float 4-tap weights, constant 0 border."""
    src = np.asarray(src, np.float32)
    H, W = src.shape
    m = np.asarray(map_xy, np.float32)
    sx = np.rint(m[..., 0].astype(np.float64) * 32).astype(np.int64)
    sy = np.rint(m[..., 1].astype(np.float64) * 32).astype(np.int64)
    ix, iy, fx, fy = sx >> 5, sy >> 5, (sx & 31).astype(np.float32) / np.float32(32), (sy & 31).astype(np.float32) / np.float32(32)
    ix, iy = np.clip(ix, -32768, 32767), np.clip(iy, -32768, 32767)  # saturate_cast<short>
    w00, w01 = (np.float32(1) - fy) * (np.float32(1) - fx), (np.float32(1) - fy) * fx
    w10, w11 = fy * (np.float32(1) - fx), fy * fx

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        return np.where(ok, src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)], np.float32(0))
    out = tap(iy, ix) * w00
    out = out + tap(iy, ix + 1) * w01
    out = out + tap(iy + 1, ix) * w10
    out = out + tap(iy + 1, ix + 1) * w11
    return out.astype(np.float32)


def propagate_step(running_avg, s_prev, flow_uv, w_r=0.85):
    """One step of propagate (:166-185 / :215-231): flow_uv [H,W,2] = (u, v) from the current frame to the previous one;
    the previous mask and the running average are pulled to the current frame and blended."""
    h, w = flow_uv.shape[:2]
    m = flow_uv.astype(np.float64).copy()
    m[:, :, 0] += np.arange(w)
    m[:, :, 1] += np.arange(h)[:, np.newaxis]
    m = m.astype(np.float32)
    s2 = remap_bilinear(np.asarray(s_prev, np.float32), m)
    s2 = s2 / (np.amax(s2) + 1e-8)
    ra = remap_bilinear(np.asarray(running_avg, np.float32), m)
    ra = ra / (np.amax(ra) + 1e-8)
    ra = (1 - w_r) * s2 + w_r * ra
    return ra / (np.amax(ra) + 1e-8)


# --------------------------------------------------------------------------------------------------------- CRF ----
def select_candidate(pred_mask, pred_f, pred_b, gt_mask):
    """crf_refine.py:40-50: the candidate (raw / forward-propagated / backward-propagated) with the largest object score."""
    def objscore(p):
        return np.sum(np.multiply(p, gt_mask)) / (np.sum(p) + 1e-8)
    m, f, b = objscore(pred_mask), objscore(pred_f), objscore(pred_b)
    if m >= f and m >= b:
        return pred_mask, 0
    if f >= m and f >= b:
        return pred_f, 1
    return pred_b, 2


def gaussian_filter(x, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(x, sigma) (mode='reflect'): separable, radius int(truncate*sigma + 0.5)."""
    x = np.asarray(x, np.float64)
    r = int(truncate * float(sigma) + 0.5)
    if r == 0:
        return x.copy()
    k = np.exp(-0.5 / (sigma * sigma) * np.arange(-r, r + 1) ** 2)
    k /= k.sum()
    for ax in (0, 1):
        pad = [(r, r) if a == ax else (0, 0) for a in (0, 1)]
        xp = np.pad(x, pad, mode="symmetric")
        x = sum(k[i] * np.take(xp, np.arange(i, i + x.shape[ax]), axis=ax) for i in range(2 * r + 1))
    return x


def unary_from_mask(mask, gk):
    """crf_refine.py:113-121: U = gaussian(mask) / max, clipped to [1e-6, 1-1e-6]; energies -log([1-U, U]) as float32 [2,H,W]."""
    U = gaussian_filter(mask, gk)
    U = U / (np.amax(U) + 1e-8)
    U = np.clip(U, 1e-6, 1.0 - 1e-6)
    return np.float32(-np.log(np.stack([1.0 - U, U], 0)))


def dense_crf(unary, image_u8, sxy, srgb, compat, iters=50, radius=None):
    """DenseCRF2D(W, H, 2) + setUnaryEnergy + addPairwiseBilateral(sxy, srgb, rgbim, compat) + inference(iters)
    (Kraehenbuehl & Koltun 2011, Algorithm 1) with the Gaussian bilateral kernel evaluated exactly inside a (2*radius+1)^2
    window (radius default ceil(3*sxy)), kernel k(i,i) excluded, symmetric normalisation n_i = 1/sqrt(sum_j k_ij):
        Q <- softmax(-unary);  repeat: Q <- softmax(-unary + compat * n * K (n * Q)).   Returns Q [2,H,W] float32."""
    un = np.asarray(unary, np.float32)
    _, H, W = un.shape
    img = np.asarray(image_u8, np.float32)
    R = int(math.ceil(3.0 * sxy)) if radius is None else int(radius)
    ys, xs = np.mgrid[0:H, 0:W]

    def apply_kernel(f):  # sum_j k_ij f_j, f [C,H,W]
        out = np.zeros_like(f)
        for dy in range(-R, R + 1):
            for dx in range(-R, R + 1):
                if dy == 0 and dx == 0:
                    continue
                y2, x2 = ys + dy, xs + dx
                ok = (y2 >= 0) & (y2 < H) & (x2 >= 0) & (x2 < W)
                y2c, x2c = np.clip(y2, 0, H - 1), np.clip(x2, 0, W - 1)
                dI = img - img[y2c, x2c]
                k = np.exp(np.float32(-0.5) * (np.float32(dy * dy + dx * dx) / np.float32(sxy * sxy) + (dI * dI).sum(-1) / np.float32(srgb * srgb)))
                out += np.where(ok, k, np.float32(0))[None] * f[:, y2c, x2c]
        return out
    norm = np.float32(1) / np.sqrt(apply_kernel(np.ones((1, H, W), np.float32))[0] + np.float32(1e-20))

    def softmax(e):
        e = e - e.max(0, keepdims=True)
        p = np.exp(e)
        return p / p.sum(0, keepdims=True)
    Q = softmax(-un)
    for _ in range(iters):
        msg = norm[None] * apply_kernel(Q * norm[None])
        Q = softmax(-un + np.float32(compat) * msg)
    return Q.astype(np.float32)


def refine(mask, image, gk, sxy, srgb, compat, gtmask, iters=50, radius=None):
    """crf_refine.py:110-138 -> (new_mask [H,W] in {0,1}, IoU against gt > 0.1)."""
    Q = dense_crf(unary_from_mask(mask, gk), image, sxy, srgb, compat, iters, radius)
    new_mask = np.float32(np.argmax(Q, axis=0))
    gt, bm = gtmask > 0.1, new_mask > 0.1
    return new_mask, np.float32(np.sum(gt & bm)) / np.float32(np.sum(gt | bm))
