"""Post-processing stage of the reference on the GPU ("next" row N4 of SURVEY.md section 8f), same function names and argument
meaning as post_processing/generate_soft_score_from_buffer.py and post_processing/crf_refine.py:

  sanity_check(s) / rectify_pred_mask(pred_mask, crop, H, W)      generate_soft_score_from_buffer.py:116-125, :98-114
  soft_score(...)   the per-frame body of buffer_to_soft_score   :38-93  (device tensors in, device tensor out: no .mat round trip)
  propagate(...)    flow-guided moving average of the masks       :127-231 (pyflow.so -> the path's own PWC-Net flow; cv2.remap -> udet_post_remap)
  refine(...) / select_candidate(...)                             crf_refine.py:110-138, :40-50
  buffer_to_soft_score(buffer_path, out_path, ...) / run_crf(...)  the file-level drivers over the .mat buffers the ensemble run writes

The kernels of the stage -- border statistics, bytescale, Pillow's 8-bit resampler, canvas placement, min-max / max
normalisations, remap, blending, the separable Gaussian and the dense-CRF mean field -- run in libudet.so (csrc/postproc.hip); torch
holds the device memory and does a few elementwise glue steps (adding two score maps, clamp / log of the unary).  What the reference
delegates to third-party code that is not in its tree (scipy.misc.imresize, cv2.remap, pyflow, pydensecrf) is restated from the
published algorithms -- see oracle/oracle_post.py for the restatement the kernels are tested against and for what could and could
not be pinned to the original libraries."""
from __future__ import annotations

import ctypes
import math
import os

import numpy as np
import torch

from ._ffi import c_f, c_i, c_p, c_sz, check, lib

for _n, _a in (("udet_post_border_mean", [c_p, c_i, c_i, c_i, c_p, c_p]),
               ("udet_post_bytescale", [c_p, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
               ("udet_post_resample_u8", [c_p, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_i, c_i, c_p]),
               ("udet_post_place", [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
               ("udet_post_minmax_norm", [c_p, c_i, c_p, c_p]),
               ("udet_post_remap", [c_p, c_p, c_p, c_i, c_i, c_p]),
               ("udet_post_blend", [c_p, c_f, c_p, c_f, c_i, c_i, c_p]),
               ("udet_post_gauss1d", [c_p, c_p, c_i, c_i, c_p, c_i, c_i, c_p]),
               ("udet_post_dense_crf", [c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_i, c_i, c_p, c_p, c_sz, c_p])):
    getattr(lib, _n).restype = c_i
    getattr(lib, _n).argtypes = _a
lib.udet_post_crf_workspace_bytes.restype = c_sz
lib.udet_post_crf_workspace_bytes.argtypes = [c_i, c_i]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _dev(x, dtype):
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x))
    return t.to("cuda", dtype).contiguous()


# ---------------------------------------------------------------------------------------------- Pillow coefficients ----
_coeff_cache = {}


def _pil_coeffs(in_size: int, out_size: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (host side, exactly Pillow's double
    arithmetic): device int32 tables kk [out][ksize] and bounds [out][2]."""
    key = (in_size, out_size)
    if key not in _coeff_cache:
        scale = filterscale = in_size / out_size
        if filterscale < 1.0:
            filterscale = 1.0
        support = 1.0 * filterscale
        ksize = int(math.ceil(support)) * 2 + 1
        kk = np.zeros((out_size, ksize), np.int32)
        bounds = np.zeros((out_size, 2), np.int32)
        ss = 1.0 / filterscale
        for xx in range(out_size):
            center = (xx + 0.5) * scale
            xmin = max(int(center - support + 0.5), 0)
            xmax = min(int(center + support + 0.5), in_size) - xmin
            w = [0.0] * ksize
            ww = 0.0
            for x in range(xmax):
                v = abs((x + xmin - center + 0.5) * ss)
                w[x] = 1.0 - v if v < 1.0 else 0.0
                ww += w[x]
            for x in range(ksize):
                v = w[x] / ww if (x < xmax and ww != 0.0) else w[x]
                kk[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
            bounds[xx] = (xmin, xmax)
        _coeff_cache[key] = (torch.from_numpy(kk).cuda(), torch.from_numpy(bounds).cuda(), ksize)
    return _coeff_cache[key]


def _imresize_window(src64: torch.Tensor, y0, x0, h, w, out_h, out_w) -> torch.Tensor:
    """scipy.misc.imresize(src[y0:y0+h, x0:x0+w], (out_h, out_w)) -> uint8 device tensor [out_h, out_w]."""
    H, W = src64.shape
    u8 = torch.empty((h, w), dtype=torch.uint8, device=src64.device)
    check(lib.udet_post_bytescale(src64.data_ptr(), W, y0, x0, h, w, u8.data_ptr(), _stream()))
    cur, ch, cw = u8, h, w
    if out_w != cw:
        kk, bounds, ks = _pil_coeffs(cw, out_w)
        nxt = torch.empty((ch, out_w), dtype=torch.uint8, device=src64.device)
        check(lib.udet_post_resample_u8(cur.data_ptr(), ch, cw, nxt.data_ptr(), ch, out_w, kk.data_ptr(), bounds.data_ptr(), ks, 1, _stream()))
        cur, cw = nxt, out_w
    if out_h != ch:
        kk, bounds, ks = _pil_coeffs(ch, out_h)
        nxt = torch.empty((out_h, cw), dtype=torch.uint8, device=src64.device)
        check(lib.udet_post_resample_u8(cur.data_ptr(), ch, cw, nxt.data_ptr(), out_h, cw, kk.data_ptr(), bounds.data_ptr(), ks, 0, _stream()))
        cur = nxt
    return cur


# ------------------------------------------------------------------------------------------------------ soft score ----
def sanity_check(s):
    """mean over the four two-pixel border strips (generate_soft_score_from_buffer.py:116-125); s: [H,W] or [N,H,W] -> float / array."""
    t = _dev(s, torch.float32)
    single = t.dim() == 2
    t = t.view(-1, t.shape[-2], t.shape[-1])
    out = torch.empty(t.shape[0], dtype=torch.float64, device=t.device)
    check(lib.udet_post_border_mean(t.data_ptr(), t.shape[0], t.shape[1], t.shape[2], out.data_ptr(), _stream()))
    v = out.cpu().numpy()
    return float(v[0]) if single else v


def rectify_pred_mask(pred_mask, crop, H, W):
    """Bring a prediction made on another central crop back to the base crop (:98-114) -> device float64 [H,W]."""
    p = _dev(pred_mask, torch.float64)
    canvas = torch.empty((H, W), dtype=torch.float64, device=p.device)
    if crop > 1:
        crop = 1.0 / crop
        hh, ww = int(H * crop), int(W * crop)
        h, w = int((H - hh) / 2), int((W - ww) / 2)
        patch = _imresize_window(p, h, w, hh, ww, H, W)
        check(lib.udet_post_place(patch.data_ptr(), H, W, 0, 0, H, W, canvas.data_ptr(), _stream()))
    else:
        hh, ww = int(H * crop), int(W * crop)
        patch = _imresize_window(p, 0, 0, p.shape[0], p.shape[1], hh, ww)
        h, w = max(int((H - hh) / 2), 0), max(int((W - ww) / 2), 0)
        check(lib.udet_post_place(patch.data_ptr(), hh, ww, h, w, H, W, canvas.data_ptr(), _stream()))
    return canvas


def soft_score(preds_b, preds_f, crops=(85, 90, 95, 100), base_crop=90.0, base_hw=(192, 384), san_t=0.6):
    """The per-frame body of buffer_to_soft_score (:38-93).  preds_b / preds_f: [shift-1][crop index] soft masks [H,W] (device
    tensors or arrays) of the backward (-shift) / forward (+shift) ensemble runs -> pred_mask, device float64 [H,W]."""
    H, W = base_hw
    ns, nc = len(preds_b), len(crops)
    allm = torch.stack([_dev(np.squeeze(m) if not isinstance(m, torch.Tensor) else m.squeeze(), torch.float32)
                        for grp in (preds_b, preds_f) for row in grp for m in row], 0)
    sani = sanity_check(allm).reshape(2, ns, nc)  # one launch for every mask of the frame
    score = None
    for si in range(ns):
        for ci, crop in enumerate(crops):
            s_b, s_f = allm[si * nc + ci], allm[(ns + si) * nc + ci]
            bad_b, bad_f = sani[0, si, ci] >= san_t, sani[1, si, ci] >= san_t
            if bad_b and bad_f:
                s_b, s_f = torch.zeros_like(s_b), torch.zeros_like(s_f)
            elif bad_b:
                s_b = s_f
            elif bad_f:
                s_f = s_b
            if si == 0 and crop == base_crop:
                term = s_b.double() + s_f.double()
            else:
                ratio = crop / base_crop
                term = rectify_pred_mask(s_b, ratio, H, W) + rectify_pred_mask(s_f, ratio, H, W)
            score = term if score is None else score + term
    out = torch.empty_like(score)
    check(lib.udet_post_minmax_norm(score.data_ptr(), score.numel(), out.data_ptr(), _stream()))
    return out


# ------------------------------------------------------------------------------------------------------ propagation ----
def remap(src, flow_uv):
    """cv2.remap(src, flow_uv + pixel grid, None, cv2.INTER_LINEAR) (:171-176): src [H,W] float32, flow_uv [H,W,2] = (u, v)."""
    s, f = _dev(src, torch.float32), _dev(flow_uv, torch.float32)
    H, W = s.shape
    out = torch.empty_like(s)
    check(lib.udet_post_remap(s.data_ptr(), f.data_ptr(), out.data_ptr(), H, W, _stream()))
    return out


def propagate_step(running_avg, s_prev, flow_uv, w_r=0.85):
    """One step of propagate (:166-185): pull the previous mask and the running average along flow_uv to the current frame and
    blend them; every intermediate is max-normalised like the reference.  Returns the new running average (device float32)."""
    s2 = remap(s_prev, flow_uv)
    ra_w = remap(running_avg, flow_uv)
    ra = torch.empty_like(ra_w)
    n = ra.numel()
    check(lib.udet_post_blend(ra_w.data_ptr(), 1.0, ra.data_ptr(), 0.0, n, 0, _stream()))                 # ra = ra_w / (max + 1e-8)
    check(lib.udet_post_blend(s2.data_ptr(), float(np.float32(1 - w_r)), ra.data_ptr(), float(np.float32(w_r)), n, 1, _stream()))
    return ra


class PWCFlow(object):
    """Replacement of pyflow.coarse2fine_flow in propagate(): the path's own PWC-Net.  flow(I_from, I_to) returns
    (u, v) [H,W,2] such that I_from(x, y) ~ I_to(x + u, y + v) -- the convention of the call at :158-160."""

    def __init__(self, model=None):
        from .functional import ModelPWCNet
        self.model = model or ModelPWCNet()

    def __call__(self, img_from_u8, img_to_u8):
        a = _dev(img_from_u8, torch.float32) / 255.0 - 0.5
        b = _dev(img_to_u8, torch.float32) / 255.0 - 0.5
        # the trained network's channel 0 is u (the x displacement of the ground-truth flows it was fitted to), channel 1 is v:
        # the order pyflow returns and cv2.remap's map expects (:162-165)
        return self.model.predict_from_img_pairs(a.unsqueeze(0).contiguous(), b.unsqueeze(0).contiguous())[0].contiguous()


def propagate(pred_masks, images_u8, flow_fn, w_r=0.85):
    """Moving average of a sequence's masks along the optical flow, forward and backward (:127-231).  pred_masks: list of [H,W]
    soft masks, images_u8: list of [H,W,3] uint8 frames, flow_fn(I_a, I_b) -> (u, v) field used as in the reference's calls
    (forward pass: flow_fn(I_k, I_{k-1}); backward pass: flow_fn(I_k, I_{k+1})).  Returns (running_avg_f, running_avg_b) lists."""
    n = len(pred_masks)
    masks = [_dev(np.squeeze(m) if not isinstance(m, torch.Tensor) else m.squeeze(), torch.float32) for m in pred_masks]
    fwd, bwd = [None] * n, [None] * n
    ra = masks[0]
    fwd[0] = ra
    for k in range(1, n):
        ra = propagate_step(ra, masks[k - 1], flow_fn(images_u8[k], images_u8[k - 1]), w_r)
        fwd[k] = ra
    ra = masks[n - 1]
    bwd[n - 1] = ra
    for k in range(n - 2, -1, -1):
        ra = propagate_step(ra, masks[k + 1], flow_fn(images_u8[k], images_u8[k + 1]), w_r)
        bwd[k] = ra
    return fwd, bwd


# --------------------------------------------------------------------------------------------------------------- CRF ----
def select_candidate(pred_mask, pred_f, pred_b, gt_mask):
    """crf_refine.py:40-50: the candidate with the largest object score sum(p * gt) / (sum(p) + 1e-8)."""
    gt = _dev(gt_mask, torch.float32)

    def objscore(p):
        p = _dev(p, torch.float32)
        return float((p * gt).sum() / (p.sum() + 1e-8))
    m, f, b = objscore(pred_mask), objscore(pred_f), objscore(pred_b)
    if m >= f and m >= b:
        return pred_mask, 0
    if f >= m and f >= b:
        return pred_f, 1
    return pred_b, 2


def gaussian_filter(x, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter(x, sigma) (mode='reflect') on the device, float64."""
    t = _dev(x, torch.float64)
    r = int(truncate * float(sigma) + 0.5)
    if r == 0:
        return t.clone()
    k = np.exp(-0.5 / (sigma * sigma) * np.arange(-r, r + 1) ** 2)
    k = torch.from_numpy(k / k.sum()).cuda()
    H, W = t.shape
    for ax in (0, 1):
        out = torch.empty_like(t)
        check(lib.udet_post_gauss1d(t.data_ptr(), out.data_ptr(), H, W, k.data_ptr(), r, ax, _stream()))
        t = out
    return t


def dense_crf(unary, image_u8, sxy, srgb, compat, iters=50, radius=None):
    """DenseCRF2D + setUnaryEnergy + addPairwiseBilateral + inference(iters) (crf_refine.py:111-130) -> Q device float32 [2,H,W]."""
    un = _dev(unary, torch.float32)
    img = _dev(image_u8, torch.uint8)
    _, H, W = un.shape
    R = int(math.ceil(3.0 * sxy)) if radius is None else int(radius)
    q = torch.empty_like(un)
    ws = torch.empty(int(lib.udet_post_crf_workspace_bytes(H, W)), dtype=torch.uint8, device=un.device)
    check(lib.udet_post_dense_crf(un.data_ptr(), img.data_ptr(), H, W, sxy, srgb, compat, iters, R, q.data_ptr(), ws.data_ptr(), ws.numel(),
                                  _stream()))
    return q


def refine(mask, image, gk, sxy, srgb, compat, gtmask, iters=50, radius=None):
    """crf_refine.py:110-138 -> (new_mask numpy float32 [H,W] in {0,1}, IoU against gt > 0.1)."""
    U = gaussian_filter(mask, gk)
    U = U / (U.max() + 1e-8)
    U = torch.clamp(U, 1e-6, 1.0 - 1e-6)
    unary = (-torch.log(torch.stack([1.0 - U, U], 0))).float()
    Q = dense_crf(unary, image, sxy, srgb, compat, iters, radius)
    new_mask = (Q[1] > Q[0]).float().cpu().numpy()  # np.argmax(Q, axis=0): label 1 only where strictly larger
    gt, bm = np.asarray(gtmask) > 0.1, new_mask > 0.1
    return new_mask, np.float32(np.sum(gt & bm)) / np.float32(np.sum(gt | bm))


# ----------------------------------------------------------------------------------------------------- file drivers ----
def buffer_to_soft_score(buffer_path, out_path, seq_names, seq_num, max_shift=2, base_crop=90.0, dprefix="davis_shift", flow_fn=None):
    """generate_soft_score_from_buffer.buffer_to_soft_score over the result_<k>.mat buffers test_generator_ensemble writes
    (evaluation.evaluate_ensemble), followed by propagate(); writes result_<k>.mat with pred_mask / img1 / gt_mask /
    running_avg_f / running_avg_b like the reference (:91-93, :148, :184, :199, :229)."""
    import scipy.io as sio
    crops = list(range(85, 101, 5))
    for name, num in zip(seq_names, seq_num):
        out_dir = os.path.join(out_path, name)
        os.makedirs(out_dir, exist_ok=True)
        print(out_dir)
        masks, imgs, gts = [], [], []
        for k in range(1, num + 1):
            pb, pf, r_f1 = [], [], None
            for shift in range(1, max_shift + 1):
                r_b = sio.loadmat(os.path.join(buffer_path, "%s_%d" % (dprefix, -shift), name, "result_%d.mat" % k))
                r_f = sio.loadmat(os.path.join(buffer_path, "%s_%d" % (dprefix, shift), name, "result_%d.mat" % k))
                pb.append([np.squeeze(r_b["pred_mask_%03d" % c]) for c in crops])
                pf.append([np.squeeze(r_f["pred_mask_%03d" % c]) for c in crops])
                if shift == 1:
                    r_f1 = r_f
            masks.append(soft_score(pb, pf, crops, base_crop))
            imgs.append(((r_f1["img_1_%03d" % int(base_crop)] + 0.5) * 255).astype("uint8"))
            gts.append(r_f1["gt_mask_%03d" % int(base_crop)])
        fwd, bwd = propagate(masks, imgs, flow_fn or PWCFlow())
        for k in range(num):
            sio.savemat(os.path.join(out_dir, "result_%d.mat" % (k + 1)),
                        {"pred_mask": masks[k].cpu().numpy(), "img1": imgs[k], "gt_mask": gts[k],
                         "running_avg_f": fwd[k].cpu().numpy(), "running_avg_b": bwd[k].cpu().numpy()})


def run_crf(path_soft, sxy, srgb, scomp, gauss_k, out_path="./post_processed_davis"):
    """crf_refine.run_crf (:9-59) over the soft-score folder; returns the average IoU."""
    import scipy.io as sio
    sum_iou, total = 0.0, 0.0
    for seq in os.listdir(path_soft):
        seq_path = os.path.join(path_soft, seq)
        seq_len = len([n for n in os.listdir(seq_path) if n.endswith(".mat")])
        out_dir = os.path.join(out_path, seq)
        os.makedirs(out_dir, exist_ok=True)
        print(out_dir)
        for k in range(seq_len):
            result = sio.loadmat(os.path.join(seq_path, "result_%d.mat" % (k + 1)))
            total += 1.0
            pm, pf, pb = (np.float32(np.squeeze(result[n])) for n in ("pred_mask", "running_avg_f", "running_avg_b"))
            gt = np.float32(np.squeeze(result["gt_mask"]))
            mask, _ = select_candidate(pm, pf, pb, gt)
            new_mask, iou = refine(mask, result["img1"], gauss_k, sxy, srgb, scomp, gt)
            sio.savemat(os.path.join(out_dir, "result_%d.mat" % (k + 1)), {"gt_mask": gt, "soft_mask": mask, "mask": new_mask})
            sum_iou += iou
    return sum_iou / total
