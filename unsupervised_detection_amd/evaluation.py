"""Evaluation tail of the hot path ("next" row N2 of SURVEY.md section 8f), backed by udet_mask_stats.

Mirrors, with the reference's names and return contracts:
  compute_boundary_score(segmentation)                  models/utils/general_utils.py:122-138
  disambiguate_forw_back(pred_masks, threshold=0.1)     models/utils/general_utils.py:100-110
  compute_all_IoU(pred_masks, gt_masks, threshold=0.1)  models/utils/general_utils.py:112-116 (+ tf_iou_computation :89-98)
  compute_IoU(gt_mask, pred_mask_f, threshold=0.1)      test_generator.py:19-35
  compute_mae(gt_mask, pred_mask_f)                     test_generator.py:38-40
  evaluate_masks(learner, ...)                          the aggregation / report of test_generator.py:43-130

The batch functions take device tensors [B,H,W,1]; one kernel pass produces every per-sample sum both IoU variants and
the MAE need (exact counts, double accumulation), the few scalar operations that remain run on the host."""
from __future__ import annotations


import numpy as np
import torch

from ._ffi import c_f, c_i, c_p, check, lib

lib.udet_mask_stats.restype = c_i
lib.udet_mask_stats.argtypes = [c_p, c_p, c_i, c_i, c_i, c_f, c_f, c_p, c_p]

MASK_THRESHOLD = 0.6  # test_generator.py:16 / general_utils.py:101


def mask_stats(pred_masks: torch.Tensor, gt_masks: torch.Tensor, threshold: float = 0.1, gt_threshold: float = 0.0) -> np.ndarray:
    """[B,8] float64: border sum, |pred|, |gt|, |pred&gt|, sum pred|gt-1|, sum (1-pred)|gt|, sum (1-pred)|gt-1|, sum pred|gt|."""
    for t, name in ((pred_masks, "pred_masks"), (gt_masks, "gt_masks")):
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise ValueError(f"{name} must be a contiguous float32 CUDA(HIP) tensor")
    if pred_masks.shape != gt_masks.shape or pred_masks.dim() != 4 or pred_masks.shape[-1] != 1:
        raise ValueError("masks must both be [B,H,W,1]")
    b, h, w, _ = pred_masks.shape
    out = torch.empty((b, 8), dtype=torch.float64, device=pred_masks.device)
    check(lib.udet_mask_stats(pred_masks.data_ptr(), gt_masks.data_ptr(), b, h, w, threshold, gt_threshold, out.data_ptr(),
                              torch.cuda.current_stream().cuda_stream))
    return out.cpu().numpy()


def _border_score(stats, h, w):
    return stats[:, 0] / float(2 * 2 * w + 2 * 2 * h)


def compute_boundary_score(segmentation) -> float:
    """Fraction of the four 2-pixel image borders the (boolean / 0-1) mask covers; >= 0.6 means background."""
    seg = torch.as_tensor(np.asarray(segmentation, dtype=np.float32)).reshape(1, segmentation.shape[0], segmentation.shape[1], 1)
    st = mask_stats(seg.cuda().contiguous(), torch.zeros_like(seg).cuda(), threshold=0.5)
    return float(_border_score(st, seg.shape[1], seg.shape[2])[0])


def disambiguate_forw_back(pred_masks: torch.Tensor, threshold: float = 0.1) -> torch.Tensor:
    """Binary masks, complemented per sample when they cover the image borders (score >= 0.6)."""
    b, h, w, _ = pred_masks.shape
    st = mask_stats(pred_masks, torch.zeros_like(pred_masks), threshold)
    fg = torch.as_tensor(_border_score(st, h, w) < MASK_THRESHOLD, device=pred_masks.device).view(-1, 1, 1, 1)
    binm = (pred_masks > threshold).to(torch.float32)
    return torch.where(fg, binm, 1.0 - binm)


def _iou_terms(st, hw, flip):
    n_pred, n_gt, inter = st[:, 1], st[:, 2], st[:, 3]
    inter_c, n_pred_c = n_gt - inter, hw - n_pred
    i = np.where(flip, inter_c, inter)
    u = np.where(flip, n_pred_c + n_gt - inter_c, n_pred + n_gt - inter)
    ann = np.where(flip, n_pred_c, n_pred)
    return i, u, ann


def compute_all_IoU(pred_masks: torch.Tensor, gt_masks: torch.Tensor, threshold: float = 0.1) -> np.ndarray:
    """The validation IoU of the training graph (adversarial_learner.py:135-139): gt > 0.01, |and| / (|or| + 1e-8)."""
    b, h, w, _ = pred_masks.shape
    st = mask_stats(pred_masks, gt_masks, threshold, 0.01)
    flip = _border_score(st, h, w) >= MASK_THRESHOLD
    i, u, _ = _iou_terms(st, float(h * w), flip)
    return i / (u + 1e-8)


def evaluate_batch(gt_masks: torch.Tensor, pred_masks: torch.Tensor, threshold: float = 0.1):
    """Per sample (iou, mae, flipped) with the semantics of test_generator.py compute_IoU / compute_mae:
    gt cast to bool for the IoU (1.0 when annotation and gt are both empty), MAE = mean |gt - annotation|."""
    b, h, w, _ = pred_masks.shape
    st = mask_stats(pred_masks, gt_masks, threshold, 0.0)
    hw = float(h * w)
    flip = _border_score(st, h, w) >= MASK_THRESHOLD
    i, u, ann = _iou_terms(st, hw, flip)
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where((ann == 0) & (st[:, 2] == 0), 1.0, i / u.astype(np.float32))
    mae = np.where(flip, st[:, 6] + st[:, 7], st[:, 4] + st[:, 5]) / hw
    return iou, mae, flip


def compute_IoU(gt_mask, pred_mask_f, threshold: float = 0.1):
    """Single-image form of test_generator.py:19-35: returns (iou, annotation) -- or a bare 1 when both are empty."""
    g = torch.as_tensor(np.asarray(gt_mask, dtype=np.float32)).reshape(1, *np.asarray(gt_mask).shape[:2], 1).cuda().contiguous()
    p = torch.as_tensor(np.asarray(pred_mask_f, dtype=np.float32)).reshape(g.shape).cuda().contiguous()
    iou, _, flip = evaluate_batch(g, p, threshold)
    pred = np.asarray(pred_mask_f) > threshold
    annotation = np.logical_not(pred) if flip[0] else pred
    if not annotation.any() and not np.asarray(gt_mask).astype(bool).any():
        return 1
    return float(iou[0]), annotation


def compute_mae(gt_mask, pred_mask_f) -> float:
    return float(np.mean(np.abs(np.asarray(gt_mask, dtype=np.float64) - np.asarray(pred_mask_f, dtype=np.float64))))


def evaluate_masks(learner, n_steps=None, verbose=True):
    """The loop of test_generator.py:_test_masks (:43-130) over learner.inference(): per-category IoU / MAE lists and
    the three reported averages.  `learner` is an AdversarialLearner after setup_inference(config, aug_test=False)."""
    cat_iou, cat_mae = {}, {}
    batch = getattr(learner.config, "batch_size", 1)
    if n_steps is None:
        n_steps = int(np.ceil(learner.test_samples / float(batch)))
    frames = 0
    for _ in range(n_steps):
        try:
            inf = learner.inference(None)
        except StopIteration:
            if verbose:
                print("End of testing dataset")
            break
        pm = torch.as_tensor(np.ascontiguousarray(inf["gen_masks"], dtype=np.float32)).cuda()
        gm = torch.zeros_like(pm) if inf["gt_masks"] is None else \
            torch.as_tensor(np.ascontiguousarray(inf["gt_masks"], dtype=np.float32)).cuda()  # (synthetic data: no annotation)
        iou, mae, _ = evaluate_batch(gm, pm)
        for b in range(pm.shape[0]):
            name = inf["img_fname"][b]
            name = name.decode("utf-8") if isinstance(name, (bytes, bytearray)) else str(name)
            parts = name.split("/")
            category = parts[-2] if len(parts) > 1 else "all"
            cat_iou.setdefault(category, []).append(float(iou[b]))
            cat_mae.setdefault(category, []).append(float(mae[b]))
            frames += 1
    tot_iou = sum(sum(v) for v in cat_iou.values())
    tot_mae = sum(sum(v) for v in cat_mae.values())
    per_cat = [float(np.mean(v)) for v in cat_iou.values()]
    res = {"category_iou": {k: float(np.mean(v)) for k, v in cat_iou.items()},
           "category_mae": {k: float(np.mean(v)) for k, v in cat_mae.items()},
           "dataset_iou": tot_iou / max(frames, 1), "dataset_mae": tot_mae / max(frames, 1),
           "sequence_iou": float(np.mean(per_cat)) if per_cat else 0.0, "frames": frames}
    if verbose:
        for cat in cat_iou:
            print("Category {}: IoU is {} and MAE is {}".format(cat, res["category_iou"][cat], res["category_mae"][cat]))
        print("The Average over the dataset: IoU is {} and MAE is {}".format(res["dataset_iou"], res["dataset_mae"]))
        print("The Average over sequences IoU is {}".format(res["sequence_iou"]))
        print("Success: Processed {} frames".format(frames))
    return res


def evaluate_ensemble(learner, n_steps=None, save_dir=None, verbose=True):
    """The loop of test_generator_ensemble.py:_test_masks (:20-125) over learner.inference() of the augmented graph:
    per frame the IoU / MAE of every central crop are averaged; with `save_dir` the per-frame buffers the offline
    post-processing reads are written as result_<k>.mat with the reference's keys (img_1_%03d, pred_mask_%03d,
    gt_mask_%03d, :101-111).  Like the reference, the first frame of a category enters its list with the LAST crop's
    score instead of the crop mean (:70-75)."""
    import os
    cat_iou, cat_mae = {}, {}
    crops = learner.test_crops
    if n_steps is None:
        n_steps = int(learner.test_samples)
    frames = 0
    for _ in range(n_steps):
        try:
            inf = learner.inference(None)
        except StopIteration:
            if verbose:
                print("End of testing dataset")
            break
        outs, fname = inf["outs"], inf["img_fname"]
        c_iou, c_mae = [], []
        iou = mae = 0.0
        for crop in crops:
            gt, pm = outs["gt_masks"][crop], outs["pred_masks"][crop]
            if gt is None:  # synthetic data: no annotation
                gt = outs["gt_masks"][crop] = np.zeros_like(np.asarray(pm), dtype=np.float32)
            res = compute_IoU(gt_mask=gt, pred_mask_f=pm)
            if isinstance(res, tuple):
                iou, out_mask = res
            else:  # both empty: the reference returns a bare 1 (and would fail to unpack it); score it as 1 / all-background
                iou, out_mask = 1.0, np.zeros_like(np.asarray(pm), dtype=bool)
            outs["pred_masks"][crop] = out_mask
            mae = compute_mae(gt_mask=gt, pred_mask_f=out_mask)
            c_iou.append(iou)
            c_mae.append(mae)
        name = fname.decode("utf-8") if isinstance(fname, (bytes, bytearray)) else str(fname)
        parts = name.split("/")
        category = parts[-2] if len(parts) > 1 else "all"
        if category in cat_iou:
            cat_iou[category].append(float(np.mean(c_iou)))
            cat_mae[category].append(float(np.mean(c_mae)))
        else:
            cat_iou[category], cat_mae[category] = [float(iou)], [float(mae)]
        if save_dir:
            import scipy.io as sio
            d = os.path.join(save_dir, category)
            os.makedirs(d, exist_ok=True)
            mat = {}
            for crop in crops:
                k = int(crop * 100)
                mat["img_1_{:03d}".format(k)] = outs["img_1s"][crop]
                mat["pred_mask_{:03d}".format(k)] = outs["pred_masks"][crop]
                mat["gt_mask_{:03d}".format(k)] = outs["gt_masks"][crop]
            sio.savemat(os.path.join(d, "result_{}.mat".format(len(cat_iou[category]))), mat)
        frames += 1
    tot_iou = sum(sum(v) for v in cat_iou.values())
    tot_mae = sum(sum(v) for v in cat_mae.values())
    res = {"category_iou": {k: float(np.mean(v)) for k, v in cat_iou.items()},
           "category_mae": {k: float(np.mean(v)) for k, v in cat_mae.items()},
           "dataset_iou": tot_iou / max(frames, 1), "dataset_mae": tot_mae / max(frames, 1), "frames": frames}
    if verbose:
        for cat in cat_iou:
            print("Category {}: IoU is {} and MAE is {}".format(cat, res["category_iou"][cat], res["category_mae"][cat]))
        print("The Average over the dataset: IoU is {} and MAE is {}".format(res["dataset_iou"], res["dataset_mae"]))
        print("Success: Processed {} frames".format(frames))
    return res
