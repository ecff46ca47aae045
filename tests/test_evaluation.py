"""Evaluation tail ("next" row N2): the numpy oracle restated from general_utils.py / test_generator.py is pinned by
known answers on CPU; the HIP statistics kernel + host finishing (unsupervised_detection_amd.evaluation) must agree
with it exactly (integer counts) on random, empty, full and border-hugging masks."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as ON


def _frame(h=12, w=16):
    m = np.zeros((h, w, 1), np.float32)
    return m


def test_boundary_score_known_answers():
    m = _frame()
    assert ON.compute_boundary_score(m) == 0.0
    m[:] = 1.0
    assert ON.compute_boundary_score(m) == 1.0
    m = _frame()
    m[0:2] = 1.0                      # the top strip only: 2*16 of 2*(2*16)+2*(2*12) + the 4+4 corner overlaps of the side strips
    expect = (2 * 16 + 2 * 2 + 2 * 2) / float(2 * 2 * 16 + 2 * 2 * 12)
    assert abs(ON.compute_boundary_score(m) - expect) < 1e-12


def test_compute_iou_flips_background_and_handles_empty():
    gt = _frame()
    gt[4:8, 5:10] = 1.0
    pred = _frame()
    pred[4:8, 5:10] = 1.0
    iou, ann = ON.compute_IoU(gt, pred)
    assert iou == 1.0 and ann.sum() == 20
    inv = 1.0 - pred                  # the complementary mask hugs every border -> flipped back
    iou2, ann2 = ON.compute_IoU(gt, inv)
    assert iou2 == 1.0 and np.array_equal(ann2, ann)
    assert ON.compute_IoU(_frame(), _frame()) == 1   # both empty: the reference returns a bare 1
    assert abs(ON.compute_mae(gt, ann) - 0.0) < 1e-12
    tf_iou = ON.compute_all_IoU(np.stack([pred, inv]), np.stack([gt, gt]))
    assert np.allclose(tf_iou, 1.0, atol=1e-6)


@pytest.mark.gpu
def test_hip_statistics_match_oracle():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import evaluation as E
    rng = np.random.default_rng(0)
    h, w = 48, 64
    preds, gts = [], []
    for k in range(6):
        p = rng.random((h, w, 1), dtype=np.float32)
        g = (rng.random((h, w, 1)) > 0.7).astype(np.float32)
        if k == 1:
            p[:] = 0.0; g[:] = 0.0            # empty / empty
        if k == 2:
            p[:] = 1.0                         # everything foreground -> flipped to empty
        if k == 3:
            p[:] = 0.0; p[10:30, 20:40] = 0.8; g[:] = 0.0; g[12:28, 22:44] = 1.0
        if k == 4:
            p = 1.0 - (p > 0.97).astype(np.float32)   # hugs the borders
        preds.append(p); gts.append(g)
    P, G = np.stack(preds), np.stack(gts)
    pt, gt_ = torch.from_numpy(P).cuda(), torch.from_numpy(G).cuda()
    iou, mae, flip = E.evaluate_batch(gt_, pt)
    for b in range(P.shape[0]):
        ref = ON.compute_IoU(G[b], P[b])
        if ref == 1:
            assert iou[b] == 1.0
            ann = np.zeros_like(P[b], bool)
        else:
            assert abs(iou[b] - float(ref[0])) < 1e-6
            ann = ref[1]
        assert abs(mae[b] - ON.compute_mae(G[b], ann)) < 1e-9
        assert abs(E.compute_boundary_score(P[b] > 0.1) - ON.compute_boundary_score(P[b] > 0.1)) < 1e-12
    assert np.allclose(E.compute_all_IoU(pt, gt_), ON.compute_all_IoU(P, G), atol=1e-9)
    fb = E.disambiguate_forw_back(pt).cpu().numpy()
    for b in range(P.shape[0]):
        binm = (P[b] > 0.1).astype(np.float32)
        ref = binm if ON.compute_boundary_score(binm) < 0.6 else 1.0 - binm
        assert np.array_equal(fb[b], ref)
    single = E.compute_IoU(G[3], P[3])
    assert abs(single[0] - float(ON.compute_IoU(G[3], P[3])[0])) < 1e-6 and np.array_equal(single[1], ON.compute_IoU(G[3], P[3])[1])
    assert E.compute_IoU(G[1], P[1]) == 1


@pytest.mark.gpu
def test_evaluate_masks_report(monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import evaluation as E

    class FakeLearner:
        class config:
            batch_size = 2
        test_samples = 4

        def __init__(self):
            self.k = 0

        def inference(self, sess):
            if self.k == 2:
                raise StopIteration
            self.k += 1
            g = np.zeros((2, 16, 24, 1), np.float32); g[:, 4:10, 6:14] = 1.0
            p = g * 0.9
            return {"gt_masks": g, "gen_masks": p, "img_fname": [b"davis/bear/00001.jpg", b"davis/camel/00001.jpg"]}

    res = E.evaluate_masks(FakeLearner(), verbose=False)
    assert res["frames"] == 4 and res["dataset_iou"] == 1.0 and res["dataset_mae"] == 0.0
    assert set(res["category_iou"]) == {"bear", "camel"} and res["sequence_iou"] == 1.0


@pytest.mark.gpu
def test_ensemble_report_and_mat_buffers(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import scipy.io as sio
    from unsupervised_detection_amd import evaluation as E

    class FakeLearner:
        test_crops = [0.9, 1.0]
        test_samples = 2

        def __init__(self):
            self.k = 0

        def inference(self, sess):
            self.k += 1
            g = np.zeros((16, 24, 1), np.float32); g[4:10, 6:14] = 1.0
            outs = {"pred_masks": {c: g * 0.8 for c in self.test_crops}, "gt_masks": {c: g.copy() for c in self.test_crops},
                    "img_1s": {c: np.zeros((16, 24, 3), np.float32) for c in self.test_crops}}
            return {"outs": outs, "img_fname": b"fbms/cars1/%05d.jpg" % self.k}

    res = E.evaluate_ensemble(FakeLearner(), save_dir=str(tmp_path), verbose=False)
    assert res["frames"] == 2 and res["dataset_iou"] == 1.0 and res["category_iou"] == {"cars1": 1.0}
    m = sio.loadmat(str(tmp_path / "cars1" / "result_2.mat"))
    assert {"img_1_090", "pred_mask_090", "gt_mask_090", "img_1_100", "pred_mask_100", "gt_mask_100"} <= set(m)
    assert m["pred_mask_100"].sum() == 48
