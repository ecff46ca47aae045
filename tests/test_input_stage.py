"""Input stage ("next" row N1): host logic and oracle pinned on CPU; the fused crop / flip / resize kernel against the
numpy oracle on the GPU (bit-exact: same float32 op order, no FMA contraction)."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_np as ON


def test_central_crop_boxes_match_survey_sizes():
    # 384x640 -> 328x544, 346x576, 366x608, 384x640
    assert ON.central_crop_box(384, 640, 0.85)[2:] == (328, 544)
    assert ON.central_crop_box(384, 640, 0.9)[2:] == (346, 576)
    assert ON.central_crop_box(384, 640, 0.95)[2:] == (366, 608)
    assert ON.central_crop_box(384, 640, 1.0) == (0, 0, 384, 640)


def test_pair_tables():
    tr = ON.pair_table([5, 4], 3, True)          # max_temporal_len 3: forward from the first len-3, backward from the last len-3
    assert tr[:, 0].tolist() == [0, 1, 5, 3, 4, 8] and tr[:, 1].tolist() == [1, 1, 1, -1, -1, -1]
    te = ON.pair_table([5, 4], 2, False)         # every frame exactly once; the last t_len of a sequence look backwards
    assert sorted(te[:, 0].tolist()) == list(range(9))
    assert te[te[:, 1] < 0][:, 0].tolist() == [3, 4, 7, 8]
    tn = ON.pair_table([5], -2, False)
    assert tn[tn[:, 1] > 0][:, 0].tolist() == [0, 1] and tn[tn[:, 1] < 0][:, 0].tolist() == [2, 3, 4]


def test_nearest_legacy_known_answer():
    x = np.arange(6, dtype=np.float32).reshape(1, 1, 6, 1)
    y = ON.resize_nearest_legacy(x, 1, 4)       # scale 1.5: floor(0,1.5,3,4.5) = 0,1,3,4
    assert y.reshape(-1).tolist() == [0, 1, 3, 4]


def _make_davis(tmp, seqs=(("bear", 4), ("camel", 3)), h=60, w=80):
    from PIL import Image
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(tmp, "ImageSets", "480p"))
    lines = []
    for name, n in seqs:
        os.makedirs(os.path.join(tmp, "JPEGImages", "480p", name))
        os.makedirs(os.path.join(tmp, "Annotations", "480p", name))
        for i in range(n):
            img = (rng.random((h, w, 3)) * 255).astype(np.uint8)
            ann = ((rng.random((h, w)) > 0.5) * 255).astype(np.uint8)
            Image.fromarray(img).save(os.path.join(tmp, "JPEGImages", "480p", name, "%05d.png" % i))
            Image.fromarray(ann).save(os.path.join(tmp, "Annotations", "480p", name, "%05d.png" % i))
            lines.append("/JPEGImages/480p/%s/%05d.png /Annotations/480p/%s/%05d.png" % (name, i, name, i))
    for part in ("train", "val", "trainval"):
        with open(os.path.join(tmp, "ImageSets", "480p", part + ".txt"), "w") as f:
            f.write("\n".join(lines) + "\n")


def test_directory_iterator_and_flip_draws(tmp_path, capsys):
    from unsupervised_detection_amd import data
    _make_davis(str(tmp_path))
    it = data.DirectoryIterator(str(tmp_path), "train")
    assert it.samples == 7 and it.num_experiments == 2 and [len(s) for s in it.image_filenames] == [4, 3]
    assert it.image_filenames[1][0].endswith("JPEGImages/480p/camel/00000.png")
    with pytest.raises(IOError):
        data.DirectoryIterator(str(tmp_path / "nope"), "train")
    fl = data.draw_flips(np.random.default_rng(1), 4000)
    frac = [np.mean((fl == np.array(v)).all(1)) for v in ((0, 0), (1, 1), (1, 0), (0, 1))]
    assert all(abs(f - 0.25) < 0.03 for f in frac)   # four outcomes, 25 % each (aug_flips.py:35-45)
    cr = data.draw_crops(np.random.default_rng(2), 200, 384, 640, 0.9)
    assert (cr[:, 2] >= int(384 * 0.9)).all() and (cr[:, 2] <= 384).all() and (cr[:, 0] + cr[:, 2] <= 384).all()
    assert np.array_equal(data.pair_table([5, 4], 3, True), ON.pair_table([5, 4], 3, True))


@pytest.mark.gpu
def test_kernel_matches_oracle_bit_exact():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import data
    rng = np.random.default_rng(3)
    u8 = (rng.random((2, 48, 85, 3)) * 255).astype(np.uint8)       # DAVIS-480p aspect, odd width
    out = data.preprocess_image(torch.from_numpy(u8).cuda(), 38, 64).cpu().numpy()
    assert np.array_equal(out, ON.preprocess_image(u8, 38, 64))
    m8 = ((rng.random((2, 48, 85, 1)) > 0.5) * 255).astype(np.uint8)
    assert np.array_equal(data.preprocess_mask(torch.from_numpy(m8).cuda(), 38, 64).cpu().numpy(), ON.preprocess_mask(m8, 38, 64))
    x = (rng.random((3, 38, 64, 3)).astype(np.float32) - 0.5)
    prm = np.array([[2, 3, 30, 50, 0, 0], [0, 0, 38, 64, 1, 1], [5, 7, 33, 57, 1, 0]], np.int32)
    got = data.crop_flip_resize(torch.from_numpy(x).cuda(), 38, 64, prm).cpu().numpy()
    for b in range(3):
        ref = ON.flip_crop_resize(x[b:b + 1], *prm[b, :4], prm[b, 4], prm[b, 5])
        assert np.array_equal(got[b:b + 1], ref), b
    cc = data.central_cropping(torch.from_numpy(x).cuda(), 0.85).cpu().numpy()
    y0, x0, ch, cw = ON.central_crop_box(38, 64, 0.85)
    assert np.array_equal(cc, ON.flip_crop_resize(x, y0, x0, ch, cw, 0, 0))
    with pytest.raises(ValueError):
        data.crop_flip_resize(torch.from_numpy(x).cuda(), 38, 64, np.array([[0, 0, 39, 64, 0, 0]] * 3))


@pytest.mark.gpu
def test_reader_batches(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from unsupervised_detection_amd import data
    _make_davis(str(tmp_path))
    rd = data.Davis2016Reader(str(tmp_path), max_temporal_len=2, min_temporal_len=1, num_threads=2, seed=0)
    tr = rd.image_inputs(batch_size=2, partition="train", train_crop=0.9)
    b = next(tr)
    assert b["img1"].shape == (2, 384, 640, 3) and b["img2"].shape == (2, 384, 640, 3) and b["img1"].is_cuda
    assert float(b["img1"].min()) >= -0.5 - 1e-6 and float(b["img1"].max()) <= 0.5 + 1e-6
    te = list(rd.test_inputs(batch_size=4, partition="val", t_len=1, test_crop=0.9))
    assert sum(x["img1"].shape[0] for x in te) == 7 and te[-1]["img1"].shape[0] == 3   # drop_remainder=False
    assert te[0]["gt_mask"].shape == (4, 384, 640, 1) and te[0]["fname"][0].endswith(b"bear/00000.png")
    d, fname = next(rd.augmented_inputs(partition="val", t_len=1, test_crops=[0.85, 1.0]))
    assert set(d["img_1s"]) == {0.85, 1.0} and d["img_1s"][0.85].shape == (384, 640, 3) and fname.endswith(b"00000.png")
