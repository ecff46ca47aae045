"""Input stage of the hot path ("next" row N1 of SURVEY.md section 8f) + the synthetic DAVIS-shaped pairs of the benchmark.

Mirrors the reader of the reference (data/davis2016_data_utils.py; fbms_data_utils.py and segtrackv2_data_utils.py restate
the same pipeline) with the per-image work on the GPU: one fused kernel (udet_crop_flip_resize) does
uint8 -> /255 - 0.5, flips, the crop window and the TF-1.13 legacy resize.  Same names and argument meaning:

  DirectoryIterator(directory, part)                       :6-65   ImageSets/480p/<part>.txt -> per-sequence file lists
  Davis2016Reader.preprocess_image / preprocess_mask       :86-99
  Davis2016Reader.random_crop_image_pair / central_cropping / augment_pair   :101-146, random_flip_images aug_flips.py:35-45
  Davis2016Reader.image_inputs / test_inputs / augmented_inputs              :180-354  (pair tables + batching)

JPEG / PNG decoding stays on the host (Pillow); batches are dicts {"img1","img2","gt_mask","fname"} of device tensors, the
form AdversarialLearner consumes (config.data_source)."""
from __future__ import annotations

import os

import numpy as np
import torch

from ._ffi import c_f, c_i, c_p, check, lib

lib.udet_crop_flip_resize.restype = c_i
lib.udet_crop_flip_resize.argtypes = [c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_i, c_i, c_f, c_f, c_p]

READER_H, READER_W = 384, 640


def crop_flip_resize(src: torch.Tensor, out_h: int, out_w: int, params=None, nearest: bool = False, div: float = 1.0,
                     add: float = 0.0) -> torch.Tensor:
    """src [N,H,W,C] uint8 or float32 on the GPU; params int array [N,6] = (y0, x0, crop_h, crop_w, flip_lr, flip_td)."""
    if not (src.is_cuda and src.is_contiguous() and src.dtype in (torch.uint8, torch.float32) and src.dim() == 4):
        raise ValueError("src must be a contiguous [N,H,W,C] uint8/float32 CUDA(HIP) tensor")
    n, h, w, c = src.shape
    prm = None
    if params is not None:
        pa = np.ascontiguousarray(np.asarray(params, dtype=np.int32).reshape(n, 6))
        if (pa[:, 0] < 0).any() or (pa[:, 1] < 0).any() or (pa[:, 2] < 1).any() or (pa[:, 3] < 1).any() or \
                (pa[:, 0] + pa[:, 2] > h).any() or (pa[:, 1] + pa[:, 3] > w).any():
            raise ValueError("crop window outside the image")
        prm = torch.from_numpy(pa).to(src.device)
    out = torch.empty((n, out_h, out_w, c), dtype=torch.float32, device=src.device)
    check(lib.udet_crop_flip_resize(src.data_ptr(), int(src.dtype == torch.uint8), int(nearest), n, h, w, c,
                                    prm.data_ptr() if prm is not None else None, out.data_ptr(), out_h, out_w, div, add,
                                    torch.cuda.current_stream().cuda_stream))
    return out


def preprocess_image(frames_u8: torch.Tensor, out_h: int = READER_H, out_w: int = READER_W) -> torch.Tensor:
    """frames_u8: [N,H,W,3] uint8 on the GPU -> float32 [N,out_h,out_w,3] in [-0.5, 0.5] (:86-91)."""
    return crop_flip_resize(frames_u8, out_h, out_w, None, False, 255.0, -0.5)


def preprocess_mask(mask_u8: torch.Tensor, out_h: int = READER_H, out_w: int = READER_W) -> torch.Tensor:
    """mask_u8: [N,H,W,1] uint8 -> float32 [N,out_h,out_w,1] in [0,1], nearest-neighbour resize (:93-99)."""
    return crop_flip_resize(mask_u8, out_h, out_w, None, True, 255.0, 0.0)


def central_crop_box(h: int, w: int, frac: float):
    """tf.image.central_crop's window (TF 1.13): start = int((size - size*frac)/2), extent = size - 2*start."""
    if frac >= 1.0:
        return 0, 0, h, w
    y0, x0 = int((h - h * frac) / 2), int((w - w * frac) / 2)
    return y0, x0, h - 2 * y0, w - 2 * x0


def central_cropping(img: torch.Tensor, cropping_percent: float, nearest: bool = False) -> torch.Tensor:
    """tf.image.central_crop + resize back to the input size (:129-133)."""
    n, h, w, _ = img.shape
    if cropping_percent >= 1.0:
        return img
    y0, x0, ch, cw = central_crop_box(h, w, cropping_percent)
    return crop_flip_resize(img, h, w, np.tile(np.array([y0, x0, ch, cw, 0, 0], np.int32), (n, 1)), nearest)


def draw_flips(rng, n):
    """random_flip_images (aug_flips.py:35-45): case 0 -> keep | rotate 180, case 1 -> left-right | top-down, each 50 %.
    Returns [n,2] (flip_lr, flip_td)."""
    out = np.zeros((n, 2), np.int32)
    for i in range(n):
        if rng.integers(0, 2) == 0:
            if rng.integers(0, 2) == 1:
                out[i] = (1, 1)
        else:
            out[i] = (1, 0) if rng.integers(0, 2) == 0 else (0, 1)
    return out


def draw_crops(rng, n, h, w, max_cropping_percent):
    """random_crop_image_pair (:101-127): pct = p + U(0,1)*(1-p) in float32, crop = int(size*pct), uniform offset."""
    out = np.zeros((n, 4), np.int32)
    for i in range(n):
        pct = np.float32(max_cropping_percent) + np.float32(rng.random()) * np.float32(1.0 - max_cropping_percent)
        ch, cw = int(np.float32(h) * pct), int(np.float32(w) * pct)
        out[i] = (rng.integers(0, h - ch + 1), rng.integers(0, w - cw + 1), ch, cw)
    return out


def augment_pair(image_1: torch.Tensor, image_2: torch.Tensor, train_crop: float = 1.0, rng=None):
    """random flip + equal random crop of both frames, resized back (:135-146).  Batched: one draw per pair."""
    rng = rng or np.random.default_rng()
    n, h, w, _ = image_1.shape
    fl = draw_flips(rng, n)
    cr = draw_crops(rng, n, h, w, train_crop)
    prm = np.concatenate([cr, fl], 1)
    return crop_flip_resize(image_1, h, w, prm), crop_flip_resize(image_2, h, w, prm)


class DirectoryIterator(object):
    """The DAVIS-2016 split files (`ImageSets/480p/{train,val,trainval}.txt`, one "<image> <annotation>" pair of
    root-relative paths per line) grouped into sequences.  Attributes follow the reference reader's contract
    (data/davis2016_data_utils.py:6-65): image_filenames / annotation_filenames are per-sequence lists of absolute paths,
    samples the number of frames, num_experiments the number of sequences."""

    SPLITS = ("train", "val", "trainval")

    def __init__(self, directory, part="train"):
        from itertools import groupby
        self.directory = directory
        split = os.path.join(directory, "ImageSets", "480p", "%s.txt" % part) if part in self.SPLITS else ""
        if not os.path.isfile(split):
            raise IOError("Partition file not found")
        with open(split) as f:
            rows = [ln.split()[:2] for ln in f if ln.strip()]
        absolute = lambda rel: os.path.join(directory, rel.lstrip("/"))
        sequence_of = lambda row: row[0].strip("/").split("/")[2]  # JPEGImages/480p/<sequence>/<frame>.jpg
        self.image_filenames, self.annotation_filenames = [], []
        for _, frames in groupby(rows, key=sequence_of):  # consecutive lines of one sequence, in file order
            frames = list(frames)
            self.image_filenames.append([absolute(img) for img, _ in frames])
            self.annotation_filenames.append([absolute(ann) for _, ann in frames])
        self.samples = len(rows)
        self.num_experiments = len(self.image_filenames)
        if self.samples == 0:
            raise IOError("Did not find any file in the dataset folder")
        print("Found {} images belonging to {} experiments.".format(self.samples, self.num_experiments))


def pair_table(seq_lengths, t_len, training):
    """(frame index, direction) rows of image_inputs (:196-214) / test_inputs (:252-276)."""
    first, last, n = [], [], 0
    for ln in seq_lengths:
        if training:
            last.append(np.arange(n + t_len, n + ln)); first.append(np.arange(n, n + ln - t_len))
        elif t_len < 0:
            last.append(np.arange(n + abs(t_len), n + ln)); first.append(np.arange(n, n + abs(t_len)))
        else:
            first.append(np.arange(n, n + ln - t_len)); last.append(np.arange(n + ln - t_len, n + ln))
        n += ln
    first, last = np.concatenate(first), np.concatenate(last)
    return np.vstack([np.stack([first, np.ones_like(first)], 1), np.stack([last, -np.ones_like(last)], 1)]).astype(np.float32)


def _read_image(path, channels):
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im.convert("RGB" if channels == 3 else "L"), dtype=np.uint8).reshape(im.size[1], im.size[0], channels)


class Davis2016Reader(object):
    """data/davis2016_data_utils.py:68-354 as Python iterables of device batches."""

    def __init__(self, root_dir, max_temporal_len=3, min_temporal_len=1, num_threads=6, device="cuda", seed=None, loader=None,
                 shard=(0, 1)):
        """shard = (rank, world) for data-parallel training: every rank shuffles the pair table with the SAME generator
        (seed) and takes its own batch_size rows of each global batch of batch_size * world rows -- the ranks' pairs are
        disjoint and one epoch covers the table once; temporal shifts and augmentation draws come from a per-rank
        generator (seed + rank)."""
        assert min_temporal_len < max_temporal_len, "Temporal lenghts are not consistenst"
        assert min_temporal_len > 0, "Min temporal len should be positive"
        self.root_dir, self.max_temporal_len, self.min_temporal_len = root_dir, max_temporal_len, min_temporal_len
        self.num_threads, self.device = num_threads, device
        self.rank, self.world = int(shard[0]), max(1, int(shard[1]))
        assert 0 <= self.rank < self.world, "shard = (rank, world)"
        # the ranks' rows are disjoint only if every rank shuffles the pair table with the SAME generator: no OS entropy here
        assert self.world == 1 or seed is not None, "a sharded reader (world > 1) needs an explicit seed, the same on every rank"
        self.order_rng = np.random.default_rng(seed)  # identical on every rank: the shuffle of the pair table
        # one rank: a single stream for everything (as before); several: per-rank draws beside the shared shuffle
        self.rng = self.order_rng if self.world == 1 else np.random.default_rng(None if seed is None else seed + 1 + self.rank)
        self.loader = loader or _read_image  # (path, channels) -> uint8 [H,W,C]

    def get_filenames_list(self, partition):
        it = DirectoryIterator(self.root_dir, partition)
        self.val_samples = it.samples
        return it.image_filenames, it.annotation_filenames

    def _load(self, paths, channels):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max(1, self.num_threads)) as ex:
            arrs = list(ex.map(lambda p: self.loader(p, channels), paths))
        return torch.from_numpy(np.stack(arrs)).to(self.device)

    def image_inputs(self, batch_size=32, partition="train", train_crop=1.0, num_threads=6):
        """Endless training batches {"img1","img2"} (:180-230): img2 = img1 +- U{min..max} frames, augmented."""
        t_len = self.max_temporal_len
        file_list, _ = self.get_filenames_list(partition)
        filenames = np.concatenate(file_list)
        table = pair_table([len(f) for f in file_list], t_len, True)

        def gen():
            while True:
                order = self.order_rng.permutation(len(table))
                gb = batch_size * self.world  # rows of one global batch; this rank's slice of it
                for s in range(0, len(order) - gb + 1, gb):  # drop_remainder=True
                    rows = table[order[s + self.rank * batch_size:s + (self.rank + 1) * batch_size]]
                    shift = self.rng.integers(self.min_temporal_len, self.max_temporal_len + 1, len(rows))
                    i1 = rows[:, 0].astype(np.int32)
                    i2 = (shift.astype(np.float32) * rows[:, 1] + rows[:, 0]).astype(np.int32)
                    a = preprocess_image(self._load(filenames[i1], 3))
                    b = preprocess_image(self._load(filenames[i2], 3))
                    a, b = augment_pair(a, b, train_crop, self.rng)
                    yield {"img1": a, "img2": b, "gt_mask": None, "fname": [f.encode() for f in filenames[i1]]}
        return gen()

    def test_inputs(self, batch_size=32, partition="val", t_len=2, with_fname=False, test_crop=1.0):
        """One pass over the partition (:233-327): time(img2) - time(img1) = t_len except at sequence ends."""
        file_list, ann_list = self.get_filenames_list(partition)
        filenames, annotations = np.concatenate(file_list), np.concatenate(ann_list)
        table = pair_table([len(f) for f in file_list], t_len, False)

        def gen():
            for s in range(0, len(table), batch_size):  # drop_remainder=False
                rows = table[s:s + batch_size]
                i1 = rows[:, 0].astype(np.int32)
                i2 = (np.float32(abs(t_len)) * rows[:, 1] + rows[:, 0]).astype(np.int32)
                a = central_cropping(preprocess_image(self._load(filenames[i1], 3)), test_crop)
                b = central_cropping(preprocess_image(self._load(filenames[i2], 3)), test_crop)
                g = central_cropping(preprocess_mask(self._load(annotations[i1], 1)), test_crop)  # resize_images default: bilinear (:129-133)
                yield {"img1": a, "img2": b, "gt_mask": g, "fname": [f.encode() for f in filenames[i1]]}
        src = gen()
        return src

    def augmented_inputs(self, partition="val", t_len=2, test_crops=(1.0,)):
        """Per frame a dict of centrally cropped versions for the ensemble (:329-354), batch size 1."""
        def gen():
            for batch in self.test_inputs(batch_size=1, partition=partition, t_len=t_len, with_fname=True, test_crop=1.0):
                d = {"img_1s": {}, "img_2s": {}, "seg_1s": {}}
                for crop in test_crops:
                    d["img_1s"][crop] = central_cropping(batch["img1"], crop)[0]
                    d["img_2s"][crop] = central_cropping(batch["img2"], crop)[0]
                    d["seg_1s"][crop] = central_cropping(batch["gt_mask"], crop)[0]
                yield d, batch["fname"][0]
        return gen()


def _smooth_noise(rng, shape, sigma):
    """low-pass filtered uniform noise in [0,1] (separable box filters ~ gaussian), numpy."""
    a = rng.random(shape, dtype=np.float32)
    k = max(1, int(sigma))
    for _ in range(3):
        for ax in (0, 1):
            c = np.cumsum(np.pad(a, [(k, k) if i == ax else (0, 0) for i in range(a.ndim)], mode="reflect"), axis=ax, dtype=np.float64)
            n = a.shape[ax]
            hi = np.take(c, np.arange(2 * k, 2 * k + n), axis=ax)
            lo = np.take(c, np.arange(0, n), axis=ax)
            a = ((hi - lo) / (2 * k)).astype(np.float32)
    a -= a.min()
    a /= max(a.max(), 1e-6)
    return a


def synthetic_davis_pairs(batch: int, seed: int, h: int = 480, w: int = 854, max_disp: float = 8.0):
    """DAVIS-480p-shaped uint8 frame pairs: frame 2 = frame 1 warped by a smooth random displacement field of
    <= max_disp px + 1 % noise.  Returns two uint8 arrays [B,h,w,3]."""
    rng = np.random.default_rng(seed)
    f1 = np.empty((batch, h, w, 3), np.uint8)
    f2 = np.empty((batch, h, w, 3), np.uint8)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing="ij")
    for b in range(batch):
        img = 0.6 * _smooth_noise(rng, (h, w, 3), 6) + 0.4 * _smooth_noise(rng, (h, w, 3), 2)
        dy = (_smooth_noise(rng, (h, w), 40) * 2 - 1) * max_disp
        dx = (_smooth_noise(rng, (h, w), 40) * 2 - 1) * max_disp
        sy = np.clip(yy + dy, 0, h - 1.001)
        sx = np.clip(xx + dx, 0, w - 1.001)
        y0, x0 = sy.astype(np.int32), sx.astype(np.int32)
        ty, tx = (sy - y0)[..., None], (sx - x0)[..., None]
        warped = ((1 - ty) * ((1 - tx) * img[y0, x0] + tx * img[y0, x0 + 1]) +
                  ty * ((1 - tx) * img[y0 + 1, x0] + tx * img[y0 + 1, x0 + 1]))
        warped = np.clip(warped + 0.01 * rng.standard_normal(warped.shape).astype(np.float32), 0, 1)
        f1[b] = np.round(img * 255).astype(np.uint8)
        f2[b] = np.round(warped * 255).astype(np.uint8)
    return f1, f2
