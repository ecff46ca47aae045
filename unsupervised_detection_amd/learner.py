"""The reference's Python surface for the hot path, backed by libudet.so.

Mirrors (same names, argument meaning, return contract) what the three entry scripts of the reference use:
  AdversarialLearner.train(config)                     models/adversarial_learner.py:312
  AdversarialLearner.setup_inference(config, aug_test) models/adversarial_learner.py:594
  AdversarialLearner.inference(sess)                   models/adversarial_learner.py:606
  attributes test_iterator / test_samples / test_crops test_generator.py:60,62; test_generator_ensemble.py:30
and the functional sub-surface generator_net / recover_net / ModelPWCNet.predict_from_img_pairs /
charbonnier_loss / preprocess_flow_batch on PyTorch-ROCm NHWC float32 tensors.

`config` is any object with the attribute names of common_flags.py (config.FLAGS restates the defaults).
Dataset readers are outside this path: batches come from `config.data_source`, an
iterable of dicts {"img1","img2"[, "gt_mask","fname"]} holding reader-preprocessed tensors [B,384,640,3]
in [-0.5,0.5]; without one, synthetic DAVIS-shaped pairs are used (there is no dataset on this machine)."""
from __future__ import annotations

import math
import time
from itertools import count

import numpy as np
import torch

from . import data as _data
from . import weights as W
from .engine import GEN, REC, Engine, EngineConfig
from .trainer import TrainState, flush_weights, train_step

TEST_CROPS = [0.85, 0.9, 0.95, 1.0]  # adversarial_learner.py:531


def _engine_config(config, batch=None, in_hw=(384, 640)):
    g = lambda k, d: getattr(config, k, d)
    return EngineConfig(batch_size=batch or g("batch_size", 16), in_height=in_hw[0], in_width=in_hw[1],
                        img_height=g("img_height", 192), img_width=g("img_width", 384), flow_normalizer=g("flow_normalizer", 80.0),
                        cbn=g("cbn", 0.5), epsilon=g("epsilon", 75.0), beta1=g("beta1", 0.9), conv_fp16=bool(g("conv_fp16", False)))


def _latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint analogue: the model-<N> with the largest N among the files save() writes (`model-N`
    torch files, `model-N.tf.index` Saver prefixes of --save_tf_checkpoint) AND the reference's own `model-N.index`
    (adversarial_learner.py:347-348 resumes from whatever its Saver left in checkpoint_dir); else model.best; '' when the
    directory holds none.  A Saver checkpoint is returned as its prefix (what _load_weights' reader takes)."""
    import os
    import re
    if not (checkpoint_dir and os.path.isdir(checkpoint_dir)):
        return ""
    best, best_n = "", -1
    for name in sorted(os.listdir(checkpoint_dir)):
        m = re.fullmatch(r"(model-(\d+)(\.tf)?)(\.index)?", name)
        if not m or (m.group(3) and not m.group(4)):  # model-N | model-N.index | model-N.tf.index
            continue
        n = int(m.group(2))
        if n > best_n or (n == best_n and not m.group(4)):  # same N: prefer the native torch file
            best, best_n = os.path.join(checkpoint_dir, m.group(1)), n
    if not best and os.path.isfile(os.path.join(checkpoint_dir, "model.best")):
        best = os.path.join(checkpoint_dir, "model.best")
    return best


def pad_batch(batch, batch_size):
    """A one-pass reader (test_inputs, drop_remainder=False) ends with a short batch; the plans are batch-specialised, so the
    last batch is padded by repeating its final sample.  Returns (padded batch, number of valid rows): callers evaluate
    only the valid rows."""
    n = batch["img1"].shape[0]
    if n == batch_size:
        return batch, n
    if n > batch_size:
        raise ValueError("batch of %d pairs for a plan of %d" % (n, batch_size))
    out = dict(batch)
    for k in ("img1", "img2", "gt_mask"):
        t = batch.get(k)
        if t is not None:
            out[k] = torch.cat([t, t[-1:].expand(batch_size - n, *t.shape[1:])], 0).contiguous()
    return out, n


class _SyntheticSource:
    def __init__(self, batch, n_batches, seed=8964):
        self.batch, self.n, self.seed = batch, n_batches, seed

    def __iter__(self):
        for i in range(self.n):
            f1, f2 = _data.synthetic_davis_pairs(self.batch, self.seed + i)
            yield {"img1": _data.preprocess_image(torch.from_numpy(f1).cuda()),
                   "img2": _data.preprocess_image(torch.from_numpy(f2).cuda()),
                   "gt_mask": None, "fname": [b"synthetic_%05d" % (i * self.batch + j) for j in range(self.batch)]}


class AdversarialLearner(object):
    def __init__(self):
        self.engine = None
        self.state = None

    # ------------------------------------------------------------------ training ----
    def _load_weights(self, config, mode="train"):
        """Checkpoint policy of train() (:339-360) and of the test scripts (test_generator.py:45-55).  Accepted: a
        tf.train.Saver V2 checkpoint prefix (`<prefix>.index` + `<prefix>.data-*`, read by tf_checkpoint.py; a path that names
        the `.index` / `.data-00000-of-00001` file itself is reduced to its prefix, as the reference's own test script passes
        it), or {variable name: array} dicts -- torch.save'd, or .npz -- under the TF checkpoint's own names
        ("MaskNet//conv1/kernel", ...; optimizer slots and BN moving statistics are ignored) or the canonical ones of
        weights.param_table().
          flow_ckpt        -> PWC-Net, mandatory (flow_saver, :329,339-343; IOError when missing unless config.synthetic)
          train, resume_train:      full_model_ckpt, else the latest model-* of checkpoint_dir -> every network + global_step (:345-353)
          train, not resume_train:  recover_ckpt -> recover net (:354-356); full_model_ckpt is NOT read
          test:            ckpt_file -> every network it holds (test_generator.py:45-55, test_generator_ensemble.py)"""
        import os
        import re

        def prefix_of(path):
            return re.sub(r"\.(index|data-\d{5}-of-\d{5})$", "", path)

        def read(path):
            pre = prefix_of(path)
            if os.path.isfile(pre + ".index"):  # a Saver prefix, e.g. pwcnet.ckpt-595000
                from .tf_checkpoint import read_checkpoint
                return read_checkpoint(pre), "restored from TF checkpoint"
            if os.path.isfile(path):
                return (dict(np.load(path)) if path.endswith(".npz") else torch.load(path, map_location="cpu")), "loaded from"
            raise IOError("Could not find checkpoint file {}. Aborting.".format(path))

        def restore_all(d, flag, path, how):
            found = []
            for key, net in (("w_pwc", W.NET_PWC), ("w_gen", W.NET_GEN), ("w_rec", W.NET_REC)):
                try:
                    out[key] = W.from_tf_dict(d, net)
                    found.append(key[2:])
                except KeyError:
                    pass  # this checkpoint does not hold that network
            if not found:
                raise IOError("{} {} holds none of the networks' variables".format(flag, path))
            for k in ("global_step", "train_op/global_step"):
                if k in d:
                    self.global_step = int(np.asarray(d[k]))
            print("{} {} {} ({})".format(flag, how, path, ", ".join(found)))

        out = {}
        synthetic = bool(getattr(config, "synthetic", False))
        path = getattr(config, "flow_ckpt", "")
        if path:
            d, how = read(path)
            out["w_pwc"] = W.from_tf_dict(d, W.NET_PWC)
            print("Flow net loaded from {}".format(path))
        if mode == "train":
            if getattr(config, "resume_train", False):
                ckpt = getattr(config, "full_model_ckpt", "")
                # :345-350: full_model_ckpt when it exists, ELSE the latest checkpoint of checkpoint_dir (a missing
                # full_model_ckpt is not an error there)
                if not (ckpt and (os.path.isfile(prefix_of(ckpt) + ".index") or os.path.isfile(ckpt))):
                    ckpt = _latest_checkpoint(getattr(config, "checkpoint_dir", ""))
                assert ckpt, "Found no checkpoint to resume training!"
                d, how = read(ckpt)
                restore_all(d, "full_model_ckpt", ckpt, how)
                print("Resumed training from model {}".format(ckpt))
            elif getattr(config, "recover_ckpt", ""):
                d, how = read(config.recover_ckpt)
                out["w_rec"] = W.from_tf_dict(d, W.NET_REC)
                print("Recover net loaded from previous ckpt")
            else:
                print("No recover checkpoint found! Train Recover from Scratch")
        else:
            ckpt = getattr(config, "ckpt_file", "")
            if ckpt:
                d, how = read(ckpt)
                restore_all(d, "ckpt_file", ckpt, how)
                print("Resume model from checkpoint {}".format(ckpt))
            elif not synthetic:
                raise IOError("Checkpoint file not found")  # test_generator.py:52-53
        if "w_pwc" not in out and not synthetic:
            raise IOError("Could not find flow ckpt file. Aborting.")  # :342-343 -- a random-init PWC-Net is never used silently
        return out

    def train(self, config):
        """High level train function (adversarial_learner.py:312-420): alternates `iters_rec` recover steps and
        `iters_gen` generator steps (step % (iters_rec+iters_gen) < iters_rec -> recover), each on a new batch."""
        self.config = config
        B = config.batch_size
        self.engine = Engine(_engine_config(config, B))
        self.global_step = 0
        self.state = TrainState(self.engine, seed=getattr(config, "seed", 8964), autotune=bool(getattr(config, "autotune", False)),
                                **self._load_weights(config, "train"))
        n_params = sum(W.param_total(n) for n in (W.NET_PWC, W.NET_GEN, W.NET_REC))
        print("Number of params: {}".format(n_params))
        # data parallel: every step consumes batch_size pairs on each of `world` ranks (cli.py shards the pair table), so an
        # epoch -- validation, save_freq, max_epochs -- is num_samples_train / (batch_size * world) steps; world = 1 is the
        # reference's ceil(num_samples_train / batch_size) (:320)
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.train_steps_per_epoch = int(math.ceil(config.num_samples_train / (config.batch_size * world)))
        iters_rec, iters_gen = config.iters_rec, config.iters_gen
        print("-------------------------------------")
        print("Training {} Recover and {} Generator".format(iters_rec, iters_gen))
        print("-------------------------------------")
        sum_iters = iters_rec + iters_gen
        max_steps = self.train_steps_per_epoch * config.max_epochs
        source = getattr(config, "data_source", None)
        if source is None:
            if not getattr(config, "synthetic", False):
                raise IOError("no dataset: pass config.data_source (cli.py builds it from --root_dir) or opt in to synthetic pairs "
                              "with config.synthetic / --synthetic")
            source = _SyntheticSource(B, max_steps)
        it = iter(source)
        # cross-step pipelining (trainer.train_step): one batch of look-ahead, so that the frozen PWC-Net's flow of the
        # next batch is computed beside this step's backward pass
        nxt = next(it, None)
        for step in count(start=1):
            batch = nxt
            if batch is None:
                break
            nxt = next(it, None) if step < max_steps else None
            start_time = time.time()
            if step % sum_iters == 0:
                self.global_step += 1
            which = REC if (step % sum_iters) < iters_rec else GEN
            train_step(self.state, batch["img1"], batch["img2"], which,
                       next_pair=None if nxt is None else (nxt["img1"], nxt["img2"]))
            if step % config.summary_freq == 0:
                L = self.engine.losses()
                train_epoch = math.ceil(step / self.train_steps_per_epoch)
                train_step_ = step - (train_epoch - 1) * self.train_steps_per_epoch
                print("Epoch: [%2d] [%5d/%5d] time: %4.4f/it loss_generator: %4.4f loss_recover %4.4f"
                      % (train_epoch, train_step_, self.train_steps_per_epoch, time.time() - start_time, L["generator"], L["recover"]))
            if step % self.train_steps_per_epoch == 0:
                train_epoch = int(step / self.train_steps_per_epoch)
                self.epoch_end_callback(train_epoch)
                if train_epoch == config.max_epochs:
                    print("-------------------------------")
                    print("Training completed successfully")
                    print("-------------------------------")
                    break

    def validation_iou(self, source, n_steps=None):
        """Sum of compute_all_IoU over the validation batches / (steps * batch_size) (:422-433, :135-139): the test
        graph's generator on each pair, disambiguated masks against gt > 0.01."""
        from .evaluation import compute_all_IoU
        e = self.engine
        if self.state is not None:
            flush_weights(self.state)  # the last optimizer apply's re-layout is deferred: validate the weights save() writes
        total, steps, samples = 0.0, 0, 0
        for batch in source:
            if n_steps is not None and steps >= n_steps:
                break
            if batch.get("gt_mask") is None:
                continue
            batch, valid = pad_batch(batch, e.cfg.batch_size)
            e.forward(batch["img1"], batch["img2"], 0)
            total += float(compute_all_IoU(e.buffer("mask")[:valid].contiguous(), self._resize_gt(batch["gt_mask"])[:valid].contiguous()).sum())
            steps += 1
            samples += valid
        # the reference divides by steps * batch_size (its repeat()ed reader never yields a short batch); equal when none is short
        return total / max(samples, 1)

    def epoch_end_callback(self, epoch_num):
        """:422-448: validation IoU over `config.val_source` (when given), 'best' checkpoint when it improves, periodic
        checkpoint every save_freq epochs (trainable variables only; Adam slots are not saved by the reference)."""
        import os
        ckdir = getattr(self.config, "checkpoint_dir", "")
        val_source = getattr(self.config, "val_source", None)
        if val_source is not None:
            print("\nComputing Validation IoU")
            viou = self.validation_iou(val_source, getattr(self, "val_steps_per_epoch", None))
            print("Epoch [{}] Validation IoU: {}".format(epoch_num, viou))
            self.last_val_iou = viou
            if viou > getattr(self, "min_val_iou", -1.0e12):
                if ckdir:
                    os.makedirs(ckdir, exist_ok=True)
                    self.save(ckdir, "best")
                self.min_val_iou = viou
        if ckdir and epoch_num % getattr(self.config, "save_freq", 5) == 0:
            os.makedirs(ckdir, exist_ok=True)
            self.save(ckdir, epoch_num)

    def save(self, checkpoint_dir, step):
        import os
        print(" [*] Saving checkpoint to {}/model-{}".format(checkpoint_dir, step))
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_rank() != 0:
            return  # replicas are identical: rank 0 writes
        d = {}  # self.saver = every trainable variable (:317-319): pwcnet/* is trainable in the reference graph and is stored too
        d.update(W.as_dict(self.state.w_pwc.cpu(), W.NET_PWC))
        d.update(W.as_dict(self.state.w_gen.cpu(), W.NET_GEN))
        d.update(W.as_dict(self.state.w_rec.cpu(), W.NET_REC))
        d["global_step"] = torch.tensor(self.global_step)
        name = "model.best" if step == "best" else "model-{}".format(step)
        torch.save({k: v.clone() for k, v in d.items()}, os.path.join(checkpoint_dir, name))
        if getattr(self.config, "save_tf_checkpoint", False):
            # the same weights as a tf.train.Saver V2 checkpoint under the reference's variable names: its
            # test_generator.py / train.py --full_model_ckpt can restore them (tf_checkpoint.py)
            from .tf_checkpoint import tf_variable_name, write_checkpoint
            tf_vars = {tf_variable_name(k): v.numpy() for k, v in d.items() if k != "global_step"}
            tf_vars["train_op/global_step"] = np.array(self.global_step, np.int64)
            write_checkpoint(os.path.join(checkpoint_dir, name + ".tf"), tf_vars)

    # ----------------------------------------------------------------- inference ----
    def setup_inference(self, config, aug_test=False):
        """Sets up the inference graph (:594-604): test graph (batch_size, one recover call) or the augmented
        4-crop graph (batch 1, generator only, :525-592)."""
        self.config = config
        self.aug_test = aug_test
        self._ahead, self._exhausted = None, False  # inference() looks one batch ahead (PWC flow prefetch)
        # the augmented graph feeds ONE frame pair (batch 1, :547) through four central crops: here the crops are the batch of
        # one plan, so a frame costs one PWC-Net + generator pass at batch 4 instead of four passes at batch 1 (every op of
        # the path is per sample, so the masks are the same numbers)
        B = 1 if aug_test else config.batch_size
        self.engine = Engine(_engine_config(config, len(TEST_CROPS) if aug_test else B))
        self.state = TrainState(self.engine, seed=getattr(config, "seed", 8964), **self._load_weights(config, "test"))
        source = getattr(config, "data_source", None)
        if source is None:
            if not getattr(config, "synthetic", False):
                raise IOError("no dataset: pass config.data_source or opt in to synthetic pairs with config.synthetic / --synthetic")
            source = _SyntheticSource(B, 4)
        self.test_iterator = iter(source)
        self.test_samples = getattr(source, "n", 4) * B
        if aug_test:
            self.test_crops = TEST_CROPS
            print("Evaluating the following crops {}".format(TEST_CROPS))

    def _central_crop_resize(self, img, frac):
        """tf.image.central_crop + resize back to 384x640 (data/davis2016_data_utils.py:129-133,328-354)."""
        return _data.central_cropping(img, frac)

    def _resize_gt(self, gt):
        """The graphs resize the annotation to the working resolution with nearest-neighbour sampling
        (adversarial_learner.py:92-94 train / validation, :498-500 test, :568-570 augmented test)."""
        e = self.engine.cfg
        if gt is None or (gt.shape[1] == e.img_height and gt.shape[2] == e.img_width):
            return gt
        return _data.crop_flip_resize(gt.contiguous(), e.img_height, e.img_width, None, True)

    def inference(self, sess=None):
        """Outputs a dictionary with the results of the required operations (:606-623).  `sess` is accepted and
        ignored.  Raises StopIteration at the end of the data (the reference raises tf.errors.OutOfRangeError)."""
        e = self.engine
        if self.aug_test:
            # The frozen PWC-Net of the NEXT pair runs on the plan's prefetch lanes beside this pair's generator pass and the
            # device -> host copies of its results (one pair of look-ahead on the iterator; same results as forward()).
            def staged(b):
                if b["img1"].shape[0] != 1:
                    raise ValueError("the augmented test graph takes one frame pair per step (adversarial_learner.py:547)")
                c1 = torch.cat([self._central_crop_resize(b["img1"], crop) for crop in self.test_crops], 0)
                c2 = torch.cat([self._central_crop_resize(b["img2"], crop) for crop in self.test_crops], 0)
                e.prefetch_flow(c1, c2)
                return b
            ahead = getattr(self, "_ahead", None)
            if ahead is None:
                if getattr(self, "_exhausted", False):
                    raise StopIteration
                ahead = staged(next(self.test_iterator))
            batch = ahead
            e.prefetch_consume()
            try:
                self._ahead = staged(next(self.test_iterator))
            except StopIteration:
                self._ahead, self._exhausted = None, True
            e.forward_in_place(0)
            outs = {"pred_masks": {}, "gt_masks": {}, "img_1s": {}}
            masks, images = e.buffer("mask").cpu().numpy(), e.buffer("image").cpu().numpy()
            gt = batch.get("gt_mask")  # seg_1s[crop]: central_cropping of the annotation (bilinear, :348), then nearest
            for k, crop in enumerate(self.test_crops):
                outs["pred_masks"][crop] = masks[k]
                outs["img_1s"][crop] = images[k]
                outs["gt_masks"][crop] = None if gt is None else \
                    self._resize_gt(self._central_crop_resize(gt, crop))[0].cpu().numpy()
            return {"outs": outs, "img_fname": batch["fname"][0]}
        def staged_batch(b):  # (same one-pair look-ahead as above)
            b, nvalid = pad_batch(b, e.cfg.batch_size)  # short last batch of a one-pass reader: only its valid rows are returned
            e.prefetch_flow(b["img1"], b["img2"])
            return b, nvalid
        ahead = getattr(self, "_ahead", None)
        if ahead is None:
            if getattr(self, "_exhausted", False):
                raise StopIteration
            ahead = staged_batch(next(self.test_iterator))
        batch, n = ahead
        e.prefetch_consume()
        try:
            self._ahead = staged_batch(next(self.test_iterator))
        except StopIteration:
            self._ahead, self._exhausted = None, True
        e.forward_in_place(1)
        gt = self._resize_gt(batch.get("gt_mask"))
        return {"gen_masks": e.buffer("mask")[:n].cpu().numpy(), "pred_flow": e.buffer("pred")[:n].cpu().numpy(),
                "input_image": e.buffer("image")[:n].cpu().numpy(), "gt_flow": e.buffer("flow")[:n].cpu().numpy(),
                "gt_masks": None if gt is None else gt[:n].cpu().numpy(), "img_fname": np.array(batch["fname"])}


# ---------------------------------------------------------------- functional API ----
class HotPath:
    """Functional sub-surface bound to one set of weights and one shape (engine plans are shape-specialised)."""

    def __init__(self, batch, img_hw=(192, 384), in_hw=(384, 640), w_pwc=None, w_gen=None, w_rec=None, seed=8964, **flags):
        self.engine = Engine(EngineConfig(batch_size=batch, in_height=in_hw[0], in_width=in_hw[1], img_height=img_hw[0],
                                          img_width=img_hw[1], **flags))
        self.state = TrainState(self.engine, seed, w_pwc, w_gen, w_rec)

    def predict_from_img_pairs(self, img1s, img2s):
        """ModelPWCNet().predict_from_img_pairs (models/PWCNet/model_pwcnet.py:61-76)."""
        return self.engine.pwc_forward(img1s, img2s).clone()

    def generator_net(self, images, flows, scope=None, reuse=None, training=True):
        """models/nets.py:4.  NOTE: like the reference's call site (adversarial_learner.py:99-105) `flows` is the
        normalised flow *before* preprocess_flow_batch; the standardisation is fused into the input packer."""
        from ._ffi import check, lib
        e = self.engine
        e.buffer("image").copy_(images)
        e.buffer("flow").copy_(flows)
        check(lib.udet_generator_forward(e._h, e.ws.data_ptr(), e._stream()))
        return e.buffer("mask").clone()

    def recover_net(self, img1, flow_masked, mask, scope=None, reuse=None, f=0.25, training=True):
        """models/nets.py:45."""
        from ._ffi import check, lib
        e = self.engine
        B = e.cfg.batch_size
        fin, imgin = e.buffer("rec.fin"), e.buffer("rec.imgin")
        fin[:B, ..., 0:2] = flow_masked
        fin[:B, ..., 2:3] = 1.0
        fin[:B, ..., 3:4] = 1.0 - mask
        imgin[:B, ..., 0:3] = img1
        check(lib.udet_recover_forward(e._h, 1, e.ws.data_ptr(), e._stream()))
        return e.buffer("pred")[:B].clone()


# module-level names of the reference's function surface (models/nets.py:4,45; model_pwcnet.py:22; loss_utils.py:12,34;
# flow_utils.py:5), all drivers over libudet.so -- see functional.py
from .functional import (AdamOptimizer, AdversarialGraph, ModelPWCNet, charbonnier_loss, generator_net,  # noqa: E402,F401
                         preprocess_flow_batch, recover_net, train_op)
