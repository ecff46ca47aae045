"""The flags of common_flags.py:6-54 (same names, same defaults) without python-gflags (not installed here).
`FLAGS` can be passed as `config` wherever the reference passes gflags.FLAGS."""
import argparse
from types import SimpleNamespace

_DEFS = [
    ("img_width", int, 384), ("img_height", int, 192), ("batch_size", int, 16), ("beta1", float, 0.9),
    ("flow_normalizer", float, 80.0), ("max_epochs", int, 40), ("num_samples_train", int, 5000), ("train_crop", float, 0.9),
    ("max_temporal_len", int, 2), ("min_temporal_len", int, 1), ("cbn", float, 0.5), ("epsilon", float, 75.0),
    ("iters_rec", int, 1), ("iters_gen", int, 3), ("num_threads", int, 6), ("resume_train", bool, False),
    ("root_dir", str, "/your/path/to/DAVIS_2016"), ("train_partition", str, "trainval"), ("dataset", str, "DAVIS2016"),
    ("recover_ckpt", str, ""), ("flow_ckpt", str, ""), ("full_model_ckpt", str, ""), ("checkpoint_dir", str, ""),
    ("summary_freq", int, 30), ("save_freq", int, 5), ("generate_visualization", bool, False), ("test_crop", float, 0.9),
    ("test_temporal_shift", int, 1), ("ckpt_file", str, ""), ("test_partition", str, "val"), ("test_save_dir", str, ""),
    # not a reference flag: also write checkpoints in tf.train.Saver format under the reference's variable names
    ("save_tf_checkpoint", bool, False),
    # not a reference flag: explicit opt-in to synthetic DAVIS-shaped pairs / seeded random weights when no dataset or checkpoint
    # is given (the reference raises IOError in those cases, adversarial_learner.py:66-67,339-343; so does this port without it)
    ("synthetic", bool, False),
    # not a reference flag: BASELINE.json configs[4] -- fp16 multiplication (fp32 accumulation) in the convolution GEMMs
    ("conv_fp16", bool, False),
    # not a reference flag: the one-off kernel autotune of the training plan at start-up (~10 s; rank 0 tunes, the other ranks of a
    # data-parallel job load its configurations).  Untuned plans run on the built-in tile heuristics (12.3 instead of 10.9 ms per step at the benchmark shape).
    ("autotune", bool, True),
]


def default_flags():
    return SimpleNamespace(**{k: v for k, _, v in _DEFS})


def parse_flags(argv=None):
    ap = argparse.ArgumentParser()
    for k, t, v in _DEFS:
        if t is bool:
            ap.add_argument("--" + k, type=lambda s: s.lower() in ("1", "true", "yes"), default=v, nargs="?", const=True)
        else:
            ap.add_argument("--" + k, type=t, default=v)
    return ap.parse_args(argv)


FLAGS = default_flags()
