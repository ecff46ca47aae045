"""MI355X-native (gfx950) implementation of the adversarial_learner hot path of
antonilo/unsupervised_detection, behind the reference's own Python surface.

Compute runs in hand-written HIP kernels (libudet.so, C ABI in include/udet.h);
PyTorch-ROCm is used only for device memory, streams and torch.distributed."""
__version__ = "0.1.0"
