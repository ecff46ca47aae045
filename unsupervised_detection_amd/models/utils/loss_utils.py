"""`from .utils.loss_utils import charbonnier_loss, train_op` (models/adversarial_learner.py:8)."""
from ...functional import charbonnier_loss, train_op  # noqa: F401
