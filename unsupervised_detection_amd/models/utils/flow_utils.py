"""`from .utils.flow_utils import preprocess_flow_batch` (models/adversarial_learner.py:9)."""
from ...functional import preprocess_flow_batch  # noqa: F401
