"""`from .utils.general_utils import compute_all_IoU, disambiguate_forw_back` (models/adversarial_learner.py:11;
models/utils/general_utils.py:89-159): one statistics kernel of libudet.so behind both (evaluation.py)."""
from ...evaluation import compute_all_IoU, compute_boundary_score, disambiguate_forw_back  # noqa: F401
