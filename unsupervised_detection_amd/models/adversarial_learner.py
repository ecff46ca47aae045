"""`from models.adversarial_learner import AdversarialLearner` (train.py:4, test_generator.py:9)."""
from ..learner import AdversarialLearner  # noqa: F401
