"""`from .PWCNet.model_pwcnet import ModelPWCNet, _DEFAULT_PWCNET_TEST_OPTIONS` (models/adversarial_learner.py:10)."""
from ...functional import ModelPWCNet  # noqa: F401

# the lg-6-2 test configuration is the only one the path uses (model_pwcnet.py:5-16); the HIP plan is specialised to it
_DEFAULT_PWCNET_TEST_OPTIONS = {"verbose": False, "use_tf_data": False, "pyr_lvls": 6, "flow_pred_lvl": 2, "search_range": 4,
                                "use_dense_cx": True, "use_res_cx": True}
