"""The reference's module paths for the hot path (`models.nets`, `models.PWCNet.model_pwcnet`, `models.utils.loss_utils`,
`models.utils.flow_utils`, `models.adversarial_learner`), so that its entry scripts switch over by changing the package prefix of
their imports.  Every name re-exported here is a driver over libudet.so; nothing is implemented in this folder."""
