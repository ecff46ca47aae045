"""`from models.nets import generator_net, recover_net` (models/nets.py:4,45) -> HIP-backed versions."""
from ..functional import generator_net, recover_net  # noqa: F401
