from .loss import LOSS_REGISTRY, Loss  # noqa: F401
from . import contra_loss  # noqa: F401  (registers the contrastive losses)
