from .build import make_lr_scheduler, make_optimizer  # noqa: F401
from .lr_scheduler import WarmupMultiStepLR  # noqa: F401
