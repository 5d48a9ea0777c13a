"""Fine-tuning of the SAM mask decoder on the HIP kernels (reference ``micro_sam/training``)."""
from .sam_trainer import SamTrainer  # noqa: F401
from .trainable_sam import TrainableSAM  # noqa: F401
from .util import ConvertToSamInputs  # noqa: F401
