"""Fine-tuning of SAM on the HIP kernels (reference ``micro_sam/training``)."""
from .sam_trainer import SamTrainer  # noqa: F401
from .trainable_sam import TrainableSAM  # noqa: F401
from .util import ConvertToSamInputs, get_trainable_sam_model  # noqa: F401
