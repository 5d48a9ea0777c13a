from . import peft_sam  # noqa: F401
