"""Data side of the hot path: NIfTI IO + RAS orientation in numpy, the MONAI transform chain of
ref:params/VSparams.py:205-245 on GPU-cached volumes."""
from . import nifti  # noqa: F401
