"""Downstream predictors trained on the enhanced embedding.  Not part of the pretraining hot path: they are
ordinary ``torch.nn`` modules on the GPU (rocBLAS / MIOpen through PyTorch-ROCm) behind the frozen HIP encoder of ``enhance.py``."""
from .stgcn import STGCN   # noqa: F401
