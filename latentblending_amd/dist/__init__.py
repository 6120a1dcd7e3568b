"""Multi-GPU branch farm (one process per GPU, torch.distributed: RCCL on MI355X, gloo in tests)."""
from .farm import BranchFarm

__all__ = ["BranchFarm"]
