"""datafusion_amd - MI355X (gfx950) implementation of the FusionCG hot path of FluoLab/datafusion."""
from .baseline import baseline
from .io import load_data
from .fusion import Fusion, FusionAdam, FusionCG, SumOperator, SumPoolOperator, squared_l2
from .optim import DfxAdam

__all__ = ["baseline", "load_data", "DfxAdam", "Fusion", "FusionAdam", "FusionCG", "SumOperator", "SumPoolOperator", "squared_l2"]
