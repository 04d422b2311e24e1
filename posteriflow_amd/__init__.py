"""posteriflow_amd -- MI355X-native hot path of PosteriFlow (flow density / sampling
and strain embedding) behind the reference's Python API.  See DESIGN.md."""
from .flows import FLOW_NORM_BOUND, NSFPosteriorFlow, PSDScaledNormal, create_flow_model  # noqa: F401

from .npe import (PARAM_NAMES, CoherentEncoder, LeanNPE, LeanStrainEncoder, ParamScaler,  # noqa: F401
                  batch_nll)

from .remix import RemixDataset  # noqa: F401

__all__ = ["RemixDataset", "NSFPosteriorFlow", "PSDScaledNormal", "create_flow_model", "FLOW_NORM_BOUND",
           "LeanNPE", "LeanStrainEncoder", "CoherentEncoder", "ParamScaler", "PARAM_NAMES", "batch_nll"]
