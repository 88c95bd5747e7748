"""MI355X-native Relation-Network hot path (drop-in for the reference's model.py).

The directory name carries a hyphen (the upstream repo's name); import it through the
`relationnetworks_clevr_amd` shim at the repository root, or put this directory on
sys.path and `from model import RN` exactly as with the reference."""
from . import rn_hip, functional, options    # noqa: F401
from .model import RN, RelationalLayer, RelationalLayerBase, ConvInputModel, QuestionEmbedModel   # noqa: F401

__all__ = ["RN", "RelationalLayer", "RelationalLayerBase", "ConvInputModel", "QuestionEmbedModel", "rn_hip", "functional", "options"]
