"""Reference-compatible layer surface (layers/ of JiaruiFeng/KP-GNN)."""
from .combine import AttentionCombine, GeometricCombine  # noqa: F401
from .gine import GINEConv  # noqa: F401
from .KPGCN import KPGCNConv  # noqa: F401
from .KPGIN import KPGINConv  # noqa: F401
from .KPGINplus import KPGINPlusConv  # noqa: F401
from .KPGraphSAGE import KPGraphSAGEConv  # noqa: F401
from .KGIN import KGINConv  # noqa: F401
from .layer_utils import make_gnn_layer  # noqa: F401
