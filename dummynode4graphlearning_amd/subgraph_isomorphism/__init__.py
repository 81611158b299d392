from .rgcn import RGCNLayer, RGCNRepNet  # noqa: F401
from .rgin import RGINLayer, RGINRepNet  # noqa: F401
from . import bookkeeping  # noqa: F401
from .dual import CompGCNLayer, DMPLayer  # noqa: F401
from .pred import MeanPredictNet, PredictNet, SumPredictNet, mask_dummy_nodes  # noqa: F401
from .dl import split_and_batchify_graph_feats  # noqa: F401
