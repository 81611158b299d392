from .conv import GCNConv, GINConv, RGCNConv, SAGEConv, global_add_pool, global_max_pool, global_mean_pool  # noqa: F401
from .models import GCN, GIN, RGCN, RGIN, GCN_concat_readout, GraphSAGE  # noqa: F401
