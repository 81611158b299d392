from .conv import GINConv, RGCNConv, global_add_pool, global_max_pool, global_mean_pool  # noqa: F401
from .models import GIN, RGCN, RGIN  # noqa: F401
