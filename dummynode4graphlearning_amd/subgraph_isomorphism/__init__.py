from .rgcn import RGCNLayer, RGCNRepNet  # noqa: F401
from .rgin import RGINLayer, RGINRepNet  # noqa: F401
from . import bookkeeping  # noqa: F401
from .dual import CompGCNLayer, DMPLayer  # noqa: F401
