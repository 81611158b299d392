"""MI355X-native hot path of HKUST-KnowComp/DummyNode4GraphLearning: dummy-node / edge-to-vertex index builds and
GIN / RGCN / RGIN message passing as hand-written gfx950 HIP kernels behind a C ABI (libdn_hip.so), exposed
through the reference's nn.Module surface.  GPU only -- importing is cheap, every operator raises without the
built library or with CPU tensors."""
from . import graph, ops, transforms, tu_io  # noqa: F401
from .graph import BatchedGraph, GraphBatch  # noqa: F401
from .hipgraph import StepGraph  # noqa: F401

__version__ = "0.1.0"
