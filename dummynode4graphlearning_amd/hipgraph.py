"""HIP-graph capture of a training / inference step.

The GC model steps of the reference's configs are 100-200 launches of 5-30 us each (a 20 k-node batch): launched eagerly from
Python they take 1.6-2.5 ms, replayed from a HIP graph 0.5-1.1 ms.  `StepGraph` captures a callable once
and replays it; everything the callable reads must live in tensors that keep their address (copy new data INTO them), which is
how the reference's loops already use a batch (`data.to(device)` once per step, `main.py:39`).

    step = StepGraph(lambda: F.nll_loss(model(data), data.y).backward(), warmup=3)
    for _ in range(epochs):
        optimizer.zero_grad(set_to_none=False)      # gradients must keep their storage between replays
        step()                                      # forward + loss + backward, one graph launch
        optimizer.step()

Index structures (`graph.edge_index_of(data)`, `row_index_of`, ...) are built during the warm-up calls and cached on the batch, so
the captured region contains kernels only.  Collectives are left outside the graph (`parallel.FlatGradBucket.all_reduce()` after
the replay), as `bench.py` does."""
import torch


class StepGraph:
    """callable() -> replays `fn` from a HIP graph (captured after `warmup` eager calls on a side stream)."""

    def __init__(self, fn, warmup=3, fallback=True):
        self.fn, self.graph, self.result = fn, None, None
        if not torch.cuda.is_available():
            raise RuntimeError("StepGraph needs a GPU")
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):                              # warm-up: index builds, allocator growth, code-object load
            for _ in range(max(int(warmup), 1)):
                fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                self.result = fn()
            self.graph = g
        except RuntimeError as exc:
            # Only what stream capture itself refuses (a host read-back, a synchronising call, an allocation the capture cannot
            # serve) falls back to eager launches -- loudly, because the warm-up calls and the aborted capture have side effects
            # the caller should know about (BatchNorm running statistics advanced `warmup` extra times on the same batch, .grad
            # tensors re-bound).  Argument errors of the HIP library (DnHipError), shape bugs etc. propagate.
            torch.cuda.synchronize()
            msg = str(exc)
            capture_related = any(k in msg for k in ("captur", "Captur", "hipErrorStreamCapture", "cudaErrorStreamCapture",
                                                     "operation not permitted when stream is capturing"))
            from ._lib import DnHipError
            if not fallback or not capture_related or isinstance(exc, DnHipError):
                raise
            import warnings
            warnings.warn("StepGraph: HIP-graph capture failed (%s); the step runs eagerly from here on" % msg.splitlines()[0])
            self.graph = None

    @property
    def captured(self):
        return self.graph is not None

    def __call__(self):
        if self.graph is None:
            return self.fn()
        self.graph.replay()
        return self.result
