"""Data-parallel plumbing for the hot path: one process per GPU, graphs sharded across ranks, ONE collective per
step -- a sum all-reduce of a flat gradient bucket (torch.distributed backend "nccl" == RCCL over xGMI on ROCm).

The reference is single-device; a batch is a disjoint union of independent graphs, so forward and
backward need no exchange.  Gradients of this path are a few MB, i.e. latency-bound on xGMI, so they
travel as a single flattened bucket whose slices ARE the parameters' .grad tensors (no pack/unpack copies).
"""
import torch
import torch.distributed as dist


def shard_graphs(batch_num_nodes, batch_num_edges, world_size):
    """Contiguous graph ranges per rank, balanced by nodes + edges.  Returns [(g_begin, g_end)] * world_size."""
    w = (torch.as_tensor(batch_num_nodes).long() + torch.as_tensor(batch_num_edges).long()).cpu()
    G = int(w.numel())
    csum = torch.cumsum(w, 0)
    total = int(csum[-1]) if G else 0
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first graph whose cumulative weight reaches the target closes the shard
        g = int(torch.searchsorted(csum.double(), torch.tensor([target], dtype=torch.float64), right=False)) + 1 if G else 0
        bounds.append(min(max(g, bounds[-1]), G))
    bounds.append(G)
    return [(bounds[r], bounds[r + 1]) for r in range(world_size)]


class FlatGradBucket:
    """All parameters' gradients live in one flat buffer; all_reduce() sums it across ranks in one collective."""

    def __init__(self, params, average=True):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dt, dev = self.params[0].dtype, self.params[0].device
        if any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("FlatGradBucket needs parameters of one dtype on one device")
        self.average = average
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(self.numel, dtype=dt, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)     # autograd accumulates in place into the view
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, async_op=False):
        if not (dist.is_available() and dist.is_initialized()):
            return None
        if self.average and dist.get_world_size() > 1:
            self.flat.div_(dist.get_world_size())
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)

    def bytes(self):
        return self.numel * self.flat.element_size()
