"""Ragged -> padded conversion after the representation net.

reference: subgraph_isomorphism/utils/dl.py:51-81 ``split_and_batchify_graph_feats`` -- a Python loop of ``th.cat`` per
graph; here one gather launch (every padded slot is a 0- or 1-element segment of the HIP gather kernel), with the
backward as the inverse row gather.  Same signature and return value: (feats [B, max_n, ...], mask [B, max_n])."""
import torch as th

from .. import ops


class _PadRows(th.autograd.Function):
    @staticmethod
    def forward(ctx, x, slot_ptr, slot_src, row_slot, shape):
        out = ops.gather_segsum(x.contiguous(), slot_src, slot_ptr, slot_ptr.numel() - 1)
        ctx.save_for_backward(row_slot)
        return out.view(shape)

    @staticmethod
    def backward(ctx, g):
        (row_slot,) = ctx.saved_tensors
        g2 = g.contiguous().view(-1, g.shape[-1])
        return ops.gather_segsum(g2, row_slot, None), None, None, None, None


def split_and_batchify_graph_feats(batched_graph_feats, graph_sizes, pre_pad=False):
    bsz = graph_sizes.size(0)
    device = batched_graph_feats.device
    sizes = graph_sizes.view(-1).long()
    min_size, max_size = int(sizes.min()), int(sizes.max())
    if min_size == max_size:                                              # dl.py:56-59
        feats = batched_graph_feats.view(bsz, max_size, -1)
        return feats, th.ones((bsz, max_size), dtype=th.bool, device=device)
    pos = th.arange(max_size, device=device).view(1, -1)
    if pre_pad:                                                           # dl.py:66-72: zeros first, rows right-aligned
        mask = pos >= (max_size - sizes).view(-1, 1)
    else:                                                                 # dl.py:73-79
        mask = pos < sizes.view(-1, 1)
    flat = mask.reshape(-1)
    slot_ptr = th.cat([th.zeros(1, dtype=th.long, device=device), th.cumsum(flat.long(), 0)]).to(th.int32)
    row_slot = th.nonzero(flat).reshape(-1).to(th.int32)                  # padded slot of every real row (row order kept)
    slot_src = th.arange(row_slot.numel(), device=device, dtype=th.int32)
    x2 = batched_graph_feats.reshape(batched_graph_feats.shape[0], -1)
    if not x2.is_floating_point():                                        # flags / ids (e.g. the dummy mask, basemodel.py:907)
        feats = _PadRows.apply(x2.float(), slot_ptr, slot_src, row_slot, (bsz, max_size, x2.shape[1]))
        return feats.to(batched_graph_feats.dtype), mask
    feats = _PadRows.apply(x2, slot_ptr, slot_src, row_slot, (bsz, max_size, x2.shape[1]))
    return feats, mask
