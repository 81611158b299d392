"""Count-prediction heads after the representation net.

Mirror of subgraph_isomorphism/models/pred.py:17-216 (PredictNet, MeanPredictNet, SumPredictNet: same constructor,
parameter names `p_fc g_fc pred_fc1 pred_fc2 weight_fc1 weight_fc2`, same `forward(p_rep, p_mask, g_rep, g_mask) ->
(y, w)`) and of the dummy-node masking that precedes it (models/basemodel.py:905-912).  These are dense products on the
padded [batch, max_len, dim] tensors that `split_and_batchify_graph_feats` produced -- no message passing; they run on
rocBLAS through torch, kept here so that a representation net from this package plugs into the reference's model head
with nothing in between.  (The reference sums / averages over ALL padded positions, including the `g_fc` bias of the
zero rows; that is reproduced, not fixed.)"""
import torch as th
import torch.nn as nn

from .act import map_activation_str_to_layer
from .dl import split_and_batchify_graph_feats
from .init import init_module

DUMMYFLAG = "is_dummy"


def mask_dummy_nodes(mask, dummy_flag, lens):
    """basemodel.py:905-912: positions of dummy vertices are cleared from the padded node mask ([B, max_n] bool).
    dummy_flag: [N] flag per node of the batched graph, lens: [B] nodes per graph."""
    dm = split_and_batchify_graph_feats(dummy_flag.view(-1, 1), lens, pre_pad=True)[0]
    return mask.masked_fill(dm.view(mask.shape).bool(), 0)


class PredictNet(nn.Module):
    def __init__(self, input_dim, hidden_dim, act_func="relu", dropout=0.0, return_weights=False):
        super().__init__()
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.act = map_activation_str_to_layer(act_func)
        self.drop = nn.Dropout(dropout)
        self.p_fc = nn.Linear(input_dim, hidden_dim)
        self.g_fc = nn.Linear(input_dim, hidden_dim)
        self.pred_fc1 = nn.Linear(hidden_dim * 4 + 4, hidden_dim)
        self.pred_fc2 = nn.Linear(hidden_dim + 4, 1)
        if return_weights:
            self.weight_fc1 = nn.Linear(hidden_dim * 4 + 2, hidden_dim)
            self.weight_fc2 = nn.Linear(hidden_dim + 2, 1)
        else:
            self.weight_fc1 = self.weight_fc2 = None
        for m, init in ((self.p_fc, "normal"), (self.g_fc, "normal"), (self.pred_fc1, "normal"), (self.pred_fc2, "zero")):
            init_module(m, activation=act_func, init=init)
        if return_weights:
            init_module(self.weight_fc1, activation=act_func, init="normal")
            init_module(self.weight_fc2, activation=act_func, init="zero")

    def agg_graph(self, g_rep, g_mask=None):
        raise NotImplementedError

    def agg_pattern(self, p_rep, p_mask=None):
        return self.agg_graph(p_rep, p_mask)

    def forward(self, p_rep, p_mask, g_rep, g_mask):
        bsz, g_len = p_mask.size(0), g_mask.size(1)
        pl = p_mask.float().sum(dim=1).view(bsz, 1)
        gl = g_mask.float().sum(dim=1).view(bsz, 1)
        pl_inv, gl_inv = 1.0 / pl, 1.0 / gl
        if p_rep.dim() == 2:
            p = p_rep.unsqueeze(1).expand(bsz, g_len, -1)
        elif p_rep.dim() == 3:
            p = self.agg_pattern(self.drop(self.p_fc(p_rep)), p_mask).unsqueeze(1).expand(bsz, g_len, -1)
        else:
            raise ValueError
        g = self.drop(self.g_fc(g_rep))
        w = None
        if self.weight_fc1 is not None:
            ex = lambda t: t.expand(bsz, g_len).unsqueeze(-1)  # noqa: E731
            w = self.act(self.weight_fc1(th.cat([p, g, g - p, g * p, ex(pl), ex(pl_inv)], dim=2)))
            w = self.weight_fc2(th.cat([w, ex(pl), ex(pl_inv)], dim=2)).squeeze(-1)
        p = p[:, 0, :]
        g = self.agg_graph(g)
        y = self.act(self.pred_fc1(th.cat([p, g, g - p, g * p, pl, gl, pl_inv, gl_inv], dim=1)))
        y = self.pred_fc2(th.cat([y, pl, gl, pl_inv, gl_inv], dim=1))
        return y, w


class MeanPredictNet(PredictNet):
    def agg_graph(self, g_rep, g_mask=None):
        return th.mean(g_rep, dim=1)


class SumPredictNet(PredictNet):
    def agg_graph(self, g_rep, g_mask=None):
        return th.sum(g_rep, dim=1)
