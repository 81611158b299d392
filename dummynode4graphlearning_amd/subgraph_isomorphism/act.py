"""Activation table of the SI models (reference: subgraph_isomorphism/utils/act.py:457-489).

The reference returns SHARED singleton modules from a global dict; stateless activations make that
unobservable, so fresh modules are created here (PReLU would otherwise share one parameter across layers --
SURVEY.md appendix A)."""
import torch.nn as nn

LEAKY_RELU_A = 1 / 5.5  # reference: constants.py:10


class Identity(nn.Module):
    def forward(self, x):
        return x


_FACTORY = {
    "none": Identity,
    "sigmoid": nn.Sigmoid,
    "tanh": nn.Tanh,
    "relu": nn.ReLU,
    "relu6": nn.ReLU6,
    "leaky_relu": lambda: nn.LeakyReLU(negative_slope=LEAKY_RELU_A),
    "prelu": lambda: nn.PReLU(init=LEAKY_RELU_A),
    "elu": nn.ELU,
    "celu": nn.CELU,
    "selu": nn.SELU,
    "gelu": nn.GELU,
    "softmax": lambda: nn.Softmax(dim=-1),
}


def map_activation_str_to_layer(act_func, **kw):
    if act_func not in _FACTORY:
        raise NotImplementedError(act_func)
    act = _FACTORY[act_func]()
    for k, v in kw.items():
        if hasattr(act, k):
            try:
                setattr(act, k, v)
            except Exception:
                pass
    return act
