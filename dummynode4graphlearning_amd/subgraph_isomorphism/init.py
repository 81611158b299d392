"""Weight initialisation of the SI layers (reference: subgraph_isomorphism/utils/init.py:17-75,125-143).

Same RNG consumption as the reference (one nn.init.uniform_ per tensor), so the same torch.manual_seed
gives bit-identical initial weights (tests/test_host_logic.py pins this against the golden fixtures)."""
import math

import torch.nn as nn

from .act import LEAKY_RELU_A


def calculate_gain(activation):
    """init.py:17-49 (string branch)."""
    if activation in ("none", "maximum", "minimum"):
        nonlinearity = "linear"
    elif activation in ("relu", "relu6", "elu", "selu", "celu", "gelu"):
        nonlinearity = "relu"
    elif activation in ("leaky_relu", "prelu"):
        nonlinearity = "leaky_relu"
    elif activation in ("softmax", "sparsemax", "gumbel_softmax"):
        nonlinearity = "sigmoid"
    elif activation in ("sigmoid", "tanh"):
        nonlinearity = activation
    else:
        raise NotImplementedError(activation)
    return nn.init.calculate_gain(nonlinearity, LEAKY_RELU_A)


def calculate_fan_in_and_fan_out(x):
    """init.py:52-63: for [R, in, out] this gives fan_in = in*out, fan_out = R*out (sic)."""
    if x.dim() < 2:
        x = x.unsqueeze(-1)
    receptive = x[0][0].numel() if x.dim() > 2 else 1
    return x.size(1) * receptive, x.size(0) * receptive


def xavier_uniform_init(x, gain=1.0):
    """init.py:70-75."""
    fan_in, fan_out = calculate_fan_in_and_fan_out(x)
    std = gain * math.sqrt(2.0 / float(fan_in + fan_out))
    a = 1.7320508075688772 * std
    return nn.init.uniform_(x, -a, a)


def kaiming_normal_init(x, gain=1.0):
    """init.py:75-78: N(0, gain / sqrt(fan_in)) with the reference's own fan computation."""
    fan_in, _ = calculate_fan_in_and_fan_out(x)
    return nn.init.normal_(x, 0, gain / math.sqrt(fan_in))


def init_weight(x, activation="none", init="uniform"):
    """init.py:125-143 ('uniform', 'normal' and 'zero' are the branches the built layers reach)."""
    gain = calculate_gain(activation)
    if init == "uniform":
        xavier_uniform_init(x, gain=gain)
    elif init == "normal":
        kaiming_normal_init(x, gain=gain)
    elif init == "zero":
        nn.init.zeros_(x)
    else:
        raise ValueError("init=%s is not supported now." % init)


def init_module(x, activation="none", init="uniform"):
    """init.py:146-166 for nn.Linear: weight by `init`, bias zero."""
    if not isinstance(x, nn.Linear):
        raise ValueError("init_module: only nn.Linear is built")
    init_weight(x.weight, activation=activation, init=init)
    if x.bias is not None:
        nn.init.zeros_(x.bias)
