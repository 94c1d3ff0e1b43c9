import torch.nn.functional as F
from torch import nn

from ..utils import USE_PEFT_BACKEND
from .lora import LoRACompatibleLinear

ACTIVATION_FUNCTIONS = {"swish": nn.SiLU, "silu": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}


def get_activation(act_fn):
    act_fn = act_fn.lower()
    if act_fn not in ACTIVATION_FUNCTIONS:
        raise ValueError(f"Unsupported activation function: {act_fn}")
    return ACTIVATION_FUNCTIONS[act_fn]()


class GELU(nn.Module):
    def __init__(self, dim_in, dim_out, approximate="none", bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)
        self.approximate = approximate

    def forward(self, hidden_states):
        return F.gelu(self.proj(hidden_states), approximate=self.approximate)


class GEGLU(nn.Module):
    """proj -> chunk(2) -> hidden * gelu(gate)  (erf form)."""

    def __init__(self, dim_in, dim_out, bias=True):      # (`bias`: the reference's FeedForward passes it, attentionhacked_tryon.py:657)
        super().__init__()
        linear_cls = LoRACompatibleLinear if not USE_PEFT_BACKEND else nn.Linear
        self.proj = linear_cls(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states, scale: float = 1.0):
        args = () if USE_PEFT_BACKEND else (scale,)
        hidden_states, gate = self.proj(hidden_states, *args).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class ApproximateGELU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=bias)

    def forward(self, x):
        x = self.proj(x)
        return x * (1.702 * x).sigmoid()
