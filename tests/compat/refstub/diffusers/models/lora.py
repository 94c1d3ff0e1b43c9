"""LoRA-compatible layers of diffusers 0.25.0 with no LoRA attached: plain Linear / Conv2d that accept a `scale` argument."""
from torch import nn


class LoRACompatibleLinear(nn.Linear):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer

    def forward(self, hidden_states, scale: float = 1.0):
        return super().forward(hidden_states)


class LoRACompatibleConv(nn.Conv2d):
    def __init__(self, *args, lora_layer=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.lora_layer = lora_layer

    def forward(self, hidden_states, scale: float = 1.0):
        return super().forward(hidden_states)


class LoRALinearLayer(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("LoRA is not on the IDM-VTON inference path")


class LoRAConv2dLayer(LoRALinearLayer):
    pass


def adjust_lora_scale_text_encoder(text_encoder, lora_scale: float = 1.0):
    raise NotImplementedError("LoRA is not on the IDM-VTON inference path")
