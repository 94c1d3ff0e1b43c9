import torch


def maybe_allow_in_graph(cls):
    return cls


def apply_freeu(resolution_idx, hidden_states, res_hidden_states, **freeu_kwargs):
    raise NotImplementedError("FreeU is not on the IDM-VTON path")


RECORD = None                                        # tests: a list that receives every draw, in order


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor: draws on the generator's device (CPU generator -> CPU draw, then moved)."""
    out = _randn_tensor(shape, generator, device, dtype, layout)
    if RECORD is not None:
        RECORD.append(out.clone())
    return out


def _randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    rand_device = device
    if generator is not None:
        gen_device_type = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_device_type != getattr(device, "type", device) and gen_device_type == "cpu":
            rand_device = "cpu"
    layout = layout or torch.strided
    if isinstance(generator, list):
        shape = (1,) + tuple(shape[1:])
        latents = torch.cat([torch.randn(shape, generator=g, device=rand_device, dtype=dtype, layout=layout) for g in generator], dim=0)
    else:
        latents = torch.randn(shape, generator=generator, device=rand_device, dtype=dtype, layout=layout)
    return latents.to(device)
