"""Host helpers of the sampling path."""
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """Gaussian noise factory with the reference's seed semantics (utils.py:62-97 of BrepGen, itself the
    diffusers helper): a CPU generator always draws on the CPU and the result is then moved, so a seeded run is
    reproducible across devices; a list of generators seeds every batch element separately."""
    device = torch.device(device) if device is not None else torch.device("cpu")
    layout = layout or torch.strided
    rand_device = device
    if generator is not None:
        gen_type = (generator[0] if isinstance(generator, (list, tuple)) else generator).device.type
        if gen_type != device.type:
            if gen_type == "cpu":
                rand_device = torch.device("cpu")
            else:
                raise ValueError(f"Cannot generate a {device} tensor from a generator of type {gen_type}.")
    if isinstance(generator, (list, tuple)):
        one = (1,) + tuple(shape[1:])
        parts = [torch.randn(one, generator=g, device=rand_device, dtype=dtype, layout=layout) for g in generator]
        return torch.cat(parts, dim=0).to(device)
    return torch.randn(tuple(shape), generator=generator, device=rand_device, dtype=dtype, layout=layout).to(device)
