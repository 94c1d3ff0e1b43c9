import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    """PIL image / HxWxC uint8 ndarray -> float32 CxHxW in [0, 1]."""

    def __call__(self, pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        return t.float().div(255.0) if t.dtype == torch.uint8 else t.float()


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = list(mean), list(std)

    def __call__(self, t):
        c = t.shape[0]
        m = torch.tensor((self.mean * c)[:c] if len(self.mean) == 1 else self.mean, dtype=t.dtype).view(-1, 1, 1)
        s = torch.tensor((self.std * c)[:c] if len(self.std) == 1 else self.std, dtype=t.dtype).view(-1, 1, 1)
        return (t - m) / s
