"""Compose / ToTensor / Normalize with torchvision's documented semantics: ToTensor turns an (H, W[, C]) uint8 PIL image into
a float32 (C, H, W) tensor scaled by 1/255; Normalize subtracts the per-channel mean and divides by the per-channel std."""
import numpy as np
import torch


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x


class ToTensor:
    def __call__(self, pic):
        a = np.asarray(pic)
        if a.ndim == 2:
            a = a[:, :, None]
        assert a.dtype == np.uint8
        return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).to(torch.float32).div(255)


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)

    def __call__(self, t):
        return (t - self.mean[:, None, None]) / self.std[:, None, None]
