"""Waveform distances (reference ``audiotools/metrics/distance.py``)."""
import torch
from torch import nn


def _pick(x, y, attribute):
    """Signals contribute ``attribute``; plain tensors pass through (distance.py:36-39, 93-98)."""
    if hasattr(x, "audio_data") and not torch.is_tensor(x):
        return getattr(x, attribute), getattr(y, attribute)
    return x, y


class L1Loss(nn.L1Loss):
    """``nn.L1Loss`` between one attribute (default ``audio_data``) of two signals
    (distance.py:7-40).  ``weight`` is carried for the trainer, not applied."""

    def __init__(self, attribute: str = "audio_data", weight: float = 1.0, **kwargs):
        self.attribute = attribute
        self.weight = weight
        super().__init__(**kwargs)

    def forward(self, x, y):
        a, b = _pick(x, y, self.attribute)
        return super().forward(a, b)


class SISDRLoss(nn.Module):
    """Negative scale-invariant SDR in dB per batch item (distance.py:43-131).

    NOTE the reference's argument roles: the FIRST argument is treated as the reference and the
    second as the estimate.  ``scaling=False`` gives plain SNR; ``zero_mean`` removes the mean of
    both; ``clip_min`` floors the loss; ``reduction`` in {"mean", "sum", anything else = none}."""

    def __init__(self, scaling: int = True, reduction: str = "mean", zero_mean: int = True, clip_min: int = None,
                 weight: float = 1.0):
        super().__init__()
        self.scaling, self.reduction, self.zero_mean, self.clip_min, self.weight = scaling, reduction, zero_mean, clip_min, weight

    def forward(self, x, y):
        eps = 1e-8
        ref, est = _pick(x, y, "audio_data")
        nb = ref.shape[0]
        ref = ref.reshape(nb, -1, 1)      # (batch, samples of all channels, 1)
        est = est.reshape(nb, -1, 1)
        if self.zero_mean:
            ref = ref - ref.mean(dim=1, keepdim=True)
            est = est - est.mean(dim=1, keepdim=True)
        if self.scaling:
            alpha = ((est * ref).sum(dim=1) + eps) / ((ref ** 2).sum(dim=1) + eps)
            target = alpha.unsqueeze(1) * ref
        else:
            target = ref
        residual = est - target
        ratio = (target ** 2).sum(dim=1) / (residual ** 2).sum(dim=1)
        loss = -10 * torch.log10(ratio + eps)
        if self.clip_min is not None:
            loss = torch.clamp(loss, min=self.clip_min)
        if self.reduction == "mean":
            return loss.mean()
        if self.reduction == "sum":
            return loss.sum()
        return loss
