"""Host-side mirror of the reference's mixture-of-logistics sampler (utils/distribution.py:87-123), the one piece of
the WaveRNN path that the mounted tree still holds.  Same name, arguments and return as the reference function; the
arithmetic runs in libwrnn_mi355.so (``wrnn_mol_sample``), there is no CPU path.

The reference draws its two uniform tensors internally (``uniform_(1e-5, 1 - 1e-5)``, :106 and :118); here they may
be supplied (``noise=(u_mix [B,T,nr_mix], u_logistic [B,T])``) so that runs are reproducible, and are drawn with
torch.rand over the same interval otherwise.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib

U_LO, U_HI = 1e-5, 1.0 - 1e-5          # distribution.py:106,118


def draw_uniforms(shape, device) -> torch.Tensor:
    return torch.rand(shape, device=device) * (U_HI - U_LO) + U_LO


def sample_from_discretized_mix_logistic(y: torch.Tensor, log_scale_min: Optional[float] = None,
                                         noise: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                                         return_mixture: bool = False):
    """y [B, C, T] with C = 3 * nr_mix -> samples [B, T] in [-1, 1] (and the picked mixture index [B, T])."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))
    if y.device.type != 'cuda':
        raise RuntimeError("wavernn_amd.distribution runs only on an MI355X (HIP) device; there is no CPU fallback")
    if y.dtype != torch.float32 or y.dim() != 3:
        raise TypeError("expected a float32 tensor [B, C, T]")
    assert y.size(1) % 3 == 0                                           # :98
    B, C, T = y.shape
    nr_mix = C // 3
    if noise is None:
        noise = (draw_uniforms((B, T, nr_mix), y.device), draw_uniforms((B, T), y.device))
    u_mix, u_log = (u.to(y.device, torch.float32).contiguous() for u in noise)
    if tuple(u_mix.shape) != (B, T, nr_mix) or tuple(u_log.shape) != (B, T):
        raise ValueError(f"noise must be (u_mix [B,T,{nr_mix}], u_logistic [B,T])")
    yc = y.contiguous()
    out = torch.empty(B, T, device=y.device)
    mix = torch.empty(B, T, dtype=torch.int32, device=y.device) if return_mixture else None
    with torch.cuda.device(y.device):
        _lib.check(_lib.load().wrnn_mol_sample(yc.data_ptr(), u_mix.data_ptr(), u_log.data_ptr(), B, nr_mix, T,
                                               float(log_scale_min), out.data_ptr(),
                                               None if mix is None else mix.data_ptr(),
                                               torch.cuda.current_stream(y.device).cuda_stream), "wrnn_mol_sample")
    return (out, mix) if return_mixture else out
