"""Host-side mirror of reference LengthRegulator (models/common_layers.py:12-24).

Same constructor and ``forward(x[B,T,C], dur[B,T]) -> [B, max_i sum(round(dur_i)), C]``,
including the in-place clamp of negative durations on the caller's tensor.  The expand runs in
libwrnn_mi355.so (prefix-sum kernel + HBM-streaming row copy); there is no CPU fallback.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


class LengthRegulator(nn.Module):

    def __init__(self):
        super().__init__()

    def forward(self, x: torch.Tensor, dur: torch.Tensor) -> torch.Tensor:
        if x.device.type != 'cuda':
            raise RuntimeError("wavernn_amd.LengthRegulator runs only on an MI355X (HIP) device; "
                               "there is no CPU fallback")
        if x.dtype != torch.float32 or dur.dtype != torch.float32:
            raise TypeError("LengthRegulator expects float32 x and dur (the reference path is fp32)")
        if x.dim() != 3 or dur.dim() != 2 or x.shape[:2] != dur.shape:
            raise ValueError(f"expected x [B,T,C] and dur [B,T], got {tuple(x.shape)} and {tuple(dur.shape)}")
        if dur.device != x.device:
            raise RuntimeError("x and dur must be on the same device")
        B, T, C = x.shape
        lib = _lib.load()
        stream = torch.cuda.current_stream(x.device).cuda_stream
        if B == 0 or T == 0:
            return x.new_zeros(B, 0, C)
        # the reference clamps the caller's tensor in place (dur[dur < 0] = 0.); a non-contiguous
        # dur is clamped through a contiguous copy that is written back
        d = dur if dur.is_contiguous() else dur.contiguous()
        cumsum = torch.empty(B, T, dtype=torch.int32, device=x.device)
        lens = torch.empty(B, dtype=torch.int32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.wrnn_lr_prepare(d.data_ptr(), B, T, cumsum.data_ptr(), lens.data_ptr(), stream),
                       "wrnn_lr_prepare")
            if d is not dur:
                dur.copy_(d)
            M = int(lens.cpu().max()) if B else 0   # the only host sync: the output length is data dependent
            out = torch.empty(B, M, C, dtype=x.dtype, device=x.device)
            if M == 0 or C == 0:
                return out
            if C % 4 != 0:
                raise ValueError(f"channel count C={C} must be a multiple of 4")
            xc = x.contiguous()
            _lib.check(lib.wrnn_lr_expand(xc.data_ptr(), cumsum.data_ptr(), B, T, C, out.data_ptr(), M, stream),
                       "wrnn_lr_expand")
        return out

    STATIC_MAX_TOKENS = 1 << 18   # wrnn_lr_regulate_static's scratch (include/wrnn_mi355.h)

    def forward_capped(self, x: torch.Tensor, dur: torch.Tensor, max_len: int, out: torch.Tensor = None):
        """The expand without the host read-back of the output length (build extension, not in the reference): the
        caller gives an upper bound ``max_len`` of the longest item; returns (out [B, max_len, C] with rows past
        each item's length zero, lens [B] int32 on the device = the true lengths, so the caller can narrow later,
        e.g. ``out[:, :int(lens.max())]`` once it has to know).  One library call (wrnn_lr_regulate_static), no
        synchronisation; with ``out`` given it allocates only ``lens``.  Clamps ``dur`` in place like ``forward``.
        Calls that overlap in time on one GPU share the library's scratch and must be serialised."""
        if x.device.type != 'cuda':
            raise RuntimeError("wavernn_amd.LengthRegulator runs only on an MI355X (HIP) device; there is no CPU fallback")
        if x.dtype != torch.float32 or dur.dtype != torch.float32:
            raise TypeError("LengthRegulator expects float32 x and dur (the reference path is fp32)")
        if x.dim() != 3 or dur.dim() != 2 or x.shape[:2] != dur.shape or dur.device != x.device:
            raise ValueError(f"expected x [B,T,C] and dur [B,T] on one device, got {tuple(x.shape)} and {tuple(dur.shape)}")
        if not dur.is_contiguous():
            raise ValueError("forward_capped needs contiguous durations (they are clamped in place)")
        B, T, C = x.shape
        max_len = int(max_len)
        if C % 4 != 0:
            raise ValueError(f"channel count C={C} must be a multiple of 4")
        if out is None:
            out = torch.empty(B, max_len, C, dtype=x.dtype, device=x.device)
        elif tuple(out.shape) != (B, max_len, C) or not out.is_contiguous() or out.device != x.device or out.dtype != x.dtype:
            raise ValueError("out must be a contiguous float32 [B, max_len, C] tensor on x's device")
        lens = torch.zeros(B, dtype=torch.int32, device=x.device)
        if B == 0 or T == 0:
            return out.zero_(), lens
        with torch.cuda.device(x.device):
            _lib.check(_lib.load().wrnn_lr_regulate_static(
                x.contiguous().data_ptr(), dur.data_ptr(), B, T, C, out.data_ptr(), max_len, lens.data_ptr(),
                torch.cuda.current_stream(x.device).cuda_stream), "wrnn_lr_regulate_static")
        return out, lens
