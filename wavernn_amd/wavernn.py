"""WaveRNN vocoder with the per-sample generation loop in libwrnn_mi355.so (SURVEY.md section 8 rows
a9 / f4).

**Build-defined, parity unpinned.**  The mounted reference commit holds no WaveRNN: gen_forward.py accepts
only griffinlim / melgan / hifigan (gen_forward.py:36-37) and there is no model file.  This module follows
the fatchord WaveRNN the reference forked from (reference README.md:4) with the hyper-parameters that
survive in tests/resources/test_config.yaml:47-75, and keeps the interface BASELINE.json's north star
names: ``generate(mels, batched, target, overlap)``.  The upsampling network (MelResNet + stretch layers)
and the fold / crossfade bookkeeping are per-utterance work on stock PyTorch ops; the sample-by-sample loop
is one call into the C ABI (``wrnn_wavernn_generate``).  RAW mode (softmax over 2**bits classes,
inverse-CDF sampling) and MOL mode (mixture of logistics, sampled as the reference's utils/distribution.py:87-123
does -- the one pinned piece); uniforms are caller-supplied (or torch.rand), so runs are reproducible.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


class ResBlock(nn.Module):
    def __init__(self, dims: int):
        super().__init__()
        self.conv1 = nn.Conv1d(dims, dims, kernel_size=1, bias=False)
        self.conv2 = nn.Conv1d(dims, dims, kernel_size=1, bias=False)
        self.batch_norm1 = nn.BatchNorm1d(dims)
        self.batch_norm2 = nn.BatchNorm1d(dims)

    def forward(self, x):
        r = x
        x = F.relu(self.batch_norm1(self.conv1(x)))
        return self.batch_norm2(self.conv2(x)) + r


class MelResNet(nn.Module):
    def __init__(self, res_blocks, in_dims, compute_dims, res_out_dims, pad):
        super().__init__()
        self.conv_in = nn.Conv1d(in_dims, compute_dims, kernel_size=2 * pad + 1, bias=False)
        self.batch_norm = nn.BatchNorm1d(compute_dims)
        self.layers = nn.ModuleList(ResBlock(compute_dims) for _ in range(res_blocks))
        self.conv_out = nn.Conv1d(compute_dims, res_out_dims, kernel_size=1)

    def forward(self, x):
        x = F.relu(self.batch_norm(self.conv_in(x)))
        for f in self.layers:
            x = f(x)
        return self.conv_out(x)


class Stretch2d(nn.Module):
    def __init__(self, x_scale: int, y_scale: int):
        super().__init__()
        self.x_scale, self.y_scale = x_scale, y_scale

    def forward(self, x):
        b, c, h, w = x.size()
        x = x.unsqueeze(-1).unsqueeze(3).repeat(1, 1, 1, self.y_scale, 1, self.x_scale)
        return x.view(b, c, h * self.y_scale, w * self.x_scale)


class UpsampleNetwork(nn.Module):
    def __init__(self, feat_dims, upsample_scales, compute_dims, res_blocks, res_out_dims, pad):
        super().__init__()
        total_scale = int(np.prod(upsample_scales))
        self.indent = pad * total_scale
        self.resnet = MelResNet(res_blocks, feat_dims, compute_dims, res_out_dims, pad)
        self.resnet_stretch = Stretch2d(total_scale, 1)
        self.up_layers = nn.ModuleList()
        for scale in upsample_scales:
            conv = nn.Conv2d(1, 1, kernel_size=(1, scale * 2 + 1), padding=(0, scale), bias=False)
            conv.weight.data.fill_(1.0 / (scale * 2 + 1))
            self.up_layers.append(Stretch2d(scale, 1))
            self.up_layers.append(conv)

    def forward(self, m):
        aux = self.resnet_stretch(self.resnet(m).unsqueeze(1)).squeeze(1)
        m = m.unsqueeze(1)
        for f in self.up_layers:
            m = f(m)
        m = m.squeeze(1)[:, :, self.indent:-self.indent]
        return m.transpose(1, 2), aux.transpose(1, 2)


class WaveRNN(nn.Module):

    def __init__(self, rnn_dims: int, fc_dims: int, bits: int, pad: int, upsample_factors: Sequence[int],
                 feat_dims: int, compute_dims: int, res_out_dims: int, res_blocks: int, hop_length: int,
                 sample_rate: int, mode: str = 'RAW'):
        super().__init__()
        if mode not in ('RAW', 'MOL'):
            raise RuntimeError(f"Unknown model mode value - {mode}")
        if int(np.prod(upsample_factors)) != hop_length:
            raise ValueError("upsample_factors must factorise hop_length")
        self.mode, self.pad = mode, pad
        # RAW: softmax over 2**bits classes; MOL: 10 logistic mixture components x (logit, mean, log-scale), sampled as
        # the reference's sample_from_discretized_mix_logistic does (utils/distribution.py:87-123)
        self.n_classes = 2 ** bits if mode == 'RAW' else 30
        self.rnn_dims, self.fc_dims, self.feat_dims = rnn_dims, fc_dims, feat_dims
        self.aux_dims = res_out_dims // 4
        self.hop_length, self.sample_rate = hop_length, sample_rate
        self.upsample = UpsampleNetwork(feat_dims, upsample_factors, compute_dims, res_blocks, res_out_dims, pad)
        self.I = nn.Linear(feat_dims + self.aux_dims + 1, rnn_dims)
        self.rnn1 = nn.GRU(rnn_dims, rnn_dims, batch_first=True)
        self.rnn2 = nn.GRU(rnn_dims + self.aux_dims, rnn_dims, batch_first=True)
        self.fc1 = nn.Linear(rnn_dims + self.aux_dims, fc_dims)
        self.fc2 = nn.Linear(fc_dims + self.aux_dims, fc_dims)
        self.fc3 = nn.Linear(fc_dims, self.n_classes)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self._handle = None
        self._handle_key = None
        self._path = 'auto'

    # ------------------------------------------------------------------ handle management
    def _loop_params(self):
        mods = dict(self.named_modules())
        return [getattr(mods[name.rsplit('.', 1)[0]], name.rsplit('.', 1)[1]) for _, name in _lib.WAVERNN_FIELDS]

    def _get_handle(self):
        dev = self.I.weight.device
        if dev.type != 'cuda':
            raise RuntimeError("wavernn_amd.WaveRNN generates only on an MI355X (HIP) device; there is no CPU fallback")
        params = self._loop_params()
        key = (dev, tuple((p.data_ptr(), p._version) for p in params))
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self._free_handle()
        host = [p.detach().to('cpu', torch.float32).contiguous() for p in params]
        w = _lib.WaveRnnWeights(**{f: t.data_ptr() for (f, _), t in zip(_lib.WAVERNN_FIELDS, host)})
        dims = _lib.WaveRnnDims(self.rnn_dims, self.fc_dims, self.feat_dims, self.aux_dims, self.n_classes,
                                _lib.WAVERNN_MODES[self.mode])
        h = C.c_void_p()
        _lib.check(_lib.load().wrnn_wavernn_create(C.byref(dims), C.byref(w),
                                                   dev.index if dev.index is not None else torch.cuda.current_device(),
                                                   C.byref(h)), "wrnn_wavernn_create")
        self._handle, self._handle_key = h, key
        _lib.check(_lib.load().wrnn_wavernn_set_path(h, _lib.FREE_PATHS[self._path]), "wrnn_wavernn_set_path")
        return h

    def set_path(self, path: str):
        """How the sample loop runs: 'auto' (one persistent weights-resident launch per call for up to 32 rows -- unbatched
        generation and the folds of an utterance -- where the model has the kernels' shape, five launches per sample
        otherwise), 'launches', or 'resident' (calls that the kernels cannot serve raise)."""
        if path not in _lib.FREE_PATHS:
            raise ValueError(f"unknown path {path!r}; one of {sorted(_lib.FREE_PATHS)}")
        self._path = path
        if self._handle is not None:
            _lib.check(_lib.load().wrnn_wavernn_set_path(self._handle, _lib.FREE_PATHS[path]), "wrnn_wavernn_set_path")

    def _free_handle(self):
        if getattr(self, '_handle', None) is not None:
            try:
                _lib.load().wrnn_wavernn_destroy(self._handle)
            except Exception:
                pass
            self._handle = None
            self._handle_key = None

    def __del__(self):
        try:
            self._free_handle()
        except Exception:   # interpreter teardown: module globals may already be gone
            pass

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_handle'] = None
        state['_handle_key'] = None
        return state

    # ------------------------------------------------------------------ the loop
    def generate_loop(self, mels_up: torch.Tensor, aux: torch.Tensor, uniforms: Optional[torch.Tensor] = None,
                      x_forced: Optional[torch.Tensor] = None, return_logits: bool = False):
        """mels_up [B, seq, feat], aux [B, seq, 4 * aux_dims] -> samples [B, seq] (and logits [B, seq, C])."""
        h = self._get_handle()
        dev = self.I.weight.device
        B, seq, _ = mels_up.shape
        if aux.shape != (B, seq, 4 * self.aux_dims) or mels_up.shape[2] != self.feat_dims:
            raise ValueError("expected mels_up [B, seq, feat_dims] and aux [B, seq, 4 * aux_dims]")
        if self.mode == 'MOL':
            from .distribution import draw_uniforms
            ushape = (B, seq, self.n_classes // 3 + 1)     # nr_mix Gumbel-max uniforms + the logistic one, per step
            u = draw_uniforms(ushape, dev) if uniforms is None else uniforms
        else:
            ushape = (B, seq)
            u = torch.rand(B, seq, device=dev) if uniforms is None else uniforms
        if tuple(u.shape) != ushape:
            raise ValueError(f"uniforms must have shape {ushape} in {self.mode} mode, got {tuple(u.shape)}")
        tens = [t.to(dev, torch.float32).contiguous() for t in (mels_up, aux, u)]
        xf = None if x_forced is None else x_forced.to(dev, torch.float32).contiguous()
        out = torch.empty(B, seq, device=dev)
        logits = torch.empty(B, seq, self.n_classes, device=dev) if return_logits else None
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().wrnn_wavernn_generate(
                h, tens[0].data_ptr(), tens[1].data_ptr(), tens[2].data_ptr(), B, seq,
                None if xf is None else xf.data_ptr(), out.data_ptr(),
                None if logits is None else logits.data_ptr(), stream), "wrnn_wavernn_generate")
        return (out, logits) if return_logits else out

    # ------------------------------------------------------------------ whole-utterance generation
    @torch.no_grad()
    def generate(self, mels, batched: bool = True, target: int = 11000, overlap: int = 550, mu_law: bool = False,
                 noise: Optional[torch.Tensor] = None) -> np.ndarray:
        """mels [1, feat_dims, frames] (tensor or array) -> audio [(frames - 1) * hop] in [-1, 1].
        ``noise``: uniforms [folds, steps] (RAW) or [folds, steps, 11] (MOL) for the sampler (drawn when omitted)."""
        if mu_law:
            raise NotImplementedError("mu-law decoding is not part of the RAW-mode loop built here")
        self.eval()
        dev = self.I.weight.device
        mels = torch.as_tensor(mels, dtype=torch.float32, device=dev)
        wave_len = (mels.size(-1) - 1) * self.hop_length
        mels_up, aux = self.upsample(F.pad(mels, (self.pad, self.pad)))
        if batched:
            mels_up = self.fold_with_overlap(mels_up, target, overlap)
            aux = self.fold_with_overlap(aux, target, overlap)
        out = self.generate_loop(mels_up, aux, noise).cpu().numpy().astype(np.float64)
        out = self.xfade_and_unfold(out, overlap) if batched else out[0]
        fade_out = np.linspace(1, 0, 20 * self.hop_length)
        out = out[:wave_len]
        out[-20 * self.hop_length:] *= fade_out
        return out

    @torch.no_grad()
    def generate_batch(self, mels_list, target: int = 11000, overlap: int = 550,
                       noise: Optional[Sequence[torch.Tensor]] = None, max_rows: int = 256):
        """Several utterances per loop call (build-defined, for the many-utterance workload of BASELINE
        configs[3]): the folds of all utterances are the batch rows of ONE generation loop, which costs
        little more per step than one utterance's folds.  Returns what ``generate(m, batched=True)`` returns
        for every ``m`` (bit for bit when given the same ``noise`` rows: folds are independent rows).
        ``max_rows`` bounds the rows per loop call (workspace: rows * steps * (feat + 4 aux + rnn) floats)."""
        self.eval()
        dev = self.I.weight.device
        folded, meta = [], []
        for m in mels_list:
            m = torch.as_tensor(m, dtype=torch.float32, device=dev)
            mu, au = self.upsample(F.pad(m, (self.pad, self.pad)))
            mu, au = self.fold_with_overlap(mu, target, overlap), self.fold_with_overlap(au, target, overlap)
            folded.append((mu, au))
            meta.append(((m.size(-1) - 1) * self.hop_length, mu.shape[0]))
        outs = [None] * len(folded)
        i = 0
        while i < len(folded):       # greedy groups of utterances with at most max_rows folds
            j, rows = i, 0
            while j < len(folded) and (j == i or rows + meta[j][1] <= max_rows):
                rows += meta[j][1]
                j += 1
            mu = torch.cat([folded[k][0] for k in range(i, j)], dim=0)
            au = torch.cat([folded[k][1] for k in range(i, j)], dim=0)
            u = None if noise is None else torch.cat([noise[k].to(dev) for k in range(i, j)], dim=0)
            y = self.generate_loop(mu, au, u).cpu().numpy().astype(np.float64)
            r0 = 0
            for k in range(i, j):
                wave_len, nf = meta[k]
                o = self.xfade_and_unfold(y[r0:r0 + nf], overlap)[:wave_len]
                o[-20 * self.hop_length:] *= np.linspace(1, 0, 20 * self.hop_length)
                outs[k] = o
                r0 += nf
            i = j
        return outs

    @staticmethod
    def fold_with_overlap(x: torch.Tensor, target: int, overlap: int) -> torch.Tensor:
        """[1, L, C] -> [folds, target + 2 * overlap, C]; consecutive folds share `overlap` steps."""
        _, total_len, features = x.size()
        num_folds = (total_len - overlap) // (target + overlap)
        extended_len = num_folds * (overlap + target) + overlap
        remaining = total_len - extended_len
        if remaining != 0:
            num_folds += 1
            x = F.pad(x, (0, 0, 0, target + 2 * overlap - remaining))
        return x[0].unfold(0, target + 2 * overlap, target + overlap).permute(0, 2, 1).contiguous()[:num_folds]

    @staticmethod
    def xfade_and_unfold(y: np.ndarray, overlap: int) -> np.ndarray:
        num_folds, length = y.shape
        target = length - 2 * overlap
        total_len = num_folds * (target + overlap) + overlap
        silence_len = overlap // 2
        fade_len = overlap - silence_len
        t = np.linspace(-1, 1, fade_len, dtype=np.float64)
        fade_in = np.concatenate([np.zeros(silence_len), np.sqrt(0.5 * (1 + t))])
        fade_out = np.concatenate([np.ones(silence_len), np.sqrt(0.5 * (1 - t))])
        y = y.copy()
        y[:, :overlap] *= fade_in
        y[:, -overlap:] *= fade_out
        unfolded = np.zeros(total_len, dtype=np.float64)
        for i in range(num_folds):
            start = i * (target + overlap)
            unfolded[start:start + target + 2 * overlap] += y[i]
        return unfolded

    def get_step(self):
        return self.step.data.item()
