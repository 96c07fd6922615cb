"""Host-side mirror of the reference decoder interface (models/tacotron.py:86-235).

``Decoder`` has the reference's constructor, buffer ``r``, ``max_r``, parameter names
(state_dict keys ``prenet.fc1.weight`` ... ``mel_proj.weight``) and ``forward`` signature and
return convention, so a reference checkpoint loads with ``load_state_dict`` and
``Tacotron.forward``'s loop (tacotron.py:339-345) can call it unchanged.  All arithmetic runs
in libwrnn_mi355.so on an MI355X; the submodules below are parameter containers only.
There is no CPU path: tensors must live on a HIP device.

Beyond the reference's per-step ``forward`` the class offers the whole loop as one library
call (``decode_teacher_forced`` = the loop of Tacotron.forward, ``decode_free`` = the
build-defined free-running decode).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib

PRENET_P = 0.5


class PreNet(nn.Module):
    """Parameter container for reference PreNet (tacotron.py:86-91); ``.training`` gates dropout."""

    def __init__(self, in_dims, fc1_dims=256, fc2_dims=128, dropout=0.5):
        super().__init__()
        self.fc1 = nn.Linear(in_dims, fc1_dims)
        self.fc2 = nn.Linear(fc1_dims, fc2_dims)
        self.p = dropout


class LSA(nn.Module):
    """Parameter + state container for reference LSA (tacotron.py:122-136)."""

    def __init__(self, attn_dim, kernel_size=31, filters=32):
        super().__init__()
        self.conv = nn.Conv1d(2, filters, padding=(kernel_size - 1) // 2, kernel_size=kernel_size, bias=False)
        self.L = nn.Linear(filters, attn_dim, bias=True)
        self.W = nn.Linear(attn_dim, attn_dim, bias=True)
        self.v = nn.Linear(attn_dim, 1, bias=False)
        self.cumulative = None
        self.attention = None


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _f32c(t: torch.Tensor, name: str, device) -> torch.Tensor:
    if t.device != device:
        raise RuntimeError(f"{name} is on {t.device}, the decoder's weights are on {device}")
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32 (the reference path is fp32), got {t.dtype}")
    return t.contiguous()


def _refresh_hook(module, incompatible_keys):
    """load_state_dict post hook (a module-level function: a local lambda in the module's __dict__ cannot be pickled)."""
    module.refresh_weights()


class Decoder(nn.Module):
    max_r = 20  # tacotron.py:162

    def __init__(self, n_mels: int, decoder_dims: int, lstm_dims: int, aligner_out_dims: int):
        super().__init__()
        self.register_buffer('r', torch.tensor(1, dtype=torch.int))
        self.n_mels = n_mels
        self.decoder_dims = decoder_dims
        self.lstm_dims = lstm_dims
        self.aligner_out_dims = aligner_out_dims
        self.prenet = PreNet(n_mels + aligner_out_dims)
        self.attn_net = LSA(decoder_dims)
        self.attn_rnn = nn.GRUCell(decoder_dims + decoder_dims // 2, decoder_dims)
        self.rnn_input = nn.Linear(2 * decoder_dims, lstm_dims)
        self.res_rnn1 = nn.LSTMCell(lstm_dims, lstm_dims)
        self.res_rnn2 = nn.LSTMCell(lstm_dims, lstm_dims)
        self.mel_proj = nn.Linear(lstm_dims, n_mels * self.max_r, bias=False)
        self._handle = None
        self._handle_key = None
        self._dirty = True
        self._r_cache = None
        self._free_path = "auto"
        self._tf_path = "auto"
        self._version_sum = -1
        self.register_load_state_dict_post_hook(_refresh_hook)

    # ------------------------------------------------------------------ handle management
    def _device(self) -> torch.device:
        return self.mel_proj.weight.device

    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def refresh_weights(self) -> None:
        """Marks the library's copy of the weights stale; the next call uploads them again.  Done automatically by
        ``load_state_dict`` and by ``.to() / .cuda()``; call it after editing parameters in place under
        ``torch.no_grad()`` (with autograd enabled every call compares the parameters' version counters itself)."""
        self._dirty = True

    def _apply(self, fn, *args, **kwargs):
        self._dirty = True
        return super()._apply(fn, *args, **kwargs)

    def _get_handle(self):
        # hot call: with autograd off (inference) the handle is reused until refresh_weights() or an in-place edit of a
        # parameter (p.mul_(), p.data.copy_(): the version counters move; their sum is a few attribute reads, the full
        # data_ptr / _version key below costs more host time than the enqueue of a decoder step)
        if self._handle is not None and not self._dirty and not torch.is_grad_enabled():
            if sum(p._version for p in self.parameters()) == self._version_sum:
                return self._handle
        dev = self._device()
        if dev.type != 'cuda':
            raise RuntimeError(
                "wavernn_amd.Decoder runs only on an MI355X (HIP) device; move the module with "
                ".cuda() -- there is no CPU fallback")
        key = (dev.index, self._weights_key())
        self._dirty = False
        self._version_sum = sum(p._version for p in self.parameters())
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self._free_handle()
        lib = _lib.load()
        sd = self.state_dict()
        keep = []
        hw = _lib.DecoderWeights()
        for field, name in _lib.WEIGHT_FIELDS:
            t = sd[name].detach().to('cpu', torch.float32).contiguous()
            keep.append(t)
            setattr(hw, field, t.data_ptr())
        dims = _lib.DecoderDims(self.n_mels, self.decoder_dims, self.lstm_dims, self.aligner_out_dims)
        h = C.c_void_p()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        _lib.check(lib.wrnn_decoder_create(C.byref(dims), C.byref(hw), idx, C.byref(h)), "wrnn_decoder_create")
        self._handle, self._handle_key = h, key
        _lib.check(lib.wrnn_decoder_set_free_path(h, _lib.FREE_PATHS[self._free_path]), "wrnn_decoder_set_free_path")
        _lib.check(lib.wrnn_decoder_set_tf_path(h, _lib.TF_PATHS[self._tf_path]), "wrnn_decoder_set_tf_path")
        return h

    def set_tf_path(self, path: str) -> None:
        """How ``decode_teacher_forced`` runs its loop: 'launches' (two launches per step, any shape), 'chain' (the
        GRU -> query -> attention recurrence as one persistent launch with the LSTM stack's launches beside it;
        B <= 64 at T <= 128, B <= 32 at T <= 256) or 'auto' (chain where it applies).  Results agree to fp32 rounding."""
        if path not in _lib.TF_PATHS:
            raise ValueError(f"unknown teacher-forced path {path!r}; one of {sorted(_lib.TF_PATHS)}")
        self._tf_path = path
        if self._handle is not None:
            _lib.check(_lib.load().wrnn_decoder_set_tf_path(self._handle, _lib.TF_PATHS[path]), "wrnn_decoder_set_tf_path")

    def set_free_path(self, path: str) -> None:
        """How ``decode_free`` runs: 'launches' (seven launches per step, any shape), 'resident' (one persistent
        weights-resident launch per call; B <= 4, B*T <= 1024, lstm_dims 512) or 'auto' (resident where it applies)."""
        if path not in _lib.FREE_PATHS:
            raise ValueError(f"unknown free-running path {path!r}; one of {sorted(_lib.FREE_PATHS)}")
        self._free_path = path
        if self._handle is not None:
            _lib.check(_lib.load().wrnn_decoder_set_free_path(self._handle, _lib.FREE_PATHS[path]), "wrnn_decoder_set_free_path")

    def _free_handle(self):
        if self._handle is not None:
            try:
                _lib.load().wrnn_decoder_destroy(self._handle)
            finally:
                self._handle = None
                self._handle_key = None

    def __del__(self):
        try:
            self._free_handle()
        except Exception:
            pass

    def __getstate__(self):
        # the library handle is per-process device state: never copied or pickled
        state = self.__dict__.copy()
        state['_handle'] = None
        state['_handle_key'] = None
        state['_dirty'] = True
        state['_r_cache'] = None
        return state

    def _r_value(self) -> int:
        # reading the int32 buffer is a device->host sync; do it once per rebind/update of self.r
        # (Tacotron.r's setter rebinds decoder.r to a new tensor, tacotron.py:286-288)
        key = (id(self.r), self.r._version)
        if self._r_cache is None or self._r_cache[0] != key:
            self._r_cache = (key, int(self.r))
        return self._r_cache[1]

    def _stream(self):
        return torch.cuda.current_stream(self._device()).cuda_stream

    # ------------------------------------------------------------------ reference interface
    def forward(self, encoder_seq, encoder_seq_proj, prenet_in, hidden_states, cell_states, context_vec, t,
                noise: Optional[dict] = None):
        """One decoder step; arguments and returns as reference Decoder.forward (tacotron.py:185-235).

        ``noise`` (optional, build extension) supplies the uniforms explicitly:
        {'drop_u1': [B,256], 'drop_u2': [B,128], 'zone_u1': [B,l], 'zone_u2': [B,l]}; without it
        they are drawn with torch.rand when prenet.training / self.training ask for them."""
        h = self._get_handle()
        dev = self._device()
        lib = _lib.load()
        enc_q = _f32c(encoder_seq, 'encoder_seq', dev)
        enc_p = _f32c(encoder_seq_proj, 'encoder_seq_proj', dev)
        B, T, _ = enc_p.shape
        pin = _f32c(prenet_in, 'prenet_in', dev)
        attn_h, h1, h2 = (_f32c(x, 'hidden_state', dev) for x in hidden_states)
        c1, c2 = (_f32c(x, 'cell_state', dev) for x in cell_states)
        ctx = _f32c(context_vec.reshape(B, -1), 'context_vec', dev)
        r = self._r_value()
        t = int(t)
        lsa = self.attn_net
        if t == 0 or lsa.cumulative is None:
            if t != 0:
                raise RuntimeError("Decoder.forward called with t > 0 before t == 0 (LSA state missing)")
            lsa.cumulative = torch.empty(B, T, device=dev)   # zero-initialised by the kernel at t == 0
            lsa.attention = torch.empty(B, T, device=dev)
        noise = noise or {}
        du1, du2 = noise.get('drop_u1'), noise.get('drop_u2')
        zu1, zu2 = noise.get('zone_u1'), noise.get('zone_u2')
        if self.prenet.training and du1 is None:
            du1 = torch.rand(B, self.prenet.fc1.out_features, device=dev)
            du2 = torch.rand(B, self.prenet.fc2.out_features, device=dev)
        if self.training and zu1 is None:
            zu1 = torch.rand(B, self.lstm_dims, device=dev)
            zu2 = torch.rand(B, self.lstm_dims, device=dev)
        opt = [None if x is None else _f32c(x, 'noise', dev) for x in (du1, du2, zu1, zu2)]
        new = [torch.empty_like(x) for x in (attn_h, h1, h2, c1, c2, ctx)]
        # the reference rebinds self.attention each step and accumulates into self.cumulative
        # (tacotron.py:153-154); a caller may hold the returned scores, so they get fresh storage
        mels = torch.empty(B, self.n_mels, r, device=dev)
        scores = torch.empty(B, 1, T, device=dev)
        st_in = _lib.DecoderState(*map(_ptr, (attn_h, h1, h2, c1, c2, ctx, lsa.cumulative, lsa.attention)))
        st_out = _lib.DecoderState(*map(_ptr, (*new, lsa.cumulative, lsa.attention)))
        _lib.check(lib.wrnn_decoder_step(h, _ptr(enc_q), _ptr(enc_p), _ptr(pin), C.byref(st_in), C.byref(st_out),
                                         B, T, t, r, *map(_ptr, opt), _ptr(mels), _ptr(scores), self._stream()),
                   "wrnn_decoder_step")
        return mels, scores, (new[0], new[1], new[2]), (new[3], new[4]), new[5]

    # ------------------------------------------------------------------ whole-loop entry points
    def decode_teacher_forced(self, encoder_seq, encoder_seq_proj, mel_in, noise: Optional[dict] = None
                              ) -> Tuple[torch.Tensor, torch.Tensor]:
        """The decoder loop of Tacotron.forward (tacotron.py:299-316,334-348,356) in one call.

        mel_in [B, n_mels + aligner_out_dims, S] -> (mel [B,n_mels,n*r], att [B,n,T]), n = ceil(S/r).
        Prenet dropout follows ``self.prenet.training`` (uniforms 'drop_u1' [n,B,256] / 'drop_u2' [n,B,128] via
        ``noise``), zoneout on both LSTM hidden states follows ``self.training`` ('zone_u1' / 'zone_u2' [n,B,l]),
        as in the reference's training-mode forward (tacotron.py:215-226)."""
        h = self._get_handle()
        dev = self._device()
        enc_q = _f32c(encoder_seq, 'encoder_seq', dev)
        enc_p = _f32c(encoder_seq_proj, 'encoder_seq_proj', dev)
        mel_in = _f32c(mel_in, 'mel_in', dev)
        B, T, _ = enc_p.shape
        S = mel_in.shape[2]
        if mel_in.shape[1] != self.n_mels + self.aligner_out_dims:
            raise ValueError(f"mel_in has {mel_in.shape[1]} channels, expected n_mels + aligner_out_dims = "
                             f"{self.n_mels + self.aligner_out_dims}")
        r = self._r_value()
        n = (S + r - 1) // r
        noise = noise or {}
        du1, du2 = noise.get('drop_u1'), noise.get('drop_u2')
        if self.prenet.training and du1 is None:
            du1 = torch.rand(n, B, self.prenet.fc1.out_features, device=dev)
            du2 = torch.rand(n, B, self.prenet.fc2.out_features, device=dev)
        zu1, zu2 = noise.get('zone_u1'), noise.get('zone_u2')
        if self.training and zu1 is None:
            zu1 = torch.rand(n, B, self.lstm_dims, device=dev)
            zu2 = torch.rand(n, B, self.lstm_dims, device=dev)
        opt = [None if x is None else _f32c(x, 'noise', dev) for x in (du1, du2, zu1, zu2)]
        for x, shape in zip(opt, ((n, B, self.prenet.fc1.out_features), (n, B, self.prenet.fc2.out_features),
                                  (n, B, self.lstm_dims), (n, B, self.lstm_dims))):
            if x is not None and tuple(x.shape) != shape:
                raise ValueError(f"noise tensor has shape {tuple(x.shape)}, expected {shape}")
        mel = torch.empty(B, self.n_mels, n * r, device=dev)
        att = torch.empty(B, n, T, device=dev)
        _lib.check(_lib.load().wrnn_decoder_run_teacher_forced(
            h, _ptr(enc_q), _ptr(enc_p), _ptr(mel_in), B, T, S, r, *map(_ptr, opt), _ptr(mel), _ptr(att),
            self._stream()), "wrnn_decoder_run_teacher_forced")
        return mel, att

    def profile_teacher_forced(self, encoder_seq, encoder_seq_proj, mel_in) -> dict:
        """One teacher-forced pass with every launch bracketed by HIP events (measurement aid):
        {launch kind: {'ms': summed duration, 'launches': count, 'avg_us': ...}}."""
        h = self._get_handle()
        lib = _lib.load()
        _lib.check(lib.wrnn_decoder_set_profiling(h, 1), "wrnn_decoder_set_profiling")
        try:
            self.decode_teacher_forced(encoder_seq, encoder_seq_proj, mel_in)
        finally:
            _lib.check(lib.wrnn_decoder_set_profiling(h, 0), "wrnn_decoder_set_profiling")
        n = len(_lib.PROFILE_KINDS)
        ms, cnt = (C.c_float * n)(), (C.c_int * n)()
        _lib.check(lib.wrnn_decoder_get_profile(h, ms, cnt), "wrnn_decoder_get_profile")
        return {k: {"ms": float(ms[i]), "launches": int(cnt[i]),
                    "avg_us": 1e3 * float(ms[i]) / max(int(cnt[i]), 1)} for i, k in enumerate(_lib.PROFILE_KINDS)}

    def decode_free(self, encoder_seq, encoder_seq_proj, max_steps: int, aligner_feats=None,
                    stop_threshold: Optional[float] = None, noise: Optional[dict] = None):
        """Free-running decode (build-defined): returns (mel [B,n_mels,n*r], att [B,n,T], n)."""
        h = self._get_handle()
        dev = self._device()
        enc_q = _f32c(encoder_seq, 'encoder_seq', dev)
        enc_p = _f32c(encoder_seq_proj, 'encoder_seq_proj', dev)
        B, T, _ = enc_p.shape
        r = self._r_value()
        ali = None
        if aligner_feats is not None:
            ali = _f32c(aligner_feats, 'aligner_feats', dev)
            if tuple(ali.shape) != (B, self.aligner_out_dims, max_steps):
                raise ValueError(f"aligner_feats must be [B, {self.aligner_out_dims}, max_steps]")
        noise = noise or {}
        du1, du2 = noise.get('drop_u1'), noise.get('drop_u2')
        if self.prenet.training and du1 is None:
            du1 = torch.rand(max_steps, B, self.prenet.fc1.out_features, device=dev)
            du2 = torch.rand(max_steps, B, self.prenet.fc2.out_features, device=dev)
        opt = [None if x is None else _f32c(x, 'noise', dev) for x in (du1, du2)]
        mel = torch.empty(B, self.n_mels, max_steps * r, device=dev)
        att = torch.empty(B, max_steps, T, device=dev)
        n_out = C.c_int(0)
        _lib.check(_lib.load().wrnn_decoder_run_free(
            h, _ptr(enc_q), _ptr(enc_p), _ptr(ali), B, T, int(max_steps), r,
            float(stop_threshold if stop_threshold is not None else 0.0), int(stop_threshold is not None),
            *map(_ptr, opt), _ptr(mel), _ptr(att), C.byref(n_out), self._stream()), "wrnn_decoder_run_free")
        n = n_out.value
        return mel[:, :, :n * r], att[:, :n, :], n
