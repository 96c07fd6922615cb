"""Host-side mirror of the bidirectional ``nn.LSTM`` that follows the length regulator in the
reference's forward models (models/forward_tacotron.py:96-99 ``self.lstm``; called packed at
:147-152 and padded at :224; multispeaker variant models/multi_forward_tacotron.py:164-168,294).
SURVEY.md section 8 row f1.

``BiLSTM(input_size, hidden_size)`` keeps torch.nn.LSTM's parameter names (``weight_ih_l0`` ...
``bias_hh_l0_reverse``), so the ``lstm.*`` entries of a reference checkpoint load unchanged, and its
call convention: a padded ``[B,M,in]`` tensor or a ``PackedSequence`` in, ``(output, (h_n, c_n))``
out.  The recurrence runs in libwrnn_mi355.so; there is no CPU fallback.  Forward/inference only.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional

import torch
import torch.nn as nn
from torch.nn.utils.rnn import PackedSequence, pack_padded_sequence, pad_packed_sequence

from . import _lib


class BiLSTM(nn.Module):

    def __init__(self, input_size: int, hidden_size: int, batch_first: bool = True, bidirectional: bool = True):
        super().__init__()
        if not (batch_first and bidirectional):
            raise ValueError("BiLSTM mirrors nn.LSTM(..., batch_first=True, bidirectional=True) only")
        self.input_size, self.hidden_size = input_size, hidden_size
        k = 1.0 / math.sqrt(hidden_size)
        for _, name in _lib.BILSTM_FIELDS:   # nn.LSTM's own registration order and init (U(-k, k))
            rows = 4 * hidden_size
            shape = (rows, input_size) if 'weight_ih' in name else (rows, hidden_size) if 'weight_hh' in name else (rows,)
            self.register_parameter(name, nn.Parameter(torch.empty(shape).uniform_(-k, k)))
        self._handle = None
        self._handle_key = None
        self._path = "auto"

    # ------------------------------------------------------------------ handle management
    def _key(self):
        p = self.weight_ih_l0
        return (p.device, tuple((q.data_ptr(), q._version) for q in self.parameters()))

    def _get_handle(self):
        dev = self.weight_ih_l0.device
        if dev.type != 'cuda':
            raise RuntimeError("wavernn_amd.BiLSTM runs only on an MI355X (HIP) device; there is no CPU fallback")
        key = self._key()
        if self._handle is not None and key == self._handle_key:
            return self._handle
        self._free_handle()
        lib = _lib.load()
        host = {name: getattr(self, name).detach().to('cpu', torch.float32).contiguous() for _, name in _lib.BILSTM_FIELDS}
        w = _lib.BiLstmWeights(**{f: host[name].data_ptr() for f, name in _lib.BILSTM_FIELDS})
        h = C.c_void_p()
        _lib.check(lib.wrnn_bilstm_create(self.input_size, self.hidden_size, C.byref(w),
                                          dev.index if dev.index is not None else torch.cuda.current_device(),
                                          C.byref(h)), "wrnn_bilstm_create")
        _lib.check(lib.wrnn_bilstm_set_path(h, _lib.FREE_PATHS[self._path]), "wrnn_bilstm_set_path")
        self._handle, self._handle_key = h, key
        return h

    def set_path(self, path: str):
        """'launches' (one launch per frame), 'resident' (one persistent weights-in-registers launch per call: in = H = 512,
        B <= 64) or 'auto' (resident where it applies).  Results agree to fp32 summation order."""
        if path not in _lib.FREE_PATHS:
            raise ValueError(f"unknown BiLSTM path {path!r}; one of {sorted(_lib.FREE_PATHS)}")
        self._path = path
        if self._handle is not None:
            _lib.check(_lib.load().wrnn_bilstm_set_path(self._handle, _lib.FREE_PATHS[path]), "wrnn_bilstm_set_path")

    def _free_handle(self):
        if getattr(self, '_handle', None) is not None:
            try:
                _lib.load().wrnn_bilstm_destroy(self._handle)
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

    def flatten_parameters(self):  # nn.LSTM API, nothing to do here
        return None

    # ------------------------------------------------------------------ forward
    def run_padded(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None, padding_value: float = 0.0):
        """x [B,M,in] (+ per-row lengths for packed semantics) -> (out [B,M,2H], (h_n, c_n) [2,B,H])."""
        h = self._get_handle()
        dev = self.weight_ih_l0.device
        if x.device != dev:
            raise RuntimeError(f"input is on {x.device}, the module on {dev}")
        if x.dtype != torch.float32:
            raise TypeError("BiLSTM expects float32 input (the reference path is fp32)")
        if x.dim() != 3 or x.shape[2] != self.input_size:
            raise ValueError(f"expected x [B,M,{self.input_size}], got {tuple(x.shape)}")
        B, M, _ = x.shape
        lens = None
        if lengths is not None:
            lc = torch.as_tensor(lengths).to('cpu')
            if lc.numel() != B or int(lc.min()) < 1 or int(lc.max()) > M:
                raise ValueError("lengths must hold B values in [1, M] (pack_padded_sequence's own rule)")
            lens = lc.to(device=dev, dtype=torch.int32)
        xc = x.contiguous()
        H = self.hidden_size
        out = torch.empty(B, M, 2 * H, device=dev)
        h_n = torch.empty(2, B, H, device=dev)
        c_n = torch.empty(2, B, H, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(_lib.load().wrnn_bilstm_forward(
                h, xc.data_ptr(), None if lens is None else lens.data_ptr(), B, M, float(padding_value),
                out.data_ptr(), h_n.data_ptr(), c_n.data_ptr(), stream), "wrnn_bilstm_forward")
        return out, (h_n, c_n)

    def forward(self, x, hx=None):
        if hx is not None:
            raise NotImplementedError("the reference always starts from the zero state")
        if isinstance(x, PackedSequence):
            padded, lengths = pad_packed_sequence(x, batch_first=True)
            out, (h_n, c_n) = self.run_padded(padded, lengths)
            packed = pack_padded_sequence(out, lengths, batch_first=True, enforce_sorted=False)
            # pad_packed_sequence returned the rows in original order; hand back a PackedSequence that
            # unpacks to the same order (what nn.LSTM does through sorted/unsorted indices)
            return packed, (h_n, c_n)
        return self.run_padded(x)
