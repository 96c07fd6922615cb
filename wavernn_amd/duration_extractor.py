"""Host-side mirror of the reference DurationExtractor (duration_extraction/duration_extractor.py:11-84),
SURVEY.md section 8 row f3.

Same constructor ``DurationExtractor(silence_threshold, silence_prob_shift)`` and the same call
``extractor(x=tokens[T], mel=[n_mels, mel_len], attention=[rows, T]) -> (durations[T] float tensor,
att_score float)``; ``extract_batch`` does a whole padded batch in one launch, straight from the decoder
loop's attention output on the device.  The shortest-path search runs in libwrnn_mi355.so as a dynamic
programme (one workgroup per utterance); there is no CPU fallback.

Differences from the reference, by design: the caller's ``attention`` is not modified (the reference
shifts its silent rows in place, :50), and where several monotonic paths have exactly the same cost the
choice is down-right > down > right instead of the order scipy's heap happens to scan ties in.
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch

from . import _lib

# utils/text/symbols.py:27 -- '_' and "!'(),.:;? " are the first 11 entries of `phonemes`
SILENT_PHONEME_INDICES = tuple(range(11))


class DurationExtractor:

    def __init__(self, silence_threshold: float, silence_prob_shift: float,
                 silent_phonemes_indices: Sequence[int] = SILENT_PHONEME_INDICES) -> None:
        self.silence_prob_shift = silence_prob_shift
        self.silence_threshold = silence_threshold
        self.silent_phonemes_indices = tuple(int(i) for i in silent_phonemes_indices)

    def extract_batch(self, attention: torch.Tensor, mel: torch.Tensor, tokens: torch.Tensor,
                      mel_len: torch.Tensor, x_len: torch.Tensor, return_cost: bool = False):
        """attention [B,rows,T], mel [B,n_mels,S], tokens [B,T] int, mel_len / x_len [B] ->
        (durations [B,T] float32, att_score [B] float64[, path_cost [B] float64]), all on the device."""
        dev = attention.device
        if dev.type != 'cuda':
            raise RuntimeError("wavernn_amd.DurationExtractor runs only on an MI355X (HIP) device; "
                               "there is no CPU fallback")
        if attention.dtype != torch.float32 or mel.dtype != torch.float32:
            raise TypeError("DurationExtractor expects float32 attention and mel")
        if attention.dim() != 3 or mel.dim() != 3 or tokens.dim() != 2:
            raise ValueError("expected attention [B,rows,T], mel [B,n_mels,S], tokens [B,T]")
        B, rows, Tp = attention.shape
        if mel.shape[0] != B or tokens.shape[0] != B or tokens.shape[1] < 1:
            raise ValueError("batch sizes differ")
        ml = torch.as_tensor(mel_len).to('cpu', torch.int64)
        xl = torch.as_tensor(x_len).to('cpu', torch.int64)
        if ml.numel() != B or xl.numel() != B:
            raise ValueError("mel_len and x_len must hold one value per batch item")
        if B == 0:
            z = torch.zeros(0, Tp, device=dev)
            return (z, torch.zeros(0, dtype=torch.float64, device=dev)) + ((torch.zeros(0, dtype=torch.float64, device=dev),) if return_cost else ())
        max_s, max_t = int(ml.max()), int(xl.max())
        if int(ml.min()) < 1 or max_s > min(rows, mel.shape[2]):
            raise ValueError(f"mel_len must be in [1, min(attention rows, mel frames)] = [1, {min(rows, mel.shape[2])}]")
        if int(xl.min()) < 1 or max_t > min(Tp, tokens.shape[1]):
            raise ValueError(f"x_len must be in [1, {min(Tp, tokens.shape[1])}]")
        att = attention.contiguous()
        melc = mel.to(dev).contiguous()
        tok = tokens.to(device=dev, dtype=torch.int32).contiguous()
        ml_d = ml.to(device=dev, dtype=torch.int32)
        xl_d = xl.to(device=dev, dtype=torch.int32)
        sil = torch.tensor(self.silent_phonemes_indices, dtype=torch.int32, device=dev)
        lib = _lib.load()
        ws = torch.empty(max(int(lib.wrnn_durations_workspace_bytes(B, rows, Tp)), 1), dtype=torch.uint8, device=dev)
        dur = torch.empty(B, Tp, device=dev)
        score = torch.empty(B, dtype=torch.float64, device=dev)
        cost = torch.empty(B, dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            _lib.check(lib.wrnn_durations_extract(
                att.data_ptr(), B, rows, Tp, melc.data_ptr(), melc.shape[1], melc.shape[2], tok.data_ptr(),
                tok.shape[1], ml_d.data_ptr(), xl_d.data_ptr(), max_s, max_t, sil.data_ptr(), sil.numel(),
                float(self.silence_threshold), float(self.silence_prob_shift), ws.data_ptr(), dur.data_ptr(), Tp,
                score.data_ptr(), cost.data_ptr(), stream), "wrnn_durations_extract")
        return (dur, score, cost) if return_cost else (dur, score)

    def __call__(self, x: torch.Tensor, mel: torch.Tensor, attention: torch.Tensor) -> Tuple[torch.Tensor, float]:
        """One utterance, the reference's signature (:23-35).  Tensors may live on the host (they are
        moved to the current HIP device) or on the device; the result is a CPU tensor and a float, as in
        the reference."""
        dev = attention.device if attention.device.type == 'cuda' else torch.device('cuda', torch.cuda.current_device())
        if not torch.cuda.is_available():
            raise RuntimeError("wavernn_amd.DurationExtractor runs only on an MI355X (HIP) device; "
                               "there is no CPU fallback")
        T = x.shape[0]
        mel_len = mel.shape[-1]
        att = attention[:, :T].to(dev, torch.float32)
        dur, score = self.extract_batch(att[None], mel.to(dev, torch.float32)[None], x.to(dev)[None],
                                        torch.tensor([mel_len]), torch.tensor([T]))
        s = float(score[0])
        if s != s:
            raise ZeroDivisionError("division by zero")   # every frame silent: what the reference raises (:82)
        return dur[0, :T].cpu(), s
