"""The data formats either side of the attention -> duration step:
what DurationExtractionPipeline writes to and reads from disk
(duration_extraction/duration_extraction_pipe.py:114-125 ``att_pred/<id>.npy``, :183 ``alg/<id>.npy``).

``save_attention_batch`` is the body of the reference's per-batch loop in ``extract_attentions``;
``extract_durations_batch`` does for a whole device-resident batch what ``extract_durations`` does per
file, and writes the same ``alg/<id>.npy`` integer arrays.  No dataset / dataloader code lives here.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Sequence, Tuple, Union

import numpy as np
import torch


def save_attention_batch(attention: torch.Tensor, x_len: torch.Tensor, mel_len: torch.Tensor,
                         item_ids: Sequence[str], att_pred_dir: Union[str, Path]) -> None:
    """attention [B,S,T] -> att_pred/<id>.npy, float32 [mel_len, x_len] each (pipe:118-125)."""
    out = Path(att_pred_dir)
    att = attention.detach().to('cpu', torch.float32)
    for b, item_id in enumerate(item_ids):
        np.save(out / f'{item_id}.npy', att[b, :int(mel_len[b]), :int(x_len[b])].numpy(), allow_pickle=False)


def extract_durations_batch(extractor, attention: torch.Tensor, mel: torch.Tensor, tokens: torch.Tensor,
                            mel_len: torch.Tensor, x_len: torch.Tensor, item_ids: Sequence[str],
                            alg_dir: Union[str, Path, None] = None) -> Dict[str, Tuple[np.ndarray, float]]:
    """One launch for the batch; returns {id: (durations int array [x_len], att_score)} and, when
    ``alg_dir`` is given, writes alg/<id>.npy exactly as pipe:183 does (``durations.astype(int)``)."""
    dur, score = extractor.extract_batch(attention, mel, tokens, mel_len, x_len)
    dur, score = dur.cpu().numpy(), score.cpu().numpy()
    res = {}
    for b, item_id in enumerate(item_ids):
        d = dur[b, :int(x_len[b])].astype(int)
        if d.sum() != int(mel_len[b]):   # the reference prints the same warning (pipe:67-68)
            print(f'WARNINNG: Sum of durations did not match mel length for item {item_id}!')
        res[item_id] = (d, float(score[b]))
        if alg_dir is not None:
            np.save(Path(alg_dir) / f'{item_id}.npy', d, allow_pickle=False)
    return res
