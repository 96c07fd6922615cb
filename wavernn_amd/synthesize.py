"""Text -> mel -> audio for many utterances: the per-sentence loop of the reference's inference script
(gen_forward.py:88-107: tokens -> ``tts_model.generate`` -> vocoder) with the build-defined WaveRNN vocoder
(absent from the mounted reference) in the place of griffinlim / melgan, for BASELINE
configs[3]: utterances are independent, so a node shards them by utterance (``sharding.my_slice``), one process
per GPU, no collective on the data path, and rank 0 gathers the audio on the host.

Every utterance goes through ``tts_model.generate`` on its own, as in the reference; the folds of many utterances
are then the rows of a few WaveRNN generation loops (``WaveRNN.generate_batch``: rows are independent, so each
utterance's audio is bit for bit what ``WaveRNN.generate`` gives it alone under the same uniforms).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import sharding


def utterance_noise(vocoder, n_frames: int, index: int, seed: int, target: int, overlap: int) -> torch.Tensor:
    """Sampler uniforms [folds, target + 2 * overlap] of utterance `index`: a function of (seed, index) only, so an
    utterance sounds the same whatever shard, batch or rank it lands in."""
    total = n_frames * vocoder.hop_length
    folds = (total - overlap) // (target + overlap)
    if total - (folds * (overlap + target) + overlap) != 0:
        folds += 1
    g = torch.Generator().manual_seed(int(seed) * 1_000_003 + int(index))
    return torch.rand(folds, target + 2 * overlap, generator=g)


@torch.no_grad()
def synthesize(tts_model, vocoder, tokens: Sequence[torch.Tensor], *, first_index: int = 0, alpha: float = 1.0,
               target: int = 11000, overlap: int = 550, seed: int = 0, max_rows: int = 128) -> List[Dict]:
    """tokens: one int64 tensor [T_i] per utterance.  Returns per utterance {'mel_post': [n_mels, M_i] (CPU),
    'wav': float64 numpy [(M_i - 1) * hop]}."""
    dev = next(tts_model.parameters()).device
    mels, noise = [], []
    for i, x in enumerate(tokens):
        gen = tts_model.generate(x=x.to(dev)[None], alpha=alpha)           # gen_forward.py:107
        m = gen['mel_post']                                                 # [1, n_mels, M]
        mels.append(m)
        noise.append(utterance_noise(vocoder, m.shape[2], first_index + i, seed, target, overlap))
    wavs = vocoder.generate_batch(mels, target=target, overlap=overlap, noise=noise, max_rows=max_rows) if mels else []
    return [{'mel_post': m[0].cpu(), 'wav': w} for m, w in zip(mels, wavs)]


def synthesize_sharded(tts_model, vocoder, tokens: Sequence[torch.Tensor], rank: int, world_size: int,
                       group=None, **kw) -> Optional[List[np.ndarray]]:
    """This rank's slice of the utterances (BASELINE configs[3]: 1k utterances over 8 GPUs); rank 0 returns every
    utterance's audio in order, the other ranks None.  The gather runs on `group`: pass `sharding.host_group()` (gloo)
    so that the audio travels host to host and no device collective is involved."""
    lo, hi = sharding.my_slice(len(tokens), rank, world_size)
    mine = synthesize(tts_model, vocoder, tokens[lo:hi], first_index=lo, **kw)
    return sharding.gather_list_to_rank0([u['wav'] for u in mine], group=group)
