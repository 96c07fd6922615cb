"""Host-side mirror of the reference ``ForwardTacotron`` (models/forward_tacotron.py:14-254), the
caller either side of the length regulator and the bidirectional LSTM (SURVEY.md section 8b "forward
models' generate", rows a10 and f1).

Same constructor, state_dict keys, ``forward(batch)`` and ``generate(x, alpha, pitch_function,
energy_function)`` (incl. the ``sum(dur_hat.long()) <= 0 -> 2.0`` fallback, :176-177).  The duration /
pitch / energy predictors, the CBHG pre- and postnet and the projections are per-utterance work on stock
PyTorch-ROCm ops; the expand (``self.lr``) and the frame-recurrent ``self.lstm`` run in libwrnn_mi355.so.
Forward/inference only.
"""
from __future__ import annotations

from typing import Any, Callable, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .bilstm import BiLSTM
from .length_regulator import LengthRegulator
from .bigru import BiGRU
from .tacotron_model import CBHG, BatchNormConv


class SeriesPredictor(nn.Module):  # forward_tacotron.py:14-40

    def __init__(self, num_chars, emb_dim=64, conv_dims=256, rnn_dims=64, dropout=0.5):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, emb_dim)
        self.convs = nn.ModuleList([BatchNormConv(emb_dim, conv_dims, 5, relu=True),
                                    BatchNormConv(conv_dims, conv_dims, 5, relu=True),
                                    BatchNormConv(conv_dims, conv_dims, 5, relu=True)])
        self.rnn = BiGRU(conv_dims, rnn_dims, batch_first=True, bidirectional=True)   # nn.GRU's parameter names
        self.lin = nn.Linear(2 * rnn_dims, 1)
        self.dropout = dropout

    def forward(self, x: torch.Tensor, alpha: float = 1.0) -> torch.Tensor:
        x = self.embedding(x).transpose(1, 2)
        for conv in self.convs:
            x = F.dropout(conv(x), p=self.dropout, training=self.training)
        x, _ = self.rnn(x.transpose(1, 2))
        return self.lin(x) / alpha


class ForwardTacotron(nn.Module):

    def __init__(self, embed_dims: int, series_embed_dims: int, num_chars: int, durpred_conv_dims: int,
                 durpred_rnn_dims: int, durpred_dropout: float, pitch_conv_dims: int, pitch_rnn_dims: int,
                 pitch_dropout: float, pitch_strength: float, energy_conv_dims: int, energy_rnn_dims: int,
                 energy_dropout: float, energy_strength: float, rnn_dims: int, prenet_dims: int, prenet_k: int,
                 postnet_num_highways: int, prenet_dropout: float, postnet_dims: int, postnet_k: int,
                 prenet_num_highways: int, postnet_dropout: float, n_mels: int, padding_value=-11.5129):
        super().__init__()
        self.rnn_dims = rnn_dims
        self.padding_value = padding_value
        self.embedding = nn.Embedding(num_chars, embed_dims)
        self.lr = LengthRegulator()
        self.dur_pred = SeriesPredictor(num_chars, series_embed_dims, durpred_conv_dims, durpred_rnn_dims, durpred_dropout)
        self.pitch_pred = SeriesPredictor(num_chars, series_embed_dims, pitch_conv_dims, pitch_rnn_dims, pitch_dropout)
        self.energy_pred = SeriesPredictor(num_chars, series_embed_dims, energy_conv_dims, energy_rnn_dims, energy_dropout)
        self.prenet = CBHG(K=prenet_k, in_channels=embed_dims, channels=prenet_dims,
                           proj_channels=[prenet_dims, embed_dims], num_highways=prenet_num_highways,
                           dropout=prenet_dropout)
        self.lstm = BiLSTM(2 * prenet_dims, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.postnet = CBHG(K=postnet_k, in_channels=n_mels, channels=postnet_dims,
                            proj_channels=[postnet_dims, n_mels], num_highways=postnet_num_highways,
                            dropout=postnet_dropout)
        self.post_proj = nn.Linear(2 * postnet_dims, n_mels, bias=False)
        self.pitch_strength = pitch_strength
        self.energy_strength = energy_strength
        self.pitch_proj = nn.Conv1d(1, 2 * prenet_dims, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, 2 * prenet_dims, kernel_size=3, padding=1)

    def __repr__(self):
        return f'ForwardTacotron, num params: {sum(p.numel() for p in self.parameters())}'

    def _condition(self, x, pitch, energy):
        """:133-143 / :210-220: embedding -> CBHG prenet, plus projected pitch and energy."""
        x = self.prenet(self.embedding(x).transpose(1, 2))
        x = x + self.pitch_proj(pitch).transpose(1, 2) * self.pitch_strength
        return x + self.energy_proj(energy).transpose(1, 2) * self.energy_strength

    def _post(self, x):
        x = self.lin(x).transpose(1, 2)
        x_post = self.post_proj(self.postnet(x)).transpose(1, 2)
        return x, x_post

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
        pitch, energy = batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1)
        if self.training:
            self.step += 1
        dur_hat = self.dur_pred(x).squeeze(-1)
        pitch_hat = self.pitch_pred(x).transpose(1, 2)
        energy_hat = self.energy_pred(x).transpose(1, 2)
        x = self.lr(self._condition(x, pitch, energy), dur)
        x = pack_padded_sequence(x, lengths=mel_lens.cpu(), enforce_sorted=False, batch_first=True)
        x, _ = self.lstm(x)
        x, _ = pad_packed_sequence(x, padding_value=self.padding_value, batch_first=True)
        x, x_post = self._post(x)
        return {'mel': self._pad(x, mel.size(2)), 'mel_post': self._pad(x_post, mel.size(2)),
                'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    def generate(self, x: torch.Tensor, alpha=1.0,
                 pitch_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x,
                 energy_function: Callable[[torch.Tensor], torch.Tensor] = lambda x: x) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            dur_hat = self.dur_pred(x, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x).transpose(1, 2))
            return self._generate_mel(x=x, dur_hat=dur_hat, pitch_hat=pitch_hat, energy_hat=energy_hat)

    def generate_jit(self, x: torch.Tensor, alpha: float = 1.0, beta: float = 1.0) -> Dict[str, torch.Tensor]:
        with torch.no_grad():
            dur_hat = self.dur_pred(x, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = self.pitch_pred(x).transpose(1, 2) * beta
            energy_hat = self.energy_pred(x).transpose(1, 2)
            return self._generate_mel(x=x, dur_hat=dur_hat, pitch_hat=pitch_hat, energy_hat=energy_hat)

    def get_step(self) -> int:
        return self.step.data.item()

    def _generate_mel(self, x, dur_hat, pitch_hat, energy_hat) -> Dict[str, torch.Tensor]:
        x = self.lr(self._condition(x, pitch_hat, energy_hat), dur_hat)
        x, _ = self.lstm(x)
        x, x_post = self._post(x)
        return {'mel': x, 'mel_post': x_post, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    def _pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        x = x[:, :, :max_len]
        return F.pad(x, [0, max_len - x.size(2), 0, 0], 'constant', self.padding_value)

    @classmethod
    def from_config(cls, config: Dict[str, Any], num_chars: int) -> 'ForwardTacotron':
        """``num_chars`` = len(phonemes) of the reference's symbol table (the reference imports it)."""
        model_config = dict(config['forward_tacotron']['model'])
        model_config['num_chars'] = num_chars
        model_config['n_mels'] = config['dsp']['num_mels']
        return cls(**model_config)
