"""Host-side mirrors of the reference's other forward models, the remaining callers of the length regulator
listed in SURVEY.md section 8b ("forward models' generate"):

* ``FastPitch``             models/fast_pitch.py:44-235       (``self.lr`` at :150 and :213)
* ``MultiForwardTacotron``  models/multi_forward_tacotron.py:96-324 (``self.lr`` :220,292, ``self.lstm`` :225,294)
* ``MultiFastPitch``        models/multi_fast_pitch.py:93-327  (``self.lr`` :228,305)

Same constructors, state_dict keys (so reference checkpoints load), ``forward(batch)`` dicts and ``generate``
signatures.  The transformer blocks, predictors and projections are per-utterance work on stock PyTorch-ROCm
ops, written here from the reference's behaviour (models/common_layers.py:127-240); the expand runs in
libwrnn_mi355.so (``LengthRegulator``), and so do the frame-recurrent ``lstm`` of the multispeaker
ForwardTacotron (``BiLSTM``) and every CBHG / predictor GRU (``BiGRU``).  Forward/inference only.

Reference quirks kept on purpose: the multispeaker ForwardTacotron builds its predictors with their default
``speaker_emb_dims=256`` whatever the model's own value is (multi_forward_tacotron.py:135-157); both multispeaker
``generate`` paths take ``argmax`` over ``pitch_cond_hat.squeeze()`` and therefore serve one utterance per call.
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

from .bigru import BiGRU
from .bilstm import BiLSTM
from .length_regulator import LengthRegulator
from .tacotron_model import CBHG, BatchNormConv

Identity = lambda v: v  # noqa: E731


# ---------------------------------------------------------------------------- transformer pieces
class PositionalEncoding(nn.Module):  # common_layers.py:127-145
    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000) -> None:
        super().__init__()
        self.dropout = nn.Dropout(p=dropout)
        self.scale = nn.Parameter(torch.ones(1))
        pos = torch.arange(max_len, dtype=torch.float)[:, None]
        freq = torch.exp(torch.arange(0, d_model, 2).float() * (-math.log(10000.0) / d_model))
        table = torch.zeros(max_len, d_model)
        table[:, 0::2], table[:, 1::2] = torch.sin(pos * freq), torch.cos(pos * freq)
        self.register_buffer('pe', table[:, None, :])          # [max_len, 1, d_model]

    def forward(self, x):                                        # [T, N, d_model]
        return self.dropout(x + self.scale * self.pe[:x.size(0)])


class FFTBlock(nn.Module):  # common_layers.py:148-185: post-norm self-attention, then a two-conv feed-forward over time
    def __init__(self, d_model: int, nhead: int, conv1_kernel: int, conv2_kernel: int, d_fft: int, dropout: float = 0.1):
        super().__init__()
        self.self_attn = nn.MultiheadAttention(d_model, nhead, dropout=dropout)
        self.dropout = nn.Dropout(dropout)
        self.conv1 = nn.Conv1d(d_model, d_fft, conv1_kernel, stride=1, padding=conv1_kernel // 2)
        self.conv2 = nn.Conv1d(d_fft, d_model, conv2_kernel, stride=1, padding=conv2_kernel // 2)
        self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2 = nn.Dropout(dropout), nn.Dropout(dropout)
        self.activation = nn.ReLU()

    def forward(self, src, src_pad_mask=None):                   # [T, N, d_model]
        att = self.self_attn(src, src, src, attn_mask=None, key_padding_mask=src_pad_mask)[0]
        src = self.norm1(src + self.dropout1(att))
        y = src.permute(1, 2, 0)                                 # [N, d_model, T] for the convs
        y = self.conv2(self.activation(self.conv1(y))).permute(2, 0, 1)
        return self.norm2(src + self.dropout2(y))


class ForwardTransformer(nn.Module):  # common_layers.py:188-223
    def __init__(self, d_model: int, d_fft: int, layers: int, heads: int, conv1_kernel: int, conv2_kernel: int,
                 dropout: float = 0.1) -> None:
        super().__init__()
        self.d_model = d_model
        self.pos_encoder = PositionalEncoding(d_model, dropout)
        self.layers = nn.ModuleList(FFTBlock(d_model=d_model, nhead=heads, d_fft=d_fft, conv1_kernel=conv1_kernel,
                                             conv2_kernel=conv2_kernel, dropout=dropout) for _ in range(layers))
        self.norm = nn.LayerNorm(d_model)

    def forward(self, x, src_pad_mask=None):                     # [N, T, d_model]
        x = self.pos_encoder(x.transpose(0, 1))
        for layer in self.layers:
            x = layer(x, src_pad_mask=src_pad_mask)
        return self.norm(x).transpose(0, 1)


def token_pad_mask(x: torch.Tensor) -> torch.Tensor:
    """make_token_len_mask(x.T) of the reference (common_layers.py:232-233): True where the token id is 0."""
    return x == 0


def frame_pad_mask(n_rows: int, n_frames: int, mel_lens, device) -> torch.Tensor:
    """fast_pitch.py:152-154: True past each row's mel length."""
    lens = torch.as_tensor(mel_lens).to(device).long()
    return torch.arange(n_frames, device=device)[None, :] >= lens[:, None]


def tile_speaker(x: torch.Tensor, semb: torch.Tensor) -> torch.Tensor:
    """[B,T,C] ++ the utterance's speaker embedding at every position."""
    return torch.cat([x, semb[:, None, :].expand(-1, x.shape[1], -1)], dim=2)


def pad_frames(x: torch.Tensor, max_len: int, value: float) -> torch.Tensor:
    x = x[:, :, :max_len]
    return F.pad(x, [0, max_len - x.size(2), 0, 0], 'constant', value)


class TransformerPredictor(nn.Module):
    """fast_pitch.py:14-41 (no speaker, no condition), multi_fast_pitch.py:14-49 (speaker) and :52-90 (speaker +
    categorical condition): embedding [++ condition embedding ++ speaker embedding] -> transformer -> linear."""

    def __init__(self, num_chars, d_model, n_heads, d_fft, layers, conv1_kernel, conv2_kernel, dropout=0.1,
                 speaker_emb_dims=0, cond_emb_size=0, cond_emb_dims=0, out_dim=1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, d_model)
        if cond_emb_size:
            self.conditional_embedding = nn.Embedding(cond_emb_size, cond_emb_dims)
        width = d_model + speaker_emb_dims + (cond_emb_dims if cond_emb_size else 0)
        self.transformer = ForwardTransformer(heads=n_heads, dropout=dropout, d_model=width, d_fft=d_fft,
                                              conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel, layers=layers)
        self.lin = nn.Linear(width, out_dim)
        self._cond, self._spk = bool(cond_emb_size), speaker_emb_dims > 0

    def forward(self, x, speaker_emb=None, x_cond=None, src_pad_mask=None, alpha: float = 1.0):
        x = self.embedding(x)
        if self._cond:
            x = torch.cat([x, self.conditional_embedding(x_cond)], dim=2)
        if self._spk:
            x = tile_speaker(x, speaker_emb)
        return self.lin(self.transformer(x, src_pad_mask=src_pad_mask)) / alpha


class ConvPredictor(nn.Module):
    """multi_forward_tacotron.py:14-51 (speaker) and :54-93 (speaker + categorical condition): embedding
    [++ condition embedding] ++ speaker embedding -> three conv blocks -> bidirectional GRU -> linear."""

    def __init__(self, num_chars, emb_dim=64, conv_dims=256, rnn_dims=64, dropout=0.5, speaker_emb_dims=256,
                 cond_emb_size=0, cond_emb_dims=8, out_dim=1):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, emb_dim)
        if cond_emb_size:
            self.pitch_cond_embedding = nn.Embedding(cond_emb_size, cond_emb_dims)
        width = emb_dim + speaker_emb_dims + (cond_emb_dims if cond_emb_size else 0)
        self.convs = nn.ModuleList([BatchNormConv(width, conv_dims, 5, relu=True),
                                    BatchNormConv(conv_dims, conv_dims, 5, relu=True),
                                    BatchNormConv(conv_dims, conv_dims, 5, relu=True)])
        self.rnn = BiGRU(conv_dims, rnn_dims, batch_first=True, bidirectional=True)
        self.lin = nn.Linear(2 * rnn_dims, out_dim)
        self.dropout = dropout
        self._cond = bool(cond_emb_size)

    def forward(self, x, semb, x_cond=None, alpha: float = 1.0):
        x = self.embedding(x)
        if self._cond:
            x = torch.cat([x, self.pitch_cond_embedding(x_cond)], dim=2)
        x = tile_speaker(x, semb).transpose(1, 2)
        for conv in self.convs:
            x = F.dropout(conv(x), p=self.dropout, training=self.training)
        x, _ = self.rnn(x.transpose(1, 2))
        return self.lin(x) / alpha


# ---------------------------------------------------------------------------- FastPitch
class FastPitch(nn.Module):

    def __init__(self, num_chars: int, durpred_dropout: float, durpred_d_model: int, durpred_n_heads: int,
                 durpred_layers: int, durpred_d_fft: int, pitch_dropout: float, pitch_d_model: int, pitch_n_heads: int,
                 pitch_layers: int, pitch_d_fft: int, energy_dropout: float, energy_d_model: int, energy_n_heads: int,
                 energy_layers: int, energy_d_fft: int, pitch_strength: float, energy_strength: float, d_model: int,
                 conv1_kernel: int, conv2_kernel: int, prenet_layers: int, prenet_heads: int, prenet_fft: int,
                 prenet_dropout: float, postnet_layers: int, postnet_heads: int, postnet_fft: int,
                 postnet_dropout: float, n_mels: int, padding_value=-11.5129):
        super().__init__()
        self.padding_value = padding_value
        self.lr = LengthRegulator()
        kern = dict(conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel)
        self.dur_pred = TransformerPredictor(num_chars, durpred_d_model, durpred_n_heads, durpred_d_fft, durpred_layers,
                                             dropout=durpred_dropout, **kern)
        self.pitch_pred = TransformerPredictor(num_chars, pitch_d_model, pitch_n_heads, pitch_d_fft, pitch_layers,
                                               dropout=pitch_dropout, **kern)
        self.energy_pred = TransformerPredictor(num_chars, energy_d_model, energy_n_heads, energy_d_fft, energy_layers,
                                                dropout=energy_dropout, **kern)
        self.embedding = nn.Embedding(num_embeddings=num_chars, embedding_dim=d_model)
        self.prenet = ForwardTransformer(heads=prenet_heads, dropout=prenet_dropout, d_model=d_model, d_fft=prenet_fft,
                                         layers=prenet_layers, **kern)
        self.postnet = ForwardTransformer(heads=postnet_heads, dropout=postnet_dropout, d_model=d_model, d_fft=postnet_fft,
                                          layers=postnet_layers, **kern)
        self.lin = nn.Linear(d_model, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.pitch_strength, self.energy_strength = pitch_strength, energy_strength
        self.pitch_proj = nn.Conv1d(1, d_model, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, d_model, kernel_size=3, padding=1)

    def __repr__(self):
        return f'FastPitch, num params: {sum(p.numel() for p in self.parameters())}'

    def _encode(self, x, pitch, energy, speaker_emb=None):
        """fast_pitch.py:139-148 / :200-211: embedding -> transformer prenet (token padding masked), plus the
        projected pitch and energy curves."""
        mask = token_pad_mask(x)
        x = self.embedding(x)
        if speaker_emb is not None:
            x = tile_speaker(x, speaker_emb)
        x = self.prenet(x, src_pad_mask=mask)
        x = x + self.pitch_proj(pitch).transpose(1, 2) * self.pitch_strength
        return x + self.energy_proj(energy).transpose(1, 2) * self.energy_strength

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
        if self.training:
            self.step += 1
        mask = token_pad_mask(x)
        dur_hat = self.dur_pred(x, src_pad_mask=mask).squeeze(-1)
        pitch_hat = self.pitch_pred(x, src_pad_mask=mask).transpose(1, 2)
        energy_hat = self.energy_pred(x, src_pad_mask=mask).transpose(1, 2)
        y = self.lr(self._encode(x, batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1)), dur)
        y = self.postnet(y, src_pad_mask=frame_pad_mask(y.size(0), y.size(1), mel_lens, y.device))
        y = self.pad(self.lin(y).transpose(1, 2), mel.size(2))
        return {'mel': y, 'mel_post': y, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    def generate(self, x: torch.Tensor, alpha=1.0, pitch_function: Callable = Identity,
                 energy_function: Callable = Identity) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            dur_hat = self.dur_pred(x, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x).transpose(1, 2))
            return self._generate_mel(x=x, dur_hat=dur_hat, pitch_hat=pitch_hat, energy_hat=energy_hat)

    def pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        return pad_frames(x, max_len, self.padding_value)

    def get_step(self) -> int:
        return self.step.data.item()

    def _generate_mel(self, x, dur_hat, pitch_hat, energy_hat) -> Dict[str, torch.Tensor]:
        y = self.lr(self._encode(x, pitch_hat, energy_hat), dur_hat)
        y = self.lin(self.postnet(y, src_pad_mask=None)).transpose(1, 2)
        return {'mel': y, 'mel_post': y, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat}

    @classmethod
    def from_config(cls, config: Dict[str, Any], num_chars: int) -> 'FastPitch':
        """``num_chars`` = len(phonemes) of the reference's symbol table (the reference imports it)."""
        return cls(**dict(config['fast_pitch']['model'], num_chars=num_chars, n_mels=config['dsp']['num_mels']))


class MultiFastPitch(FastPitch):

    def __init__(self, num_chars: int, durpred_dropout: float, durpred_d_model: int, durpred_n_heads: int,
                 durpred_layers: int, durpred_d_fft: int, pitch_dropout: float, pitch_d_model: int, pitch_n_heads: int,
                 pitch_layers: int, pitch_d_fft: int, energy_dropout: float, energy_d_model: int, energy_n_heads: int,
                 energy_layers: int, energy_d_fft: int, pitch_cond_d_model: int, pitch_cond_n_heads: int,
                 pitch_cond_layers: int, pitch_cond_d_fft: int, pitch_cond_dropout: float, pitch_cond_output_dims: int,
                 pitch_strength: float, energy_strength: float, d_model: int, conv1_kernel: int, conv2_kernel: int,
                 prenet_layers: int, prenet_heads: int, prenet_fft: int, prenet_dropout: float, postnet_layers: int,
                 postnet_heads: int, postnet_fft: int, postnet_dropout: float, n_mels: int, speaker_emb_dims: int,
                 padding_value=-11.5129):
        nn.Module.__init__(self)
        self.padding_value = padding_value
        self.lr = LengthRegulator()
        kern = dict(conv1_kernel=conv1_kernel, conv2_kernel=conv2_kernel)
        spk = dict(speaker_emb_dims=speaker_emb_dims, **kern)
        self.dur_pred = TransformerPredictor(num_chars, durpred_d_model, durpred_n_heads, durpred_d_fft, durpred_layers,
                                             dropout=durpred_dropout, cond_emb_size=4, cond_emb_dims=8, **spk)
        self.pitch_pred = TransformerPredictor(num_chars, pitch_d_model, pitch_n_heads, pitch_d_fft, pitch_layers,
                                               dropout=pitch_dropout, cond_emb_size=4, cond_emb_dims=8, **spk)
        self.pitch_cond_pred = TransformerPredictor(num_chars, pitch_cond_d_model, pitch_cond_n_heads, pitch_cond_d_fft,
                                                    pitch_cond_layers, dropout=pitch_cond_dropout,
                                                    out_dim=pitch_cond_output_dims, **spk)
        self.energy_pred = TransformerPredictor(num_chars, energy_d_model, energy_n_heads, energy_d_fft, energy_layers,
                                                dropout=energy_dropout, **spk)
        width = d_model + speaker_emb_dims
        self.embedding = nn.Embedding(num_embeddings=num_chars, embedding_dim=d_model)
        self.prenet = ForwardTransformer(heads=prenet_heads, dropout=prenet_dropout, d_model=width, d_fft=prenet_fft,
                                         layers=prenet_layers, **kern)
        self.postnet = ForwardTransformer(heads=postnet_heads, dropout=postnet_dropout, d_model=width, d_fft=postnet_fft,
                                          layers=postnet_layers, **kern)
        self.lin = nn.Linear(width, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.pitch_strength, self.energy_strength = pitch_strength, energy_strength
        self.pitch_proj = nn.Conv1d(1, width, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, width, kernel_size=3, padding=1)

    def __repr__(self):
        return f'MultiFastPitch, num params: {sum(p.numel() for p in self.parameters())}'

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
        semb, cond = batch['speaker_emb'], batch['pitch_cond']
        if self.training:
            self.step += 1
        mask = token_pad_mask(x)
        dur_hat = self.dur_pred(x, speaker_emb=semb, x_cond=cond, src_pad_mask=mask).squeeze(-1)
        pitch_hat = self.pitch_pred(x, speaker_emb=semb, x_cond=cond, src_pad_mask=mask).transpose(1, 2)
        pitch_cond_hat = self.pitch_cond_pred(x, speaker_emb=semb, src_pad_mask=mask)
        energy_hat = self.energy_pred(x, speaker_emb=semb, src_pad_mask=mask).transpose(1, 2)
        y = self.lr(self._encode(x, batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1), semb), dur)
        y = self.postnet(y, src_pad_mask=frame_pad_mask(y.size(0), y.size(1), mel_lens, y.device))
        y = self.pad(self.lin(y).transpose(1, 2), mel.size(2))
        return {'mel': y, 'mel_post': y, 'pitch_cond': pitch_cond_hat, 'dur': dur_hat, 'pitch': pitch_hat,
                'energy': energy_hat}

    def generate(self, x: torch.Tensor, speaker_emb: torch.Tensor, alpha=1.0, pitch_function: Callable = Identity,
                 energy_function: Callable = Identity) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            cond = self.pitch_cond_pred(x, speaker_emb=speaker_emb, alpha=alpha).squeeze(-1)
            cond = torch.argmax(cond.squeeze(), dim=1).long().unsqueeze(0)
            dur_hat = self.dur_pred(x, x_cond=cond, speaker_emb=speaker_emb, alpha=alpha).squeeze(2)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x, x_cond=cond, speaker_emb=speaker_emb).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x, speaker_emb=speaker_emb).transpose(1, 2))
            return self._generate_mel(x=x, dur_hat=dur_hat, pitch_hat=pitch_hat, energy_hat=energy_hat,
                                      pitch_cond_hat=cond, speaker_emb=speaker_emb)

    def _generate_mel(self, x, speaker_emb, dur_hat, pitch_hat, pitch_cond_hat, energy_hat) -> Dict[str, torch.Tensor]:
        y = self.lr(self._encode(x, pitch_hat, energy_hat, speaker_emb), dur_hat)
        y = self.lin(self.postnet(y, src_pad_mask=None)).transpose(1, 2)
        return {'mel': y, 'mel_post': y, 'dur': dur_hat, 'pitch_cond': pitch_cond_hat, 'pitch': pitch_hat,
                'energy': energy_hat}

    @classmethod
    def from_config(cls, config: Dict[str, Any], num_chars: int) -> 'MultiFastPitch':
        return cls(**dict(config['multi_fast_pitch']['model'], num_chars=num_chars, n_mels=config['dsp']['num_mels']))


# ---------------------------------------------------------------------------- multispeaker ForwardTacotron
class MultiForwardTacotron(nn.Module):

    def __init__(self, embed_dims: int, series_embed_dims: int, num_chars: int, durpred_conv_dims: int,
                 durpred_rnn_dims: int, durpred_dropout: float, pitch_conv_dims: int, pitch_rnn_dims: int,
                 pitch_dropout: float, pitch_strength: float, pitch_cond_conv_dims: int, pitch_cond_rnn_dims: int,
                 pitch_cond_dropout: float, energy_conv_dims: int, energy_rnn_dims: int, energy_dropout: float,
                 energy_strength: float, rnn_dims: int, prenet_dims: int, prenet_k: int, postnet_num_highways: int,
                 prenet_dropout: float, postnet_dims: int, postnet_k: int, prenet_num_highways: int,
                 postnet_dropout: float, n_mels: int, speaker_emb_dims: int, pitch_cond_emb_dims: int,
                 pitch_cond_categorical_dims: int, padding_value=-11.5129):
        super().__init__()
        self.rnn_dims = rnn_dims
        self.padding_value = padding_value
        self.embedding = nn.Embedding(num_chars, embed_dims)
        self.lr = LengthRegulator()
        # (the predictors keep their default speaker_emb_dims=256, as in the reference)
        self.dur_pred = ConvPredictor(num_chars, series_embed_dims, durpred_conv_dims, durpred_rnn_dims, durpred_dropout,
                                      cond_emb_size=4, cond_emb_dims=pitch_cond_emb_dims)
        self.pitch_cond_pred = ConvPredictor(num_chars, series_embed_dims, pitch_cond_conv_dims, pitch_cond_rnn_dims,
                                             pitch_cond_dropout, out_dim=pitch_cond_categorical_dims)
        self.pitch_pred = ConvPredictor(num_chars, series_embed_dims, pitch_conv_dims, pitch_rnn_dims, pitch_dropout,
                                        cond_emb_size=4, cond_emb_dims=pitch_cond_emb_dims)
        self.energy_pred = ConvPredictor(num_chars, series_embed_dims, energy_conv_dims, energy_rnn_dims, energy_dropout)
        self.prenet = CBHG(K=prenet_k, in_channels=embed_dims, channels=prenet_dims,
                           proj_channels=[prenet_dims, embed_dims], num_highways=prenet_num_highways,
                           dropout=prenet_dropout)
        width = 2 * prenet_dims + speaker_emb_dims
        self.lstm = BiLSTM(width, rnn_dims)
        self.lin = nn.Linear(2 * rnn_dims, n_mels)
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.postnet = CBHG(K=postnet_k, in_channels=n_mels, channels=postnet_dims,
                            proj_channels=[postnet_dims, n_mels], num_highways=postnet_num_highways,
                            dropout=postnet_dropout)
        self.post_proj = nn.Linear(2 * postnet_dims, n_mels, bias=False)
        self.pitch_strength, self.energy_strength = pitch_strength, energy_strength
        self.pitch_proj = nn.Conv1d(1, width, kernel_size=3, padding=1)
        self.energy_proj = nn.Conv1d(1, width, kernel_size=3, padding=1)

    def __repr__(self):
        return f'MultiForwardTacotron, num params: {sum(p.numel() for p in self.parameters())}'

    def _condition(self, x, semb, pitch, energy):
        """multi_forward_tacotron.py:205-218 / :277-290: embedding -> CBHG prenet ++ speaker embedding, plus the
        projected pitch and energy curves."""
        x = tile_speaker(self.prenet(self.embedding(x).transpose(1, 2)), semb)
        x = x + self.pitch_proj(pitch).transpose(1, 2) * self.pitch_strength
        return x + self.energy_proj(energy).transpose(1, 2) * self.energy_strength

    def _post(self, x):
        x = self.lin(x).transpose(1, 2)
        return x, self.post_proj(self.postnet(x)).transpose(1, 2)

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x, mel, dur, mel_lens = batch['x'], batch['mel'], batch['dur'], batch['mel_len']
        semb, cond = batch['speaker_emb'], batch['pitch_cond']
        if self.training:
            self.step += 1
        pitch_cond_hat = self.pitch_cond_pred(x, semb).squeeze(-1)
        dur_hat = self.dur_pred(x, semb, cond).squeeze(-1)
        pitch_hat = self.pitch_pred(x, semb, cond).transpose(1, 2)
        energy_hat = self.energy_pred(x, semb).transpose(1, 2)
        y = self.lr(self._condition(x, semb, batch['pitch'].unsqueeze(1), batch['energy'].unsqueeze(1)), dur)
        y = pack_padded_sequence(y, lengths=mel_lens.cpu(), enforce_sorted=False, batch_first=True)
        y, _ = self.lstm(y)
        y, _ = pad_packed_sequence(y, padding_value=self.padding_value, batch_first=True)
        y, y_post = self._post(y)
        return {'mel': self._pad(y, mel.size(2)), 'mel_post': self._pad(y_post, mel.size(2)), 'dur': dur_hat,
                'pitch': pitch_hat, 'energy': energy_hat, 'pitch_cond': pitch_cond_hat}

    def generate(self, x: torch.Tensor, speaker_emb: torch.Tensor, alpha=1.0, pitch_function: Callable = Identity,
                 energy_function: Callable = Identity) -> Dict[str, torch.Tensor]:
        self.eval()
        with torch.no_grad():
            cond = self.pitch_cond_pred(x, speaker_emb).squeeze(-1)
            cond = torch.argmax(cond.squeeze(), dim=1).long().unsqueeze(0)
            dur_hat = self.dur_pred(x, speaker_emb, cond, alpha=alpha).squeeze(-1)
            if torch.sum(dur_hat.long()) <= 0:
                torch.fill_(dur_hat, value=2.)
            pitch_hat = pitch_function(self.pitch_pred(x, speaker_emb, cond).transpose(1, 2))
            energy_hat = energy_function(self.energy_pred(x, speaker_emb).transpose(1, 2))
            return self._generate_mel(x=x, dur_hat=dur_hat, pitch_hat=pitch_hat, energy_hat=energy_hat,
                                      pitch_cond_hat=cond, semb=speaker_emb)

    def get_step(self) -> int:
        return self.step.data.item()

    def _generate_mel(self, x, semb, dur_hat, pitch_hat, pitch_cond_hat, energy_hat) -> Dict[str, torch.Tensor]:
        y = self.lr(self._condition(x, semb, pitch_hat, energy_hat), dur_hat)
        y, _ = self.lstm(y)
        y, y_post = self._post(y)
        return {'mel': y, 'mel_post': y_post, 'dur': dur_hat, 'pitch': pitch_hat, 'energy': energy_hat,
                'pitch_cond': pitch_cond_hat.unsqueeze(1)}

    def _pad(self, x: torch.Tensor, max_len: int) -> torch.Tensor:
        return pad_frames(x, max_len, self.padding_value)

    @classmethod
    def from_config(cls, config: Dict[str, Any], num_chars: int) -> 'MultiForwardTacotron':
        return cls(**dict(config['multi_forward_tacotron']['model'], num_chars=num_chars,
                          n_mels=config['dsp']['num_mels']))
