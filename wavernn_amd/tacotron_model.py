"""Host-side mirror of the reference ``Tacotron`` module (models/tacotron.py:238-385), SURVEY.md
section 8 row f2: the modules either side of the decoder loop (aligner, encoder + CBHG, encoder
projections, postnet CBHG) are per-utterance work on stock PyTorch-ROCm ops -- except the CBHG blocks'
bidirectional GRU, a second per-frame recurrence, which runs in the library (``BiGRU``) -- written here from the
reference's behaviour with the reference's parameter names, so that a reference checkpoint loads
with ``load_state_dict`` and ``Tacotron.forward(batch)`` returns the same dict.  The decoder loop
(tacotron.py:339-345) is ONE call into libwrnn_mi355.so (``Decoder.decode_teacher_forced``).

Layer semantics follow models/common_layers.py:27-124 (HighwayNetwork, BatchNormConv, CBHG) and
models/tacotron.py:13-100 (Aligner, Encoder, PreNet).
"""
from __future__ import annotations

from typing import Any, Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .bigru import BiGRU
from .tacotron import Decoder


class HighwayNetwork(nn.Module):  # common_layers.py:27-40
    def __init__(self, size: int) -> None:
        super().__init__()
        self.W1 = nn.Linear(size, size)
        self.W2 = nn.Linear(size, size)
        with torch.no_grad():
            self.W1.bias.zero_()

    def forward(self, x):
        gate = torch.sigmoid(self.W2(x))
        return gate * F.relu(self.W1(x)) + (1.0 - gate) * x


class BatchNormConv(nn.Module):  # common_layers.py:43-57: conv -> (relu) -> batch norm
    def __init__(self, in_channels: int, out_channels: int, kernel: int, relu: bool = True) -> None:
        super().__init__()
        self.conv = nn.Conv1d(in_channels, out_channels, kernel, stride=1, padding=kernel // 2, bias=False)
        self.bnorm = nn.BatchNorm1d(out_channels)
        self.relu = relu

    def forward(self, x):
        x = self.conv(x)
        if self.relu:
            x = F.relu(x)
        return self.bnorm(x)


class CBHG(nn.Module):  # common_layers.py:60-124
    def __init__(self, K: int, in_channels: int, channels: int, proj_channels: list, num_highways: int,
                 dropout: float = 0.5) -> None:
        super().__init__()
        self.dropout = dropout
        self.bank_kernels = list(range(1, K + 1))
        self.conv1d_bank = nn.ModuleList(BatchNormConv(in_channels, channels, k) for k in self.bank_kernels)
        self.maxpool = nn.MaxPool1d(kernel_size=2, stride=1, padding=1)
        self.conv_project1 = BatchNormConv(K * channels, proj_channels[0], 3)
        self.conv_project2 = BatchNormConv(proj_channels[0], proj_channels[1], 3, relu=False)
        self.pre_highway = nn.Linear(proj_channels[-1], channels, bias=False)
        self.highways = nn.ModuleList(HighwayNetwork(channels) for _ in range(num_highways))
        # nn.GRU's parameter names; the recurrence runs in libwrnn_mi355.so (wrnn_bigru_forward)
        self.rnn = BiGRU(channels, channels, batch_first=True, bidirectional=True)

    def forward(self, x):
        n = x.size(-1)
        bank = torch.cat([conv(x)[:, :, :n] for conv in self.conv1d_bank], dim=1)  # even kernels pad one extra
        y = self.maxpool(bank)[:, :, :n]
        y = F.dropout(y, p=self.dropout, training=self.training)
        y = self.conv_project1(y)
        y = F.dropout(y, p=self.dropout, training=self.training)
        y = self.conv_project2(y) + x
        y = self.pre_highway(y.transpose(1, 2))
        for hw in self.highways:
            y = hw(y)
        out, _ = self.rnn(y)
        return out


class EncoderPreNet(nn.Module):  # tacotron.py:86-100 (the encoder's own prenet, plain torch)
    def __init__(self, in_dims, fc1_dims=256, fc2_dims=128, dropout=0.5):
        super().__init__()
        self.fc1 = nn.Linear(in_dims, fc1_dims)
        self.fc2 = nn.Linear(fc1_dims, fc2_dims)
        self.p = dropout

    def forward(self, x):
        x = F.dropout(F.relu(self.fc1(x)), self.p, training=self.training)
        return F.dropout(F.relu(self.fc2(x)), self.p, training=self.training)


class Encoder(nn.Module):  # tacotron.py:69-83
    def __init__(self, embed_dims, num_chars, cbhg_channels, K, num_highways, dropout):
        super().__init__()
        self.embedding = nn.Embedding(num_chars, embed_dims)
        self.pre_net = EncoderPreNet(embed_dims)
        self.cbhg = CBHG(K=K, in_channels=cbhg_channels, channels=cbhg_channels,
                         proj_channels=[cbhg_channels, cbhg_channels], num_highways=num_highways)

    def forward(self, x):
        return self.cbhg(self.pre_net(self.embedding(x)).transpose(1, 2))


def aligner_distance(mel: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """-||x_t - mel_s||_2 -> [B,S,T] (tacotron.py:63-64) in libwrnn_mi355.so (wrnn_aligner_distance): the
    reference's [B,S,T,D] difference tensor is never formed.  Forward only, like every kernel of the library:
    there is no second code path, so a call that autograd is recording (training the aligner, out of scope) raises."""
    if torch.is_grad_enabled() and (mel.requires_grad or x.requires_grad):
        raise RuntimeError("wavernn_amd is a forward-only library: aligner_distance has no backward. Call the model "
                           "under torch.no_grad() (attention extraction, evaluation); train with the reference.")
    if mel.device.type != 'cuda':
        raise RuntimeError("wavernn_amd.Aligner runs only on an MI355X (HIP) device; there is no CPU fallback")
    if mel.dtype != torch.float32 or x.dtype != torch.float32:
        raise TypeError("aligner_distance expects float32 features (the reference path is fp32)")
    if mel.dim() != 3 or x.dim() != 3 or mel.shape[0] != x.shape[0] or mel.shape[2] != x.shape[2]:
        raise ValueError(f"expected mel [B,S,D] and x [B,T,D], got {tuple(mel.shape)} and {tuple(x.shape)}")
    (B, S, D), T = mel.shape, x.shape[1]
    mel, x = mel.detach().contiguous(), x.detach().contiguous()
    att = torch.empty(B, S, T, dtype=torch.float32, device=mel.device)
    with torch.cuda.device(mel.device):
        _lib.check(_lib.load().wrnn_aligner_distance(mel.data_ptr(), x.data_ptr(), B, S, T, D, att.data_ptr(),
                                                     torch.cuda.current_stream(mel.device).cuda_stream),
                   "wrnn_aligner_distance")
    return att


class Aligner(nn.Module):  # tacotron.py:13-66
    def __init__(self, num_chars: int, speaker_emb_dim: int, mel_dim: int, hidden_dim: int, out_dim: int):
        super().__init__()
        self.embedding = nn.Embedding(num_embeddings=num_chars, embedding_dim=hidden_dim)
        self.text_encoder = nn.Sequential(nn.Conv1d(hidden_dim + speaker_emb_dim, hidden_dim, kernel_size=3, padding=1))
        self.mel_encoder = nn.Sequential(nn.Conv1d(mel_dim, hidden_dim, kernel_size=3, padding=1),
                                         nn.Conv1d(hidden_dim, hidden_dim, kernel_size=3, padding=1))
        self.text_lin = nn.Linear(hidden_dim, out_dim)
        self.mel_lin = nn.Linear(hidden_dim, out_dim)
        self.speaker_emb_dim = speaker_emb_dim

    def encode_text(self, x: torch.Tensor, speaker_emb=None) -> torch.Tensor:
        """Text half of forward (tacotron.py:44-52): tokens [B,T] -> features [B,T,out]."""
        x = self.embedding(x).transpose(1, 2)
        if self.speaker_emb_dim > 0:
            spk = speaker_emb[:, :, None].expand(-1, -1, x.shape[2])
            x = torch.cat([x, spk], dim=1)
        return self.text_lin(self.text_encoder(x).transpose(1, 2))

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x = self.encode_text(batch['x'], batch.get('speaker_emb'))     # [B,T,out]
        mel = self.mel_lin(self.mel_encoder(batch['mel']).transpose(1, 2))   # [B,S,out]
        return {'x': x, 'mel': mel, 'att': aligner_distance(mel, x)}


class Tacotron(nn.Module):
    def __init__(self, embed_dims: int, num_chars: int, encoder_dims: int, decoder_dims: int, n_mels: int,
                 postnet_dims: int, encoder_k: int, lstm_dims: int, postnet_k: int, num_highways: int,
                 dropout: float, stop_threshold: float, aligner_hidden_dims: int, aligner_out_dims: int,
                 speaker_emb_dim=256) -> None:
        super().__init__()
        self.n_mels = n_mels
        self.lstm_dims = lstm_dims
        self.decoder_dims = decoder_dims
        self.encoder = Encoder(embed_dims, num_chars, encoder_dims, encoder_k, num_highways, dropout)
        proj_in = decoder_dims + speaker_emb_dim + aligner_out_dims
        self.encoder_proj_query = nn.Linear(proj_in, decoder_dims, bias=False)
        self.encoder_proj = nn.Linear(proj_in, decoder_dims, bias=False)
        self.decoder = Decoder(n_mels=n_mels, decoder_dims=decoder_dims, lstm_dims=lstm_dims,
                               aligner_out_dims=aligner_out_dims)
        self.postnet = CBHG(postnet_k, n_mels, postnet_dims, [256, 80], num_highways)
        self.post_proj = nn.Linear(postnet_dims * 2, n_mels, bias=False)
        self.speaker_emb_dim = speaker_emb_dim
        self.init_model()
        self.register_buffer('step', torch.zeros(1, dtype=torch.long))
        self.register_buffer('stop_threshold', torch.tensor(stop_threshold, dtype=torch.float32))
        # constructed after init_model(), so it keeps the framework's default init (tacotron.py:272-278)
        self.aligner = Aligner(num_chars=num_chars, mel_dim=n_mels, speaker_emb_dim=speaker_emb_dim,
                               hidden_dim=aligner_hidden_dims, out_dim=aligner_out_dims)
        self.aligner_out_dims = aligner_out_dims

    @property
    def r(self) -> int:
        return self.decoder._r_value()

    @r.setter
    def r(self, value: int) -> None:
        self.decoder.r = self.decoder.r.new_tensor(value, requires_grad=False)

    def encoder_memories(self, x, speaker_emb, aligner_text):
        """tacotron.py:318-332: CBHG encoder output ++ speaker embedding ++ aligner text features,
        projected twice -> (encoder_seq_proj_query, encoder_seq_proj), both [B,T,decoder_dims]."""
        encoder_seq = self.encoder(x)
        if self.speaker_emb_dim > 0:
            spk = speaker_emb[:, None, :].expand(-1, encoder_seq.shape[1], -1)
            encoder_seq = torch.cat([encoder_seq, spk], dim=2)
        encoder_seq = torch.cat([encoder_seq, aligner_text.detach()], dim=2)
        return self.encoder_proj_query(encoder_seq), self.encoder_proj(encoder_seq)

    def forward(self, batch: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        x, mel, speaker_emb = batch['x'], batch['mel'], batch.get('speaker_emb')
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError("wavernn_amd.Tacotron is forward-only (the decoder loop runs in libwrnn_mi355.so, which "
                               "has no backward): call it under torch.no_grad().  Training-mode forwards are supported "
                               "there too: prenet dropout and LSTM zoneout follow .train() as in the reference.")
        if self.training:
            self.step += 1
        aligner_out = self.aligner(batch)
        enc_q, enc_p = self.encoder_memories(x, speaker_emb, aligner_out['x'])
        mel_in = torch.cat([mel, aligner_out['mel'].transpose(1, 2).detach()], dim=1)
        # the whole `for t in range(0, steps, r)` loop (tacotron.py:339-348,356) in one library call
        mel_outputs, attn_scores = self.decoder.decode_teacher_forced(enc_q.detach(), enc_p.detach(), mel_in)
        linear = self.post_proj(self.postnet(mel_outputs)).transpose(1, 2)
        return {'mel': mel_outputs, 'mel_post': linear, 'att': attn_scores, 'att_aligner': aligner_out['att']}

    @torch.no_grad()
    def generate(self, x, steps: int = 1000, speaker_emb=None, aligner_feats=None, noise=None
                 ) -> Dict[str, torch.Tensor]:
        """Free-running synthesis.  NOT in the reference at the mounted commit; the
        build-defined contract of SURVEY.md 8b: feedback = last generated frame ++ ``aligner_feats``
        [B,aligner_out_dims,steps] (zeros when None), stop once a group after the 10th has every
        frame < ``stop_threshold``, at most ``steps`` decoder steps.  Returns the keys of ``forward``
        minus ``att_aligner``."""
        enc_q, enc_p = self.encoder_memories(x, speaker_emb, self.aligner.encode_text(x, speaker_emb))
        mel_outputs, attn_scores, _ = self.decoder.decode_free(
            enc_q, enc_p, int(steps), aligner_feats=aligner_feats,
            stop_threshold=float(self.stop_threshold.item()), noise=noise)
        linear = self.post_proj(self.postnet(mel_outputs)).transpose(1, 2)
        return {'mel': mel_outputs, 'mel_post': linear, 'att': attn_scores}

    def init_model(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    def get_step(self):
        return self.step.data.item()

    def reset_step(self):
        self.step = self.step.data.new_tensor(1)

    @classmethod
    def from_config(cls, config: Dict[str, Any], num_chars: int) -> 'Tacotron':
        """``num_chars`` = len(phonemes) of the reference's symbol table (135 at the mounted commit)."""
        model_config = dict(config['tacotron']['model'])
        model_config['num_chars'] = num_chars
        model_config['n_mels'] = config['dsp']['num_mels']
        return cls(**model_config)
