"""CPU ORACLE -- test infrastructure only, never the product path.

A from-scratch, functional (stateless) PyTorch-CPU restatement of the
reference hot path, used ONLY by ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg as the checker / timed CPU baseline.
Nothing under ``wavernn_amd/`` may import this module.

Pinned: ``tests/test_oracle_golden.py`` checks every function here against
``tests/golden/*.npz``, which ``tools/make_golden.py`` produced in the build
container by importing and running the real reference classes
(``/root/reference/models/tacotron.py`` Decoder, ``models/common_layers.py``
LengthRegulator).  The reference's own tests hold no vector for this path
(SURVEY.md section 4), so those generated fixtures are the pin.

What each function follows (reference file:line):
  prenet             models/tacotron.py:93-100
  gru_cell           torch.nn.GRUCell as used at models/tacotron.py:174,199-200
  lsa_attention      models/tacotron.py:132-156
  lstm_cell          torch.nn.LSTMCell as used at models/tacotron.py:176-177,213-227
  decoder_step       models/tacotron.py:185-235
  run_teacher_forced models/tacotron.py:299-316,334-348,356 (decoder loop only)
  run_free           build-defined feedback rule; the
                     per-step arithmetic is decoder_step
  length_regulate    models/common_layers.py:17-24

Noise is explicit: dropout keeps an element iff u >= p and scales by
1/(1-p); zoneout keeps the previous hidden value iff u < 0.1.
"""
from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch

MAX_R = 20          # models/tacotron.py:162
PRENET_P = 0.5      # models/tacotron.py:87
ZONEOUT_P = 0.1     # models/tacotron.py:180


def to_torch(sd, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in sd.items():
        t = torch.as_tensor(v)
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def _dropout(x, u, p):
    if u is None:
        return x
    return x * (u >= p).to(x.dtype) * (1.0 / (1.0 - p))


def prenet(W, x, drop_u: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """relu(fc1) -> dropout -> relu(fc2) -> dropout. drop_u = (u1[B,256], u2[B,128]) or None."""
    h = torch.relu(x @ W["prenet.fc1.weight"].T + W["prenet.fc1.bias"])
    h = _dropout(h, None if drop_u is None else drop_u[0], PRENET_P)
    h = torch.relu(h @ W["prenet.fc2.weight"].T + W["prenet.fc2.bias"])
    h = _dropout(h, None if drop_u is None else drop_u[1], PRENET_P)
    return h


def gru_cell(W, x, h):
    gi = x @ W["attn_rnn.weight_ih"].T + W["attn_rnn.bias_ih"]
    gh = h @ W["attn_rnn.weight_hh"].T + W["attn_rnn.bias_hh"]
    d = h.shape[1]
    r = torch.sigmoid(gi[:, :d] + gh[:, :d])
    z = torch.sigmoid(gi[:, d:2 * d] + gh[:, d:2 * d])
    n = torch.tanh(gi[:, 2 * d:] + r * gh[:, 2 * d:])
    return (1.0 - z) * n + z * h


def lsa_attention(W, enc_p, query, cumulative, attention):
    """Location-sensitive attention. Returns (scores[B,T], cumulative', attention')."""
    pq = query @ W["attn_net.W.weight"].T + W["attn_net.W.bias"]            # [B,D]
    loc = torch.stack([cumulative, attention], dim=1)                       # [B,2,T]
    k = W["attn_net.conv.weight"]
    f = torch.nn.functional.conv1d(loc, k, padding=(k.shape[2] - 1) // 2)    # [B,F,T]
    pl = f.transpose(1, 2) @ W["attn_net.L.weight"].T + W["attn_net.L.bias"]  # [B,T,D]
    u = torch.tanh(pq[:, None, :] + enc_p + pl) @ W["attn_net.v.weight"][0]   # [B,T]
    s = torch.softmax(u, dim=1)      # over ALL T, no padding mask (tacotron.py:147-152)
    return s, cumulative + s, s


def lstm_cell(W, name, x, h, c):
    g = x @ W[name + ".weight_ih"].T + W[name + ".bias_ih"] \
        + h @ W[name + ".weight_hh"].T + W[name + ".bias_hh"]
    l = h.shape[1]
    i, f, gg, o = g[:, :l], g[:, l:2 * l], g[:, 2 * l:3 * l], g[:, 3 * l:]
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def _zoneout(prev, cur, u):
    if u is None:
        return cur
    m = (u < ZONEOUT_P).to(cur.dtype)
    return prev * m + cur * (1.0 - m)


def decoder_step(W, enc_q, enc_p, prenet_in, hidden, cell, context, lsa, t: int, r: int,
                 drop_u=None, zone_u=None):
    """One decoder step.

    hidden=(attn_h,h1,h2), cell=(c1,c2), lsa=(cumulative,attention) or None at t==0.
    Returns (mels[B,n_mels,r], scores[B,1,T], hidden', cell', context', lsa')."""
    attn_h, h1, h2 = hidden
    c1, c2 = cell
    B, T, _ = enc_p.shape
    if t == 0 or lsa is None:
        z = torch.zeros(B, T, dtype=enc_p.dtype)
        lsa = (z, z.clone())
    p = prenet(W, prenet_in, drop_u)
    attn_h = gru_cell(W, torch.cat([context, p], dim=1), attn_h)
    s, cum, att = lsa_attention(W, enc_p, attn_h, lsa[0], lsa[1])
    context = (s[:, None, :] @ enc_q)[:, 0, :]
    x = torch.cat([context, attn_h], dim=1) @ W["rnn_input.weight"].T + W["rnn_input.bias"]
    h1n, c1 = lstm_cell(W, "res_rnn1", x, h1, c1)
    h1 = _zoneout(h1, h1n, None if zone_u is None else zone_u[0])
    x = x + h1
    h2n, c2 = lstm_cell(W, "res_rnn2", x, h2, c2)
    h2 = _zoneout(h2, h2n, None if zone_u is None else zone_u[1])
    x = x + h2
    n_mels = W["mel_proj.weight"].shape[0] // MAX_R
    mels = (x @ W["mel_proj.weight"].T).view(B, n_mels, MAX_R)[:, :, :r]
    return mels, s[:, None, :], (attn_h, h1, h2), (c1, c2), context, (cum, att)


def zero_state(W, B, dtype=torch.float32):
    d = W["attn_net.W.weight"].shape[0]
    l = W["rnn_input.weight"].shape[0]
    z = lambda n: torch.zeros(B, n, dtype=dtype)
    return (z(d), z(l), z(l)), (z(l), z(l)), z(d)


def run_teacher_forced(W, enc_q, enc_p, mel_in, r: int = 1,
                       drop_u: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                       zone_u: Optional[Tuple[torch.Tensor, torch.Tensor]] = None):
    """The decoder loop of Tacotron.forward. mel_in[B,n_mels+ali,S].

    drop_u = (u1[n_steps,B,256], u2[n_steps,B,128]); zone_u = (z1[n_steps,B,L], z2[...]).
    Returns mel[B,n_mels,n_steps*r], att[B,n_steps,T]."""
    B, _, S = mel_in.shape
    hidden, cell, ctx = zero_state(W, B, mel_in.dtype)
    go = torch.zeros(B, mel_in.shape[1], dtype=mel_in.dtype)
    lsa = None
    mels, atts = [], []
    for k, t in enumerate(range(0, S, r)):
        pin = mel_in[:, :, t - 1] if t > 0 else go
        du = None if drop_u is None else (drop_u[0][k], drop_u[1][k])
        zu = None if zone_u is None else (zone_u[0][k], zone_u[1][k])
        m, s, hidden, cell, ctx, lsa = decoder_step(W, enc_q, enc_p, pin, hidden, cell, ctx,
                                                    lsa, t, r, du, zu)
        mels.append(m)
        atts.append(s)
    return torch.cat(mels, dim=2), torch.cat(atts, dim=1)


def run_free(W, enc_q, enc_p, max_steps: int, r: int = 1, aligner_feats=None,
             stop_threshold: Optional[float] = None, drop_u=None):
    """Free-running decode: feeds its own last mel frame back.

    prenet_in = cat[last generated frame, aligner_feats[:, :, k] or zeros]; stops after a
    group whose frames are all < stop_threshold once more than 10 groups have run.
    Returns mel[B,n_mels,n*r], att[B,n,T], n."""
    B = enc_q.shape[0]
    n_mels = W["mel_proj.weight"].shape[0] // MAX_R
    pin_dim = W["prenet.fc1.weight"].shape[1]
    hidden, cell, ctx = zero_state(W, B, enc_q.dtype)
    pin = torch.zeros(B, pin_dim, dtype=enc_q.dtype)
    lsa = None
    mels, atts = [], []
    for k in range(max_steps):
        du = None if drop_u is None else (drop_u[0][k], drop_u[1][k])
        m, s, hidden, cell, ctx, lsa = decoder_step(W, enc_q, enc_p, pin, hidden, cell, ctx,
                                                    lsa, k * r, r, du, None)
        mels.append(m)
        atts.append(s)
        pin = torch.zeros(B, pin_dim, dtype=enc_q.dtype)
        pin[:, :n_mels] = m[:, :, -1]
        if aligner_feats is not None:
            pin[:, n_mels:] = aligner_feats[:, :, k]
        if stop_threshold is not None and k > 10 and bool((m < stop_threshold).all()):
            break
    return torch.cat(mels, dim=2), torch.cat(atts, dim=1), len(mels)


def aligner_distance(mel_feats: torch.Tensor, text_feats: torch.Tensor) -> torch.Tensor:
    """Aligner attention map (models/tacotron.py:63-64): att[b,s,t] = -||text[b,t] - mel[b,s]||_2,
    mel_feats [B,S,D], text_feats [B,T,D] -> [B,S,T]."""
    d = text_feats.unsqueeze(1) - mel_feats.unsqueeze(2)
    return -(d * d).sum(dim=-1).sqrt()


def length_regulate(x: torch.Tensor, dur: torch.Tensor) -> torch.Tensor:
    """Duration-driven expand; clamps negatives of ``dur`` IN PLACE like the reference."""
    dur[dur < 0] = 0.0
    reps = (dur + 0.5).long()
    lens = reps.sum(dim=1)
    M = int(lens.max()) if x.shape[0] > 0 else 0
    out = torch.zeros(x.shape[0], M, x.shape[2], dtype=x.dtype)
    for i in range(x.shape[0]):
        idx = torch.repeat_interleave(torch.arange(x.shape[1]), reps[i])
        out[i, : idx.numel()] = x[i, idx]
    return out


def bilstm(W, x: torch.Tensor, lengths=None, padding_value: float = 0.0):
    """Bidirectional one-layer LSTM with packed-sequence semantics, restated with explicit loops
    over ``lstm_cell``: what `pack_padded_sequence -> nn.LSTM(bidirectional) -> pad_packed_sequence`
    computes at models/forward_tacotron.py:147-152 (lengths given) and `self.lstm(x)` at :224
    (lengths None).  W: torch.nn.LSTM parameter names -> tensors.  x [B,M,in] ->
    (out [B,M,2H], h_n [2,B,H], c_n [2,B,H])."""
    B, M, _ = x.shape
    H = W["weight_hh_l0"].shape[1]
    lens = [M] * B if lengths is None else [int(v) for v in lengths]
    out = torch.full((B, M, 2 * H), float(padding_value), dtype=x.dtype)
    h_n = torch.zeros(2, B, H, dtype=x.dtype)
    c_n = torch.zeros(2, B, H, dtype=x.dtype)
    for d, sfx in enumerate(("", "_reverse")):
        cell = {"c." + k: W[k + "_l0" + sfx] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        for b in range(B):
            h = torch.zeros(1, H, dtype=x.dtype)
            c = torch.zeros(1, H, dtype=x.dtype)
            order = range(lens[b]) if d == 0 else range(lens[b] - 1, -1, -1)
            for t in order:
                h, c = lstm_cell(cell, "c", x[b:b + 1, t], h, c)
                out[b, t, d * H:(d + 1) * H] = h[0]
            h_n[d, b], c_n[d, b] = h[0], c[0]
    return out, h_n, c_n


def bigru(W, x: torch.Tensor, lengths=None, padding_value: float = 0.0):
    """Bidirectional one-layer GRU, restated with explicit loops over ``gru_cell``: what
    `self.rnn(x)` computes at the end of the reference's CBHG block (models/common_layers.py:89,123;
    ``nn.GRU(channels, channels, batch_first=True, bidirectional=True)``, zero initial state), with
    packed-sequence semantics when ``lengths`` is given.  W: torch.nn.GRU parameter names -> tensors.
    x [B,M,in] -> (out [B,M,2H], h_n [2,B,H])."""
    B, M, _ = x.shape
    H = W["weight_hh_l0"].shape[1]
    lens = [M] * B if lengths is None else [int(v) for v in lengths]
    out = torch.full((B, M, 2 * H), float(padding_value), dtype=x.dtype)
    h_n = torch.zeros(2, B, H, dtype=x.dtype)
    for d, sfx in enumerate(("", "_reverse")):
        cell = {"attn_rnn." + k: W[k + "_l0" + sfx] for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
        for b in range(B):
            h = torch.zeros(1, H, dtype=x.dtype)
            order = range(lens[b]) if d == 0 else range(lens[b] - 1, -1, -1)
            for t in order:
                h = gru_cell(cell, x[b:b + 1, t], h)
                out[b, t, d * H:(d + 1) * H] = h[0]
            h_n[d, b] = h[0]
    return out, h_n
