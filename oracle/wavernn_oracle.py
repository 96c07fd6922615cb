"""CPU oracle for the WaveRNN generation path.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's cpu-baseline leg).

**PARITY UNPINNED.**  The WaveRNN vocoder is absent from the mounted reference tree: no model source, no
golden vector, no test.  What follows restates, from public knowledge of the
fatchord WaveRNN that the reference forked from (reference README.md:4, :187) and from the
hyper-parameters the tree still holds (tests/resources/test_config.yaml:47-75), the model that the
build defines; the HIP path is checked against THIS file only and nothing here may be reported as
reference parity.

W: dict of torch tensors keyed like the mirror's state_dict (wavernn_amd/wavernn.py).
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _bn(x, W, p):
    return F.batch_norm(x, W[p + ".running_mean"], W[p + ".running_var"], W[p + ".weight"], W[p + ".bias"], False, 0.0, 1e-5)


def mel_resnet(W, x, res_blocks):
    """conv_in (k = 2*pad+1, no bias) -> BN -> relu -> res_blocks x [1x1 conv, BN, relu, 1x1 conv, BN, +x] -> 1x1 conv_out."""
    x = F.relu(_bn(F.conv1d(x, W["upsample.resnet.conv_in.weight"]), W, "upsample.resnet.batch_norm"))
    for i in range(res_blocks):
        p = f"upsample.resnet.layers.{i}."
        r = x
        x = F.relu(_bn(F.conv1d(x, W[p + "conv1.weight"]), W, p + "batch_norm1"))
        x = _bn(F.conv1d(x, W[p + "conv2.weight"]), W, p + "batch_norm2")
        x = x + r
    return F.conv1d(x, W["upsample.resnet.conv_out.weight"], W["upsample.resnet.conv_out.bias"])


def upsample(W, mels, upsample_factors, res_blocks, pad):
    """mels [B, feat, frames + 2*pad] -> (mels_up [B, frames*hop, feat], aux [B, frames*hop, res_out])."""
    total = int(np.prod(upsample_factors))
    aux = mel_resnet(W, mels, res_blocks)
    aux = aux.repeat_interleave(total, dim=2)                       # Stretch2d(total, 1)
    m = mels.unsqueeze(1)
    for i, s in enumerate(upsample_factors):
        m = m.repeat_interleave(s, dim=3)                           # Stretch2d(s, 1)
        m = F.conv2d(m, W[f"upsample.up_layers.{2 * i + 1}.weight"], padding=(0, s))
    indent = pad * total
    m = m.squeeze(1)[:, :, indent:-indent]
    return m.transpose(1, 2), aux.transpose(1, 2)


def fold_with_overlap(x, target, overlap):
    """[1, L, C] -> [folds, target + 2*overlap, C], consecutive folds sharing `overlap` steps."""
    _, total_len, features = x.shape
    num_folds = (total_len - overlap) // (target + overlap)
    extended_len = num_folds * (overlap + target) + overlap
    remaining = total_len - extended_len
    if remaining != 0:
        num_folds += 1
        padding = target + 2 * overlap - remaining
        x = F.pad(x, (0, 0, 0, padding))
    folded = torch.zeros(num_folds, target + 2 * overlap, features)
    for i in range(num_folds):
        start = i * (target + overlap)
        folded[i] = x[0, start:start + target + 2 * overlap, :]
    return folded


def xfade_and_unfold(y: np.ndarray, overlap: int) -> np.ndarray:
    """[folds, target + 2*overlap] -> [L]: equal-power crossfade over the shared steps."""
    num_folds, length = y.shape
    target = length - 2 * overlap
    total_len = num_folds * (target + overlap) + overlap
    silence_len = overlap // 2
    fade_len = overlap - silence_len
    t = np.linspace(-1, 1, fade_len, dtype=np.float64)
    fade_in = np.concatenate([np.zeros(silence_len), np.sqrt(0.5 * (1 + t))])
    fade_out = np.concatenate([np.ones(silence_len), np.sqrt(0.5 * (1 - t))])
    y = y.astype(np.float64).copy()
    y[:, :overlap] *= fade_in
    y[:, -overlap:] *= fade_out
    out = np.zeros(total_len, dtype=np.float64)
    for i in range(num_folds):
        start = i * (target + overlap)
        out[start:start + target + 2 * overlap] += y[i]
    return out


def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    gi = x @ w_ih.T + b_ih
    gh = h @ w_hh.T + b_hh
    R = h.shape[1]
    r = torch.sigmoid(gi[:, :R] + gh[:, :R])
    z = torch.sigmoid(gi[:, R:2 * R] + gh[:, R:2 * R])
    n = torch.tanh(gi[:, 2 * R:] + r * gh[:, 2 * R:])
    return (1 - z) * n + z * h


def sample_raw(logits, u):
    """Inverse-CDF categorical sampling: class = #{c : cdf_c <= u * total}, softmax terms in fp32,
    accumulated in float64 (what the HIP sampler does)."""
    m = logits.max(dim=1, keepdim=True).values
    e = torch.exp(logits - m).double()
    cdf = torch.cumsum(e, dim=1)
    thr = u.double()[:, None] * cdf[:, -1:]
    c = (cdf <= thr).sum(dim=1).clamp(max=logits.shape[1] - 1)
    return c


def sample_mol(y, u_mix, u_logistic, log_scale_min=None):
    """Sample from a discretized mixture of logistics, restating the reference's
    utils/distribution.py:87-123 (`sample_from_discretized_mix_logistic`) with its two internal
    `uniform_(1e-5, 1 - 1e-5)` draws (:106, :118) replaced by the caller's uniforms.  This is the ONE piece of
    the WaveRNN path that the mounted reference still holds, so unlike the rest of this file it IS pinned:
    tests/test_oracle_golden.py checks it against tests/golden/mol_sampler.npz, which tools/make_golden.py
    produced by calling the reference function itself.
    y [B, 3*nr_mix, T] (logit_probs | means | log_scales along dim 1), u_mix [B, T, nr_mix], u_logistic [B, T]
    -> (x [B, T] in [-1, 1], mixture index [B, T])."""
    if log_scale_min is None:
        log_scale_min = float(np.log(1e-14))                                   # :96-97
    assert y.shape[1] % 3 == 0                                                 # :98
    nr_mix = y.shape[1] // 3
    y = y.transpose(1, 2)                                                      # :102  B x T x C
    logit_probs = y[:, :, :nr_mix]
    temp = logit_probs - torch.log(-torch.log(u_mix))                          # :107  Gumbel-max
    argmax = temp.max(dim=-1).indices                                          # :108
    sel = torch.gather(y, 2, (argmax[:, :, None] + torch.tensor([nr_mix, 2 * nr_mix])).long())   # :111-115 one-hot select
    means = sel[:, :, 0]
    log_scales = torch.clamp(sel[:, :, 1], min=log_scale_min)
    x = means + torch.exp(log_scales) * (torch.log(u_logistic) - torch.log(1.0 - u_logistic))      # :119
    return torch.clamp(torch.clamp(x, min=-1.0), max=1.0), argmax                                  # :121


def generate_loop(W, mels_up, aux, u, x_forced=None, return_logits=False, mode="RAW"):
    """mels_up [B, seq, feat], aux [B, seq, 4a], u [B, seq] (RAW) or [B, seq, nr_mix + 1] (MOL: the mixture
    uniforms, then the logistic one) -> samples [B, seq] (and logits [B, seq, C])."""
    B, seq, _ = mels_up.shape
    R = W["rnn1.weight_hh_l0"].shape[1]
    C = W["fc3.weight"].shape[0]
    a = aux.shape[2] // 4
    h1 = torch.zeros(B, R)
    h2 = torch.zeros(B, R)
    x = torch.zeros(B, 1)
    out = torch.zeros(B, seq)
    logits_all = torch.zeros(B, seq, C) if return_logits else None
    g1 = [W["rnn1." + k] for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    g2 = [W["rnn2." + k] for k in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0")]
    for t in range(seq):
        a1, a2, a3, a4 = (aux[:, t, i * a:(i + 1) * a] for i in range(4))
        xx = torch.cat([x, mels_up[:, t], a1], dim=1) @ W["I.weight"].T + W["I.bias"]
        h1 = gru_cell(xx, h1, *g1)
        xx = xx + h1
        h2 = gru_cell(torch.cat([xx, a2], dim=1), h2, *g2)
        xx = xx + h2
        xx = F.relu(torch.cat([xx, a3], dim=1) @ W["fc1.weight"].T + W["fc1.bias"])
        xx = F.relu(torch.cat([xx, a4], dim=1) @ W["fc2.weight"].T + W["fc2.bias"])
        logits = xx @ W["fc3.weight"].T + W["fc3.bias"]
        if return_logits:
            logits_all[:, t] = logits
        if mode == "MOL":
            out[:, t] = sample_mol(logits[:, :, None], u[:, t, None, :-1], u[:, t, None, -1])[0][:, 0]
        else:
            c = sample_raw(logits, u[:, t])
            out[:, t] = 2.0 * c.float() / (C - 1.0) - 1.0
        x = (x_forced[:, t] if x_forced is not None else out[:, t])[:, None]
    return (out, logits_all) if return_logits else out


def generate(W, mels, cfg, u, batched=True, target=11000, overlap=550):
    """Whole WaveRNN.generate in RAW mode: mels [1, feat, frames] -> audio [frames * hop] (numpy float64).
    u: uniforms [folds, steps] for the loop."""
    pad, hop = cfg["pad"], int(np.prod(cfg["upsample_factors"]))
    wave_len = (mels.shape[-1] - 1) * hop
    m = F.pad(mels, (pad, pad))
    mels_up, aux = upsample(W, m, cfg["upsample_factors"], cfg["res_blocks"], pad)
    if batched:
        mels_up = fold_with_overlap(mels_up, target, overlap)
        aux = fold_with_overlap(aux, target, overlap)
    out = generate_loop(W, mels_up, aux, u).numpy().astype(np.float64)
    out = xfade_and_unfold(out, overlap) if batched else out[0]
    fade = np.linspace(1, 0, 20 * hop)
    out = out[:wave_len]
    out[-20 * hop:] *= fade
    return out
