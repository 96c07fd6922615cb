"""Deterministic synthetic weights and inputs (counter-based, numpy only).

There is no network for checkpoints or datasets, so tests, goldens and the
benchmark all draw weights/inputs from this generator.  It is counter-based
(splitmix64 over ``(seed, stream, index)``) and uses only integer ops plus
exact float adds/multiplies -- no libm transcendental -- so the same call
returns bit-identical arrays in the build container and on the GPU box.

Parameter names/shapes follow the reference ``Decoder`` state_dict
(reference models/tacotron.py:164-178; key list in SURVEY.md section 8b).
"""
from __future__ import annotations

import zlib
from typing import Dict

import numpy as np

_M1 = np.uint64(0x9E3779B97F4A7C15)
_M2 = np.uint64(0xBF58476D1CE4E5B9)
_M3 = np.uint64(0x94D049BB133111EB)

PRENET_FC1 = 256   # reference PreNet defaults, tacotron.py:87
PRENET_FC2 = 128
LSA_KERNEL = 31    # reference LSA defaults, tacotron.py:123
LSA_FILTERS = 32
MAX_R = 20         # reference Decoder.max_r, tacotron.py:162


def _mix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = z + _M1
        z = (z ^ (z >> np.uint64(30))) * _M2
        z = (z ^ (z >> np.uint64(27))) * _M3
        return z ^ (z >> np.uint64(31))


def _stream_id(name) -> int:
    if isinstance(name, int):
        return name
    return zlib.crc32(str(name).encode())


def uniform01(seed: int, stream, n: int) -> np.ndarray:
    """n float32 values in [0, 1), each a multiple of 2**-24."""
    base = _mix(np.array([(int(seed) << 32) ^ _stream_id(stream)], dtype=np.uint64))
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _mix(base + idx * _M1)
    return ((z >> np.uint64(40)).astype(np.float64) * (1.0 / (1 << 24))).astype(np.float32)


def uniform(seed: int, stream, shape, lo: float = 0.0, hi: float = 1.0) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, stream, n).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def normalish(seed: int, stream, shape) -> np.ndarray:
    """Unit-variance, zero-mean bell-shaped values (Irwin-Hall of 4 uniforms)."""
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(seed, stream, 4 * n).astype(np.float64).reshape(4, n)
    v = (u.sum(axis=0) - 2.0) * 1.7320508075688772
    return v.astype(np.float32).reshape(shape)


def _xavier(seed, name, shape) -> np.ndarray:
    # fan computation as torch.nn.init.xavier_uniform_ (receptive field folded in)
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    bound = float(np.sqrt(6.0 / (fan_in + fan_out)))
    return uniform(seed, name, shape, -bound, bound)


def decoder_param_shapes(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32):
    d, l = decoder_dims, lstm_dims
    pin = n_mels + aligner_out_dims
    return {
        "prenet.fc1.weight": (PRENET_FC1, pin), "prenet.fc1.bias": (PRENET_FC1,),
        "prenet.fc2.weight": (PRENET_FC2, PRENET_FC1), "prenet.fc2.bias": (PRENET_FC2,),
        "attn_net.conv.weight": (LSA_FILTERS, 2, LSA_KERNEL),
        "attn_net.L.weight": (d, LSA_FILTERS), "attn_net.L.bias": (d,),
        "attn_net.W.weight": (d, d), "attn_net.W.bias": (d,),
        "attn_net.v.weight": (1, d),
        "attn_rnn.weight_ih": (3 * d, d + d // 2), "attn_rnn.weight_hh": (3 * d, d),
        "attn_rnn.bias_ih": (3 * d,), "attn_rnn.bias_hh": (3 * d,),
        "rnn_input.weight": (l, 2 * d), "rnn_input.bias": (l,),
        "res_rnn1.weight_ih": (4 * l, l), "res_rnn1.weight_hh": (4 * l, l),
        "res_rnn1.bias_ih": (4 * l,), "res_rnn1.bias_hh": (4 * l,),
        "res_rnn2.weight_ih": (4 * l, l), "res_rnn2.weight_hh": (4 * l, l),
        "res_rnn2.bias_ih": (4 * l,), "res_rnn2.bias_hh": (4 * l,),
        "mel_proj.weight": (n_mels * MAX_R, l),
    }


def decoder_state_dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32,
                       seed=0, peaky: float = 1.0, r: int = 1) -> Dict[str, np.ndarray]:
    """Random-init decoder weights: xavier-uniform matrices, small uniform biases.

    ``peaky`` scales the LSA ``v`` and ``W`` weights so the softmax is not
    almost uniform (random-init attention peaks at <0.1 for T=100)."""
    sd: Dict[str, np.ndarray] = {"r": np.array(r, dtype=np.int32)}
    for name, shape in decoder_param_shapes(n_mels, decoder_dims, lstm_dims, aligner_out_dims).items():
        if len(shape) > 1:
            w = _xavier(seed, name, shape)
        else:
            w = uniform(seed, name, shape, -0.1, 0.1)
        if peaky != 1.0 and name in ("attn_net.v.weight", "attn_net.W.weight"):
            w = (w * np.float32(peaky)).astype(np.float32)
        sd[name] = w
    return sd


def decoder_inputs(B, T, S, n_mels=80, decoder_dims=256, aligner_out_dims=32, seed=1):
    """Encoder memories and a teacher-forcing mel_in."""
    enc_q = normalish(seed, "enc_q", (B, T, decoder_dims))
    enc_p = normalish(seed, "enc_p", (B, T, decoder_dims))
    mel = uniform(seed + 1, "mel", (B, n_mels, S), -11.5129, 2.0)
    ali = normalish(seed + 1, "ali", (B, aligner_out_dims, S))
    mel_in = np.concatenate([mel, ali], axis=1)
    return enc_q, enc_p, np.ascontiguousarray(mel_in)


def lr_inputs(B, T, C, seed=4, dur_lo=1, dur_hi=11):
    """Length-regulator inputs: x ~ N(0,1)-ish, integer durations in [lo, hi]."""
    x = normalish(seed, "lr_x", (B, T, C))
    u = uniform01(seed, "lr_dur", B * T).astype(np.float64)
    dur = np.floor(dur_lo + u * (dur_hi - dur_lo + 1)).astype(np.float32).reshape(B, T)
    return x, dur


def fill_state_dict(shapes: Dict[str, tuple], seed: int = 0, peaky: float = 1.0) -> Dict[str, np.ndarray]:
    """Deterministic values for the FLOATING-POINT entries of a whole-model state_dict
    (name -> shape): xavier-uniform matrices/conv kernels, small biases, batch-norm statistics
    that keep activations sane.  Used for the full-Tacotron fixtures."""
    out: Dict[str, np.ndarray] = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith("running_var"):
            w = uniform(seed, name, shape, 0.5, 1.5)
        elif name.endswith("running_mean"):
            w = uniform(seed, name, shape, -0.1, 0.1)
        elif name.endswith("bnorm.weight"):
            w = uniform(seed, name, shape, 0.8, 1.2)
        elif len(shape) >= 2:
            w = _xavier(seed, name, shape)
        elif len(shape) == 1:
            w = uniform(seed, name, shape, -0.1, 0.1)
        else:
            continue
        if peaky != 1.0 and name.endswith(("attn_net.v.weight", "attn_net.W.weight")):
            w = (w * np.float32(peaky)).astype(np.float32)
        out[name] = w
    return out


def forward_model_inputs(cfg, B, T, seed, multi):
    """Seeded batch for the forward models (tools/make_golden.py::case_forward_model and tests/test_forward_models.py):
    token id 0 pads ragged rows."""
    x = 1 + (uniform01(seed, "tok", B * T) * (cfg["num_chars"] - 1)).astype(np.int64).reshape(B, T)
    _, dur = lr_inputs(B, T, 4, seed=seed, dur_lo=0, dur_hi=4)
    for b in range(1, B):
        x[b, T - b:] = 0
        dur[b, T - b:] = 0.0
    mel_len = dur.sum(axis=1).astype(np.int64)
    S = int(mel_len.max()) + 2                     # exercises pad
    batch = {"x": x, "mel": uniform(seed, "mel", (B, 80, S), -11.5129, 2.0), "dur": dur, "mel_len": mel_len,
             "pitch": normalish(seed, "pitch", (B, T)), "energy": normalish(seed, "energy", (B, T))}
    if multi:
        batch["speaker_emb"] = normalish(seed, "semb", (B, cfg["speaker_emb_dims"]))
        batch["pitch_cond"] = (uniform01(seed, "pcond", B * T) * 4).astype(np.int64).reshape(B, T)
    return batch


# Hyper-parameters of the reference's single-speaker ForwardTacotron (configs/singlespeaker.yaml forward_tacotron.model,
# num_chars = 135 symbols, n_mels = 80) and of the WaveRNN whose hparams survive in tests/resources/test_config.yaml:47-75:
# the shapes BASELINE configs[1], [3] and [4] are quoted on.
FORWARD_TACOTRON_CFG = dict(embed_dims=256, series_embed_dims=64, num_chars=135, durpred_conv_dims=256, durpred_rnn_dims=64,
                            durpred_dropout=0.5, pitch_conv_dims=256, pitch_rnn_dims=128, pitch_dropout=0.5,
                            pitch_strength=1.0, energy_conv_dims=256, energy_rnn_dims=64, energy_dropout=0.5,
                            energy_strength=1.0, rnn_dims=512, prenet_dims=256, prenet_k=16, postnet_num_highways=4,
                            prenet_dropout=0.5, postnet_dims=256, postnet_k=8, prenet_num_highways=4, postnet_dropout=0.0,
                            n_mels=80)
WAVERNN_CFG = dict(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
                   res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050)


def utterance_tokens(n: int, seed: int = 6, lengths=(12, 16, 20), num_chars: int = 135):
    """n synthetic utterances (int64 token arrays, ids 1..num_chars-1), lengths cycling through `lengths`."""
    out = []
    for i in range(n):
        T = lengths[i % len(lengths)]
        out.append(1 + (uniform01(seed, f"utt{i}", T) * (num_chars - 1)).astype(np.int64))
    return out
