#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/make_golden.py
Needs /root/reference (read-only); never runs on the GPU box.  Only data is
written: inputs are re-creatable from wavernn_amd.synth seeds (recorded in each
file's ``recipe``), outputs are what the reference classes returned.

Harness notes (SURVEY.md section 8c "gotchas"):
 * LSA keeps per-module state -> every trajectory starts at t == 0 on a fresh Decoder.
 * dropout / zoneout are made reproducible by swapping torch.nn.functional.dropout and
   Decoder.zoneout for versions that consume caller-supplied uniforms
   (keep iff u >= p; zoneout mask = u < p).
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

import torch.nn.functional as F  # noqa: E402
from models.tacotron import Decoder  # noqa: E402  (the reference)
from models.common_layers import LengthRegulator  # noqa: E402  (the reference)

from wavernn_amd import synth  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(1)

_noise_queue = []
_orig_dropout = F.dropout


def _queued_dropout(x, p=0.5, training=True, inplace=False):
    if not training:
        return x
    u = _noise_queue.pop(0)
    assert u.shape == x.shape, (u.shape, x.shape)
    return x * (u >= p).to(x.dtype) / (1.0 - p)


_zone_queue = []


def _queued_zoneout(self, prev, current, p=0.1):
    u = _zone_queue.pop(0)
    mask = (u < p).to(current.dtype)
    return prev * mask + current * (1 - mask)


F.dropout = _queued_dropout
Decoder.zoneout = _queued_zoneout


def ref_decoder(dims, seed, peaky, r):
    sd = synth.decoder_state_dict(**dims, seed=seed, peaky=peaky, r=r)
    dec = Decoder(**dims)
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})
    dec.eval()
    return dec


def t(a):
    return torch.as_tensor(a)


def zero_state(dec, B, dims):
    d, l = dims["decoder_dims"], dims["lstm_dims"]
    return ((torch.zeros(B, d), torch.zeros(B, l), torch.zeros(B, l)),
            (torch.zeros(B, l), torch.zeros(B, l)), torch.zeros(B, d))


def case_steps(name, dims, B, T, r, seed, peaky, mode):
    """Three chained single steps (t = 0, r, 2r); every intermediate is recorded."""
    dec = ref_decoder(dims, seed, peaky, r)
    if mode == "prenet_train":
        dec.prenet.train()
    elif mode == "train":
        dec.train()
    enc_q, enc_p, mel_in = synth.decoder_inputs(B, T, 3 * r + 1, dims["n_mels"], dims["decoder_dims"],
                                                dims["aligner_out_dims"], seed=seed + 10)
    enc_q, enc_p, mel_in = t(enc_q), t(enc_p), t(mel_in)
    hidden, cell, ctx = zero_state(dec, B, dims)
    out = {}
    n_steps = 3
    du1 = synth.uniform(seed + 20, "du1", (n_steps, B, synth.PRENET_FC1))
    du2 = synth.uniform(seed + 20, "du2", (n_steps, B, synth.PRENET_FC2))
    zu1 = synth.uniform(seed + 20, "zu1", (n_steps, B, dims["lstm_dims"]))
    zu2 = synth.uniform(seed + 20, "zu2", (n_steps, B, dims["lstm_dims"]))
    with torch.no_grad():
        for k in range(n_steps):
            tt = k * r
            pin = mel_in[:, :, tt - 1] if tt > 0 else torch.zeros(B, mel_in.shape[1])
            if mode in ("prenet_train", "train"):
                _noise_queue[:] = [t(du1[k]), t(du2[k])]
            if mode == "train":
                _zone_queue[:] = [t(zu1[k]), t(zu2[k])]
            mels, scores, hidden, cell, ctx = dec(enc_q, enc_p, pin, hidden, cell, ctx, tt)
            out[f"mels{k}"] = mels.numpy().copy()
            out[f"scores{k}"] = scores.numpy().copy()
            out[f"attn_h{k}"], out[f"h1_{k}"], out[f"h2_{k}"] = [h.numpy().copy() for h in hidden]
            out[f"c1_{k}"], out[f"c2_{k}"] = [c.numpy().copy() for c in cell]
            out[f"ctx{k}"] = ctx.numpy().copy()
            out[f"cum{k}"] = dec.attn_net.cumulative.numpy().copy()
            out[f"att{k}"] = dec.attn_net.attention.numpy().copy()
    recipe = dict(kind="steps", dims=dims, B=B, T=T, r=r, seed=seed, peaky=peaky, mode=mode,
                  n_steps=n_steps)
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe), **out)
    print("wrote", name)


def case_loop(name, dims, B, T, S, r, seed, peaky, mode="eval"):
    """The loop of Tacotron.forward (tacotron.py:339-348,356) around the reference Decoder."""
    dec = ref_decoder(dims, seed, peaky, r)
    if mode == "prenet_train":
        dec.prenet.train()
    enc_q, enc_p, mel_in = synth.decoder_inputs(B, T, S, dims["n_mels"], dims["decoder_dims"],
                                                dims["aligner_out_dims"], seed=seed + 10)
    enc_q, enc_p, mel_in = t(enc_q), t(enc_p), t(mel_in)
    hidden, cell, ctx = zero_state(dec, B, dims)
    n_steps = len(range(0, S, r))
    du1 = synth.uniform(seed + 20, "du1", (n_steps, B, synth.PRENET_FC1))
    du2 = synth.uniform(seed + 20, "du2", (n_steps, B, synth.PRENET_FC2))
    mel_outputs, attn_scores = [], []
    with torch.no_grad():
        for k, tt in enumerate(range(0, S, r)):
            pin = mel_in[:, :, tt - 1] if tt > 0 else torch.zeros(B, mel_in.shape[1])
            if mode == "prenet_train":
                _noise_queue[:] = [t(du1[k]), t(du2[k])]
            mels, scores, hidden, cell, ctx = dec(enc_q, enc_p, pin, hidden, cell, ctx, tt)
            mel_outputs.append(mels)
            attn_scores.append(scores)
    mel = torch.cat(mel_outputs, dim=2).numpy()
    att = torch.cat(attn_scores, dim=1).numpy()
    recipe = dict(kind="loop", dims=dims, B=B, T=T, S=S, r=r, seed=seed, peaky=peaky, mode=mode)
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe), mel=mel, att=att)
    print("wrote", name, mel.shape, att.shape, "att max", float(att.max()))


def case_free(name, dims, B, T, steps, r, seed, peaky, keep_every):
    """Free-running harness around the reference Decoder: feedback = last frame + zeros."""
    dec = ref_decoder(dims, seed, peaky, r)
    enc_q, enc_p, _ = synth.decoder_inputs(B, T, 1, dims["n_mels"], dims["decoder_dims"],
                                           dims["aligner_out_dims"], seed=seed + 10)
    enc_q, enc_p = t(enc_q), t(enc_p)
    hidden, cell, ctx = zero_state(dec, B, dims)
    pin_dim = dims["n_mels"] + dims["aligner_out_dims"]
    pin = torch.zeros(B, pin_dim)
    mel_outputs, attn_scores = [], []
    with torch.no_grad():
        for k in range(steps):
            mels, scores, hidden, cell, ctx = dec(enc_q, enc_p, pin, hidden, cell, ctx, k * r)
            mel_outputs.append(mels)
            attn_scores.append(scores)
            pin = torch.zeros(B, pin_dim)
            pin[:, :dims["n_mels"]] = mels[:, :, -1]
    mel = torch.cat(mel_outputs, dim=2).numpy()
    att = torch.cat(attn_scores, dim=1).numpy()
    recipe = dict(kind="free", dims=dims, B=B, T=T, steps=steps, r=r, seed=seed, peaky=peaky,
                  keep_every=keep_every)
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe),
             mel_sub=mel[:, :, ::keep_every].copy(), mel_head=mel[:, :, :32].copy(),
             mel_tail=mel[:, :, -32:].copy(),
             att_argmax=att.argmax(axis=2).astype(np.int16), att_max=att.max(axis=2),
             att_tail=att[:, -4:, :].copy())
    print("wrote", name, mel.shape, "att max", float(att.max()))


def case_lr():
    lr = LengthRegulator()
    out = {}
    # boundary durations incl. negative / .49 / .5 / 1.5 and a zero-length row
    x = synth.normalish(7, "lrx", (3, 5, 8))
    dur = np.array([[1.4, 1.5, -2.0, 0.49, 3.0],
                    [0.0, 0.0, 0.2, -1.0, 0.0],
                    [2.5, 0.5, 1.0, 4.49, 0.51]], dtype=np.float32)
    d = t(dur.copy())
    y = lr(t(x), d)
    out.update(x0=x, dur0=dur, y0=y.numpy(), dur0_after=d.numpy())
    # ragged, larger
    x1, dur1 = synth.lr_inputs(4, 19, 24, seed=5, dur_lo=0, dur_hi=7)
    dur1[2, :] = 0.0
    dur1[1, 3] = -3.0
    d1 = t(dur1.copy())
    y1 = lr(t(x1), d1)
    out.update(y1=y1.numpy(), dur1=dur1, dur1_after=d1.numpy())
    # all rows empty -> [B, 0, C]
    x2 = synth.normalish(8, "lrx2", (2, 3, 4))
    dur2 = np.zeros((2, 3), dtype=np.float32)
    y2 = lr(t(x2), t(dur2.copy()))
    out.update(x2=x2, dur2=dur2, y2_shape=np.array(y2.shape))
    recipe = dict(kind="lr", case1=dict(B=4, T=19, C=24, seed=5, dur_lo=0, dur_hi=7))
    np.savez(os.path.join(OUT, "lr_cases.npz"), recipe=json.dumps(recipe), **out)
    print("wrote lr_cases", y.shape, y1.shape, y2.shape)


TACO_CFG = dict(embed_dims=64, num_chars=40, encoder_dims=128, decoder_dims=256, n_mels=80, postnet_dims=64,
                encoder_k=4, lstm_dims=128, postnet_k=3, num_highways=2, dropout=0.5, stop_threshold=-11.0,
                aligner_hidden_dims=32, aligner_out_dims=8, speaker_emb_dim=0)


def case_tacotron(name, r, B, T, S, seed, speaker_emb_dim=0):
    """Whole reference Tacotron.forward (eval mode) on a small config: pins row f2 (the modules
    either side of the loop) together with the loop.  speaker_emb_dim > 0: the multispeaker wiring
    (tacotron.py:323-329: speaker embedding into the aligner's text encoder and the encoder memories)."""
    from models.tacotron import Tacotron as RefTacotron
    cfg = dict(TACO_CFG, speaker_emb_dim=speaker_emb_dim)
    ref = RefTacotron(**cfg)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    sd = synth.fill_state_dict(shapes, seed=seed, peaky=3.0)
    missing = ref.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.split(".")[-1] in ("r", "step", "stop_threshold", "num_batches_tracked")
                                               for k in missing.missing_keys), missing
    ref.eval()
    ref.r = r
    x = (synth.uniform01(seed, "tok", B * T) * TACO_CFG["num_chars"]).astype(np.int64).reshape(B, T)
    mel = synth.uniform(seed, "mel", (B, 80, S), -11.5129, 2.0)
    semb = synth.normalish(seed, "semb", (B, speaker_emb_dim)) if speaker_emb_dim else np.zeros((B, 0), np.float32)
    with torch.no_grad():
        out = ref({"x": t(x), "mel": t(mel), "speaker_emb": t(semb)})
    recipe = dict(kind="tacotron", cfg=cfg, r=r, B=B, T=T, S=S, seed=seed,
                  state_dict_shapes={k: list(v.shape) for k, v in ref.state_dict().items()})
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe),
             **{k: v.numpy() for k, v in out.items()})
    print("wrote", name, {k: tuple(v.shape) for k, v in out.items()}, "att max", float(out["att"].max()))


FWD_CFG = dict(embed_dims=32, series_embed_dims=16, num_chars=40, durpred_conv_dims=32, durpred_rnn_dims=16,
               durpred_dropout=0.5, pitch_conv_dims=32, pitch_rnn_dims=16, pitch_dropout=0.5, pitch_strength=1.0,
               energy_conv_dims=32, energy_rnn_dims=16, energy_dropout=0.5, energy_strength=1.0, rnn_dims=20,
               prenet_dims=16, prenet_k=4, postnet_num_highways=2, prenet_dropout=0.5, postnet_dims=32,
               postnet_k=4, prenet_num_highways=2, postnet_dropout=0.0, n_mels=80)


def case_forward_lstm(name, B, T, seed, rnn_dims=20, prenet_dims=None):
    """Row f1: the reference ForwardTacotron's own `self.lr -> pack -> self.lstm -> pad_packed -> self.lin`
    (forward(), packed path, forward_tacotron.py:145-154) and `self.lr -> self.lstm` (generate path, :222-224),
    captured with forward hooks while the reference module runs (eval mode)."""
    from models.forward_tacotron import ForwardTacotron as RefForward
    cfg = dict(FWD_CFG, rnn_dims=rnn_dims)
    if prenet_dims is not None:    # the LSTM's input width is 2 * prenet_dims (forward_tacotron.py:96)
        cfg["prenet_dims"] = prenet_dims
    ref = RefForward(**cfg)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    sd = synth.fill_state_dict(shapes, seed=seed)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    ref.eval()
    x = (synth.uniform01(seed, "tok", B * T) * cfg["num_chars"]).astype(np.int64).reshape(B, T)
    _, dur = synth.lr_inputs(B, T, 4, seed=seed, dur_lo=0, dur_hi=4)
    dur[1:, -2:] = 0.0                                   # ragged rows
    dur[-1, T // 2:] = 0.0
    mel_len = dur.sum(axis=1).astype(np.int64)
    S = int(mel_len.max())
    mel = synth.uniform(seed, "mel", (B, 80, S), -11.5129, 2.0)
    pitch = synth.normalish(seed, "pitch", (B, T))
    energy = synth.normalish(seed, "energy", (B, T))
    cap = {}
    hooks = [ref.lr.register_forward_hook(lambda m, i, o: cap.__setitem__("lr_out", o.detach().clone())),
             ref.lin.register_forward_hook(lambda m, i, o: cap.__setitem__("lin_in", i[0].detach().clone())),
             ref.lstm.register_forward_hook(lambda m, i, o: cap.__setitem__("lstm_state", [v.detach().clone() for v in o[1]]))]
    with torch.no_grad():
        out = ref({"x": t(x), "mel": t(mel), "dur": t(dur.copy()), "mel_len": t(mel_len),
                   "pitch": t(pitch), "energy": t(energy)})
        packed = dict(lr_out=cap["lr_out"].numpy(), lstm_out=cap["lin_in"].numpy(),
                      h_n=cap["lstm_state"][0].numpy(), c_n=cap["lstm_state"][1].numpy(), mel=out["mel"].numpy())
        gen = ref._generate_mel(x=t(x), dur_hat=t(dur.copy()), pitch_hat=t(pitch).unsqueeze(1),
                                energy_hat=t(energy).unsqueeze(1))
        padded = dict(lstm_out=cap["lin_in"].numpy(), h_n=cap["lstm_state"][0].numpy(),
                      c_n=cap["lstm_state"][1].numpy(), mel=gen["mel"].numpy())
    for h in hooks:
        h.remove()
    recipe = dict(kind="forward_lstm", cfg=cfg, B=B, T=T, seed=seed, padding_value=-11.5129,
                  lstm_shapes={k[len("lstm."):]: list(v) for k, v in shapes.items() if k.startswith("lstm.")})
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe), mel_len=mel_len,
             lr_out=packed["lr_out"], packed_lstm_out=packed["lstm_out"], packed_h_n=packed["h_n"],
             packed_c_n=packed["c_n"], padded_lstm_out=padded["lstm_out"], padded_h_n=padded["h_n"],
             padded_c_n=padded["c_n"])
    print("wrote", name, packed["lr_out"].shape, packed["lstm_out"].shape, "mel_len", mel_len.tolist())


def case_forward_tacotron(name, B, T, seed, dur_bias):
    """Whole reference ForwardTacotron (eval): forward(batch) with ragged lengths and generate(x).
    dur_bias = None keeps the random-init duration predictor (all durations < 1, so generate takes the
    `fill 2.0` fallback, forward_tacotron.py:176-177); a number overrides dur_pred.lin.bias."""
    from models.forward_tacotron import ForwardTacotron as RefForward
    ref = RefForward(**FWD_CFG)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    sd = synth.fill_state_dict(shapes, seed=seed)
    if dur_bias is not None:
        sd["dur_pred.lin.bias"] = np.full((1,), dur_bias, dtype=np.float32)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    ref.eval()
    x = (synth.uniform01(seed, "tok", B * T) * FWD_CFG["num_chars"]).astype(np.int64).reshape(B, T)
    _, dur = synth.lr_inputs(B, T, 4, seed=seed, dur_lo=0, dur_hi=4)
    dur[1:, -3:] = 0.0
    mel_len = dur.sum(axis=1).astype(np.int64)
    S = int(mel_len.max()) + 2                     # exercises _pad
    mel = synth.uniform(seed, "mel", (B, 80, S), -11.5129, 2.0)
    pitch = synth.normalish(seed, "pitch", (B, T))
    energy = synth.normalish(seed, "energy", (B, T))
    with torch.no_grad():
        fwd = ref({"x": t(x), "mel": t(mel), "dur": t(dur.copy()), "mel_len": t(mel_len),
                   "pitch": t(pitch), "energy": t(energy)})
        gen = ref.generate(t(x[:1]), alpha=0.9, pitch_function=lambda p: p * 1.5)
    recipe = dict(kind="forward_tacotron", cfg=FWD_CFG, B=B, T=T, S=S, seed=seed, dur_bias=dur_bias,
                  state_dict_shapes={k: list(v.shape) for k, v in ref.state_dict().items()})
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe),
             **{"fwd_" + k: v.numpy() for k, v in fwd.items()}, **{"gen_" + k: v.numpy() for k, v in gen.items()})
    print("wrote", name, {k: tuple(v.shape) for k, v in fwd.items()}, "gen mel", tuple(gen["mel"].shape),
          "gen dur sum", float(gen["dur"].sum()))


def case_durations():
    """Row f3: the reference DurationExtractor (scipy Dijkstra over its Python-built graph) on seeded
    attention matrices.  'unique' cases have continuous attention and no probability shift, so the
    shortest path is unique; 'tie' cases (one-hot / clamped rows) have many equal-cost paths."""
    from duration_extraction.duration_extractor import DurationExtractor as RefExtractor
    out, meta = {}, []
    n = 0
    for kind, S, T, thr, shift in [("unique", 23, 7, -10.5, 0.0), ("unique", 60, 25, -10.5, 0.0),
                                   ("unique", 9, 9, -11.0, 0.0), ("unique", 12, 1, -10.5, 0.0),
                                   ("unique", 5, 8, -10.5, 0.0), ("unique", 40, 13, -20.0, 0.0),
                                   ("shift", 30, 10, -10.5, 0.25), ("shift", 45, 17, -10.5, 0.1),
                                   ("tie", 20, 6, -11.0, 0.0), ("tie", 31, 12, -11.0, 0.25),
                                   ("onesil", 16, 5, -10.5, 0.3)]:
        seed = 700 + n
        logits = synth.normalish(seed, "att", (S, T)).astype(np.float64) * 3.0
        att = torch.softmax(torch.as_tensor(logits), dim=1).float()
        mel = torch.as_tensor(synth.uniform(seed, "mel", (80, S), -12.0, -9.0))
        if kind == "tie":
            cols = np.sort((synth.uniform01(seed, "cols", S) * T).astype(np.int64))
            att = torch.zeros(S, T)
            att[torch.arange(S), torch.as_tensor(cols)] = 1.0
            att[S // 3: S // 3 + 3] = 1.0 / T
            mel = torch.full((80, S), -10.0)
            mel[:, S // 3: S // 3 + 2] = -11.6
        if kind == "onesil":
            mel = torch.full((80, S), -10.0)
            mel[:, 4] = -11.6                               # exactly one silent frame: no shift is applied (:43-44)
        x = torch.as_tensor((synth.uniform01(seed, "tok", T) * 30).astype(np.int64))
        dur, score = RefExtractor(thr, shift)(x=x, mel=mel, attention=att.clone())
        out.update({f"c{n}_x": x.numpy(), f"c{n}_mel": mel.numpy(), f"c{n}_att": att.numpy(),
                    f"c{n}_dur": dur.numpy(), f"c{n}_score": np.float64(score)})
        meta.append(dict(kind=kind, S=S, T=T, thr=thr, shift=shift))
        n += 1
    np.savez_compressed(os.path.join(OUT, "durations_cases.npz"), recipe=json.dumps(dict(kind="durations", cases=meta)), **out)
    print("wrote durations_cases", n)


FULL = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
SMALL = dict(n_mels=8, decoder_dims=256, lstm_dims=64, aligner_out_dims=4)

FP_CFG = dict(num_chars=40, durpred_dropout=0.1, durpred_d_model=16, durpred_n_heads=2, durpred_layers=1, durpred_d_fft=32,
              pitch_dropout=0.1, pitch_d_model=16, pitch_n_heads=2, pitch_layers=1, pitch_d_fft=32,
              energy_dropout=0.1, energy_d_model=16, energy_n_heads=2, energy_layers=1, energy_d_fft=32,
              pitch_strength=1.0, energy_strength=1.0, d_model=24, conv1_kernel=9, conv2_kernel=1,
              prenet_layers=2, prenet_heads=2, prenet_fft=48, prenet_dropout=0.1,
              postnet_layers=2, postnet_heads=2, postnet_fft=48, postnet_dropout=0.1, n_mels=80)
MFP_CFG = dict(FP_CFG, pitch_cond_d_model=16, pitch_cond_n_heads=2, pitch_cond_layers=1, pitch_cond_d_fft=32,
               pitch_cond_dropout=0.1, pitch_cond_output_dims=4, speaker_emb_dims=8)
MFT_CFG = dict(FWD_CFG, pitch_cond_conv_dims=32, pitch_cond_rnn_dims=16, pitch_cond_dropout=0.5, speaker_emb_dims=256,
               pitch_cond_emb_dims=8, pitch_cond_categorical_dims=4)


def case_forward_model(name, kind, cfg, B, T, seed, dur_bias):
    """the reference FastPitch / MultiFastPitch / MultiForwardTacotron (eval): forward(batch) with ragged
    lengths and generate(x[, speaker_emb]) on the first row.  The sinusoidal position tables keep their own values."""
    if kind == "fast_pitch":
        from models.fast_pitch import FastPitch as Ref
    elif kind == "multi_fast_pitch":
        from models.multi_fast_pitch import MultiFastPitch as Ref
    else:
        from models.multi_forward_tacotron import MultiForwardTacotron as Ref
    multi = kind != "fast_pitch"
    ref = Ref(**cfg)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()
              if v.is_floating_point() and v.dim() > 0 and not k.endswith(".pe")}
    sd = synth.fill_state_dict(shapes, seed=seed)
    if dur_bias is not None:
        sd["dur_pred.lin.bias"] = np.full((1,), dur_bias, dtype=np.float32)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    ref.eval()
    batch = synth.forward_model_inputs(cfg, B, T, seed, multi)
    with torch.no_grad():
        fwd = ref({k: t(v.copy()) for k, v in batch.items()})
        args = (t(batch["x"][:1]),) + ((t(batch["speaker_emb"][:1]),) if multi else ())
        gen = ref.generate(*args, alpha=0.9, pitch_function=lambda p: p * 1.5)
    recipe = dict(kind=kind, cfg=cfg, B=B, T=T, seed=seed, dur_bias=dur_bias,
                  state_dict_shapes={k: list(v.shape) for k, v in ref.state_dict().items()})
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe),
             **{"fwd_" + k: v.numpy() for k, v in fwd.items()}, **{"gen_" + k: v.numpy() for k, v in gen.items()})
    print("wrote", name, {k: tuple(v.shape) for k, v in fwd.items()}, "gen mel", tuple(gen["mel"].shape),
          "gen dur sum", float(gen["dur"].sum()))


FWDMODEL_CASES = [("fwdmodel_fastpitch_fallback", "fast_pitch", FP_CFG, 3, 11, 530, None),
                  ("fwdmodel_fastpitch_dur3", "fast_pitch", FP_CFG, 4, 13, 531, 3.0),
                  ("fwdmodel_multifastpitch_dur3", "multi_fast_pitch", MFP_CFG, 3, 12, 532, 3.0),
                  ("fwdmodel_multifwdtaco_dur3", "multi_forward_tacotron", MFT_CFG, 3, 10, 533, 3.0)]


def case_cbhg_gru(name, K, in_channels, channels, proj, highways, B, M, seed):
    """Row f2: the bidirectional GRU that ends the reference's CBHG block (common_layers.py:89,123), captured with a
    forward hook on `rnn` while the reference CBHG runs in eval mode: its input (the highway output), its output
    (= the CBHG output) and its final state."""
    from models.common_layers import CBHG as RefCBHG
    ref = RefCBHG(K=K, in_channels=in_channels, channels=channels, proj_channels=proj, num_highways=highways)
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    sd = synth.fill_state_dict(shapes, seed=seed)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    ref.eval()
    x = synth.normalish(seed, "cbhg_x", (B, in_channels, M))
    cap = {}
    hook = ref.rnn.register_forward_hook(lambda m, i, o: cap.update(rnn_in=i[0].detach().clone(), h_n=o[1].detach().clone()))
    with torch.no_grad():
        out = ref(t(x))
    hook.remove()
    recipe = dict(kind="cbhg_gru", K=K, in_channels=in_channels, channels=channels, proj=proj, highways=highways,
                  B=B, M=M, seed=seed, rnn_shapes={k[len("rnn."):]: list(v) for k, v in shapes.items() if k.startswith("rnn.")})
    np.savez(os.path.join(OUT, name + ".npz"), recipe=json.dumps(recipe), rnn_in=cap["rnn_in"].numpy(),
             rnn_out=out.numpy(), h_n=cap["h_n"].numpy())
    print("wrote", name, cap["rnn_in"].shape, out.shape)


def case_mol_sampler():
    """The reference's mixture-of-logistics sampler (utils/distribution.py:87-123), the only piece of the WaveRNN
    path the mounted tree still holds.  Its two internal `uniform_(1e-5, 1 - 1e-5)` draws are made reproducible by
    swapping Tensor.uniform_ for a version that hands out caller-supplied uniforms (already inside that range) while
    the reference function runs; F.one_hot is wrapped to record the mixture index the function picked."""
    from utils import distribution as D
    B, nr_mix, T = 4, 10, 96
    y = synth.normalish(600, "mol_y", (B, 3 * nr_mix, T)).astype(np.float32)
    y[:, nr_mix:2 * nr_mix] *= 0.4                         # means inside [-1, 1] mostly
    y[:, 2 * nr_mix:] = y[:, 2 * nr_mix:] * 1.5 - 3.0      # log-scales around e^-3
    y[0, 2 * nr_mix:, :8] = -40.0                          # below log_scale_min = ln 1e-14: the clamp acts
    y[1, nr_mix:2 * nr_mix, :8] = 3.0                      # means outside [-1, 1]: the output clamp acts
    lo, hi = 1e-5, 1.0 - 1e-5
    u_mix = (lo + (hi - lo) * synth.uniform01(601, "mol_um", B * T * nr_mix).astype(np.float64)).astype(np.float32).reshape(B, T, nr_mix)
    u_log = (lo + (hi - lo) * synth.uniform01(602, "mol_ul", B * T).astype(np.float64)).astype(np.float32).reshape(B, T)
    queue = [t(u_mix), t(u_log)]
    picked = []
    orig_uniform, orig_one_hot = torch.Tensor.uniform_, D.F.one_hot

    def queued_uniform(self, a=0.0, b=1.0):
        u = queue.pop(0)
        assert (a, b) == (1e-5, 1.0 - 1e-5) and u.shape == self.shape, (a, b, u.shape, self.shape)
        return self.copy_(u)

    def recording_one_hot(idx, n):
        picked.append(idx.clone())
        return orig_one_hot(idx, n)

    torch.Tensor.uniform_, D.F.one_hot = queued_uniform, recording_one_hot
    try:
        x = D.sample_from_discretized_mix_logistic(t(y))
        x_min7 = None
        queue.extend([t(u_mix), t(u_log)])
        x_min7 = D.sample_from_discretized_mix_logistic(t(y), log_scale_min=-7.0)
    finally:
        torch.Tensor.uniform_, D.F.one_hot = orig_uniform, orig_one_hot
    assert not queue and len(picked) == 2 and torch.equal(picked[0], picked[1])
    recipe = dict(kind="mol", B=B, nr_mix=nr_mix, T=T, seeds=[600, 601, 602])
    np.savez(os.path.join(OUT, "mol_sampler.npz"), recipe=json.dumps(recipe), y=y, u_mix=u_mix, u_log=u_log,
             x=x.numpy(), x_min7=x_min7.numpy(), mix=picked[0].numpy().astype(np.int32))
    print("wrote mol_sampler", x.shape, "clamped", int((x.abs() == 1).sum()), "mix histogram", np.bincount(picked[0].numpy().ravel(), minlength=nr_mix))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1:   # `make_golden.py cbhggru` regenerates only the fixtures whose name starts with the prefix
        if sys.argv[1] == "tacotron_spk":
            case_tacotron("tacotron_spk_r2", r=2, B=3, T=10, S=18, seed=402, speaker_emb_dim=16)
            sys.exit(0)
        if sys.argv[1] == "mol":
            case_mol_sampler()
            sys.exit(0)
        if sys.argv[1] == "cbhggru":
            case_cbhg_gru("cbhggru_postnet", K=8, in_channels=80, channels=128, proj=[256, 80], highways=4, B=3, M=37, seed=520)
            case_cbhg_gru("cbhggru_small", K=4, in_channels=16, channels=20, proj=[32, 16], highways=2, B=5, M=9, seed=521)
            sys.exit(0)
        if sys.argv[1] == "fwdlstm_h512":
            # the reference's own sizes (in = 2 * prenet_dims = 512, rnn_dims = 512: configs/singlespeaker.yaml), which is what
            # the persistent weights-in-registers kernel runs
            case_forward_lstm("fwdlstm_h512", B=2, T=6, seed=502, rnn_dims=512, prenet_dims=256)
            sys.exit(0)
        if sys.argv[1] == "fwdmodel":
            for c in FWDMODEL_CASES:
                case_forward_model(*c)
            sys.exit(0)
        sys.exit("unknown fixture group " + sys.argv[1])
    for r in (1, 2, 5):
        case_steps(f"steps_full_r{r}_eval", FULL, B=3, T=17, r=r, seed=100 + r, peaky=8.0, mode="eval")
    case_steps("steps_full_r1_prenet_train", FULL, B=3, T=17, r=1, seed=111, peaky=8.0, mode="prenet_train")
    case_steps("steps_full_r2_train", FULL, B=3, T=17, r=2, seed=112, peaky=8.0, mode="train")
    case_steps("steps_small_r3_eval", SMALL, B=5, T=9, r=3, seed=113, peaky=4.0, mode="eval")
    for r in (1, 2, 5):
        case_loop(f"loop_full_r{r}", FULL, B=2, T=24, S=40, r=r, seed=200 + r, peaky=3.0)
    case_loop("loop_full_r1_flat", FULL, B=2, T=40, S=24, r=1, seed=210, peaky=1.0)
    case_loop("loop_full_r1_drop", FULL, B=2, T=24, S=20, r=1, seed=211, peaky=3.0, mode="prenet_train")
    case_loop("loop_full_r3_ragged", FULL, B=3, T=33, S=20, r=3, seed=212, peaky=3.0)  # S % r != 0
    case_loop("loop_small_r2", SMALL, B=4, T=13, S=16, r=2, seed=213, peaky=4.0)
    case_free("free_full_r1", FULL, B=1, T=20, steps=1000, r=1, seed=300, peaky=3.0, keep_every=8)
    case_free("free_full_r2", FULL, B=2, T=16, steps=100, r=2, seed=301, peaky=3.0, keep_every=4)
    case_lr()
    case_tacotron("tacotron_r1", r=1, B=2, T=11, S=24, seed=400)
    case_tacotron("tacotron_r2", r=2, B=3, T=9, S=20, seed=401)
    case_tacotron("tacotron_spk_r2", r=2, B=3, T=10, S=18, seed=402, speaker_emb_dim=16)
    case_forward_lstm("fwdlstm_h20", B=4, T=12, seed=500, rnn_dims=20)
    case_forward_lstm("fwdlstm_h64", B=3, T=20, seed=501, rnn_dims=64)
    case_forward_lstm("fwdlstm_h512", B=2, T=6, seed=502, rnn_dims=512, prenet_dims=256)
    case_durations()
    case_forward_tacotron("fwdtaco_fallback", B=3, T=10, seed=510, dur_bias=None)
    case_forward_tacotron("fwdtaco_dur3", B=4, T=14, seed=511, dur_bias=3.0)
    case_cbhg_gru("cbhggru_postnet", K=8, in_channels=80, channels=128, proj=[256, 80], highways=4, B=3, M=37, seed=520)
    case_cbhg_gru("cbhggru_small", K=4, in_channels=16, channels=20, proj=[32, 16], highways=2, B=5, M=9, seed=521)
    for c in FWDMODEL_CASES:
        case_forward_model(*c)
    case_mol_sampler()
