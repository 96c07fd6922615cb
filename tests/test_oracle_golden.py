"""Pin the CPU oracle (oracle/taco_oracle.py) to the reference-generated goldens.

The fixtures under tests/golden/ were produced by tools/make_golden.py running the
reference's own Decoder / LengthRegulator; the reference holds no test vector for this
path.  fp32 tolerance: 1e-5 relative for single steps and short
loops (the reference itself moves 1.8e-7 between 1 and 8 threads), 1e-4 for 1000
free-running steps.
"""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import taco_oracle as O

torch.set_num_threads(4)


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a))


@pytest.mark.parametrize("name", G.names("steps_"))
def test_single_steps(name):
    rec, gold = G.load(name)
    W = O.to_torch(G.weights(rec))
    enc_q, enc_p, mel_in = map(T, G.inputs(rec))
    nz = G.noise(rec, rec["n_steps"])
    B, r = rec["B"], rec["r"]
    hidden, cell, ctx = O.zero_state(W, B)
    lsa = None
    for k in range(rec["n_steps"]):
        t = k * r
        pin = mel_in[:, :, t - 1] if t > 0 else torch.zeros(B, mel_in.shape[1])
        du = (T(nz["du1"][k]), T(nz["du2"][k])) if rec["mode"] in ("prenet_train", "train") else None
        zu = (T(nz["zu1"][k]), T(nz["zu2"][k])) if rec["mode"] == "train" else None
        mels, scores, hidden, cell, ctx, lsa = O.decoder_step(W, enc_q, enc_p, pin, hidden, cell, ctx,
                                                              lsa, t, r, du, zu)
        got = dict(mels=mels, scores=scores, attn_h=hidden[0], h1_=hidden[1], h2_=hidden[2],
                   c1_=cell[0], c2_=cell[1], ctx=ctx, cum=lsa[0], att=lsa[1])
        for key, val in got.items():
            ref = gold[f"{key}{k}"]
            assert val.shape == ref.shape, (key, k)
            assert G.rel_err(val.numpy(), ref) < 1e-5, (key, k, G.rel_err(val.numpy(), ref))


@pytest.mark.parametrize("name", G.names("loop_"))
def test_teacher_forced_loop(name):
    rec, gold = G.load(name)
    W = O.to_torch(G.weights(rec))
    enc_q, enc_p, mel_in = map(T, G.inputs(rec))
    n_steps = len(range(0, rec["S"], rec["r"]))
    du = None
    if rec["mode"] == "prenet_train":
        nz = G.noise(rec, n_steps)
        du = (T(nz["du1"]), T(nz["du2"]))
    mel, att = O.run_teacher_forced(W, enc_q, enc_p, mel_in, rec["r"], du)
    assert mel.shape == gold["mel"].shape and att.shape == gold["att"].shape
    assert G.rel_err(mel.numpy(), gold["mel"]) < 2e-5
    assert G.rel_err(att.numpy(), gold["att"]) < 2e-5


@pytest.mark.parametrize("name", G.names("free_"))
def test_free_running(name):
    rec, gold = G.load(name)
    W = O.to_torch(G.weights(rec))
    enc_q, enc_p, _ = map(T, G.inputs(rec))
    mel, att, n = O.run_free(W, enc_q, enc_p, rec["steps"], rec["r"])
    assert n == rec["steps"]
    mel, att = mel.numpy(), att.numpy()
    assert G.rel_err(mel[:, :, ::rec["keep_every"]], gold["mel_sub"]) < 1e-4
    assert G.rel_err(mel[:, :, :32], gold["mel_head"]) < 1e-5
    assert G.rel_err(mel[:, :, -32:], gold["mel_tail"]) < 1e-4
    assert (att.argmax(axis=2) == gold["att_argmax"]).all()
    assert G.rel_err(att[:, -4:, :], gold["att_tail"]) < 1e-4


def test_free_running_stop_rule():
    # stop rule is build-defined: with a threshold above every output the
    # loop ends after group index 11 (k > 10), with none it runs max_steps.
    rec, _ = G.load("free_full_r2")
    W = O.to_torch(G.weights(rec))
    enc_q, enc_p, _ = map(T, G.inputs(rec))
    _, _, n = O.run_free(W, enc_q, enc_p, 40, rec["r"], stop_threshold=1e9)
    assert n == 12
    _, _, n = O.run_free(W, enc_q, enc_p, 15, rec["r"], stop_threshold=-1e9)
    assert n == 15


def test_length_regulator():
    rec, g = G.load("lr_cases")
    d0 = T(g["dur0"].copy())
    y0 = O.length_regulate(T(g["x0"]), d0)
    assert np.array_equal(y0.numpy(), g["y0"])
    assert np.array_equal(d0.numpy(), g["dur0_after"])       # in-place clamp of negatives
    c = rec["case1"]
    from wavernn_amd import synth
    x1, _ = synth.lr_inputs(c["B"], c["T"], c["C"], seed=c["seed"], dur_lo=c["dur_lo"], dur_hi=c["dur_hi"])
    d1 = T(g["dur1"].copy())
    y1 = O.length_regulate(T(x1), d1)
    assert np.array_equal(y1.numpy(), g["y1"])
    assert np.array_equal(d1.numpy(), g["dur1_after"])
    y2 = O.length_regulate(T(g["x2"]), T(g["dur2"].copy()))
    assert list(y2.shape) == list(g["y2_shape"])


def test_mol_sampler_oracle_matches_reference_golden():
    """oracle.wavernn_oracle.sample_mol against the reference's own sample_from_discretized_mix_logistic
    (utils/distribution.py:87-123) run with the same uniforms: same CPU, same torch ops, so bit for bit."""
    import torch
    from oracle import wavernn_oracle as WO
    rec, g = G.load("mol_sampler")
    y, um, ul = (torch.as_tensor(g[k]) for k in ("y", "u_mix", "u_log"))
    x, mix = WO.sample_mol(y, um, ul)
    assert np.array_equal(mix.numpy(), g["mix"]) and np.array_equal(x.numpy(), g["x"])
    x7, _ = WO.sample_mol(y, um, ul, log_scale_min=-7.0)
    assert np.array_equal(x7.numpy(), g["x_min7"]) and not np.array_equal(g["x"], g["x_min7"])
    assert (np.abs(g["x"]) <= 1.0).all() and (np.abs(g["x"]) == 1.0).any()      # the output clamp acted
