"""GPU parity of the HIP decoder path (through the C ABI) against the reference goldens and
the CPU oracle.  Tolerance: 1e-4 relative (max|a-b| / max|b|), the bar BASELINE.json's
north_star sets for fp32 teacher-forced steps; free-running additionally needs an identical
attention argmax trajectory."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import taco_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-4


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a))


def make_decoder(rec_or_sd, dims, r):
    from wavernn_amd import Decoder
    sd = rec_or_sd
    dec = Decoder(**dims)
    dec.load_state_dict({k: T(v) for k, v in sd.items()})
    dec.r = dec.r.new_tensor(r)
    return dec.cuda().eval()


@pytest.mark.parametrize("name", G.names("steps_"))
def test_single_steps_vs_reference_golden(name):
    rec, gold = G.load(name)
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, mel_in = (T(a).cuda() for a in G.inputs(rec))
    nz = {k: T(v).cuda() for k, v in G.noise(rec, rec["n_steps"]).items()}
    B, r = rec["B"], rec["r"]
    d, l = rec["dims"]["decoder_dims"], rec["dims"]["lstm_dims"]
    z = lambda n: torch.zeros(B, n, device="cuda")
    hidden, cell, ctx = (z(d), z(l), z(l)), (z(l), z(l)), z(d)
    if rec["mode"] == "prenet_train":
        dec.prenet.train()
    elif rec["mode"] == "train":
        dec.train()
    for k in range(rec["n_steps"]):
        t = k * r
        pin = mel_in[:, :, t - 1] if t > 0 else torch.zeros(B, mel_in.shape[1], device="cuda")
        noise = {}
        if rec["mode"] in ("prenet_train", "train"):
            noise.update(drop_u1=nz["du1"][k], drop_u2=nz["du2"][k])
        if rec["mode"] == "train":
            noise.update(zone_u1=nz["zu1"][k], zone_u2=nz["zu2"][k])
        mels, scores, hidden, cell, ctx = dec(enc_q, enc_p, pin, hidden, cell, ctx, t, noise=noise)
        got = dict(mels=mels, scores=scores, attn_h=hidden[0], h1_=hidden[1], h2_=hidden[2], c1_=cell[0],
                   c2_=cell[1], ctx=ctx, cum=dec.attn_net.cumulative, att=dec.attn_net.attention)
        for key, val in got.items():
            ref = gold[f"{key}{k}"]
            assert tuple(val.shape) == ref.shape, (key, k)
            err = G.rel_err(val.cpu().numpy(), ref)
            assert err < TOL, (key, k, err)


@pytest.mark.parametrize("name", G.names("loop_"))
def test_teacher_forced_loop_vs_reference_golden(name):
    rec, gold = G.load(name)
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, mel_in = (T(a).cuda() for a in G.inputs(rec))
    noise = None
    if rec["mode"] == "prenet_train":
        dec.prenet.train()
        nz = G.noise(rec, len(range(0, rec["S"], rec["r"])))
        noise = dict(drop_u1=T(nz["du1"]).cuda(), drop_u2=T(nz["du2"]).cuda())
    mel, att = dec.decode_teacher_forced(enc_q, enc_p, mel_in, noise=noise)
    assert tuple(mel.shape) == gold["mel"].shape and tuple(att.shape) == gold["att"].shape
    assert G.rel_err(mel.cpu().numpy(), gold["mel"]) < TOL
    assert G.rel_err(att.cpu().numpy(), gold["att"]) < TOL


@pytest.mark.parametrize("name", G.names("loop_"))
def test_step_api_loop_matches_run_api(name):
    """Driving Decoder.forward step by step exactly like Tacotron.forward (tacotron.py:339-345)
    gives what the one-call loop gives."""
    rec, gold = G.load(name)
    if rec["mode"] != "eval":
        pytest.skip("noise layout differs between the two entry points")
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, mel_in = (T(a).cuda() for a in G.inputs(rec))
    B, r = rec["B"], rec["r"]
    d, l = rec["dims"]["decoder_dims"], rec["dims"]["lstm_dims"]
    z = lambda n: torch.zeros(B, n, device="cuda")
    hidden, cell, ctx = (z(d), z(l), z(l)), (z(l), z(l)), z(d)
    go = torch.zeros(B, mel_in.shape[1], device="cuda")
    mels, atts = [], []
    for t in range(0, rec["S"], r):
        pin = mel_in[:, :, t - 1] if t > 0 else go
        m, s, hidden, cell, ctx = dec(enc_q, enc_p, pin, hidden, cell, ctx, t)
        mels.append(m)
        atts.append(s)
    mel = torch.cat(mels, dim=2)
    att = torch.cat(atts, dim=1)
    mel2, att2 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert G.rel_err(mel.cpu().numpy(), gold["mel"]) < TOL
    assert torch.equal(mel, mel2) and torch.equal(att, att2)   # same kernels, same order: bit-identical


@pytest.mark.parametrize("name", G.names("free_"))
def test_free_running_vs_reference_golden(name):
    rec, gold = G.load(name)
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, _ = (T(a).cuda() for a in G.inputs(rec))
    mel, att, n = dec.decode_free(enc_q, enc_p, rec["steps"])
    assert n == rec["steps"]
    mel, att = mel.cpu().numpy(), att.cpu().numpy()
    assert G.rel_err(mel[:, :, ::rec["keep_every"]], gold["mel_sub"]) < TOL
    assert G.rel_err(mel[:, :, :32], gold["mel_head"]) < TOL
    assert G.rel_err(mel[:, :, -32:], gold["mel_tail"]) < TOL
    assert (att.argmax(axis=2) == gold["att_argmax"]).all()
    assert G.rel_err(att[:, -4:, :], gold["att_tail"]) < TOL


def test_free_running_stop_rule_and_aligner_feats():
    from wavernn_amd import synth
    rec, _ = G.load("free_full_r2")
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, _ = (T(a) for a in G.inputs(rec))
    W = O.to_torch(G.weights(rec))
    steps = 40
    ali = T(synth.normalish(77, "ali", (rec["B"], rec["dims"]["aligner_out_dims"], steps)))
    m_ref, a_ref, n_ref = O.run_free(W, enc_q, enc_p, steps, rec["r"], aligner_feats=ali, stop_threshold=1e9)
    mel, att, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, aligner_feats=ali.cuda(), stop_threshold=1e9)
    assert n == n_ref == 12
    assert G.rel_err(mel.cpu().numpy(), m_ref.numpy()) < TOL
    assert G.rel_err(att.cpu().numpy(), a_ref.numpy()) < TOL
    _, _, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), 15, stop_threshold=-1e9)
    assert n == 15


@pytest.mark.parametrize("B,Tn,S,r", [(64, 128, 24, 1), (7, 37, 12, 3), (1, 1, 5, 1), (33, 130, 8, 2),
                                         (2, 4096, 3, 1), (3, 2100, 4, 2)])   # the maximum T (128 chunks) and 66 chunks
def test_teacher_forced_vs_oracle_seeded(B, Tn, S, r):
    """Benchmark-shaped and ragged shapes against the oracle on the same seeded inputs."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0, r=r)
    dec = make_decoder(sd, dims, r)
    enc_q, enc_p, mel_in = (T(a) for a in synth.decoder_inputs(B, Tn, S, seed=31))
    m_ref, a_ref = O.run_teacher_forced(O.to_torch(sd), enc_q, enc_p, mel_in, r)
    mel, att = dec.decode_teacher_forced(enc_q.cuda(), enc_p.cuda(), mel_in.cuda())
    assert G.rel_err(mel.cpu().numpy(), m_ref.numpy()) < TOL
    assert G.rel_err(att.cpu().numpy(), a_ref.numpy()) < TOL
    # softmax rows sum to one, attention is a distribution over ALL T positions (no mask)
    assert torch.allclose(att.sum(dim=2), torch.ones_like(att.sum(dim=2)), atol=1e-5)


def test_full_size_properties():
    """BASELINE config 3 at full size (B=64, T=128, S=640): size-independent properties --
    batch rows are independent (a permuted batch gives the permuted output bit for bit,
    a sub-batch gives the same rows) and attention rows are distributions."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=0, peaky=3.0)
    dec = make_decoder(sd, dims, 1)
    B, Tn, S = 64, 128, 640
    enc_q, enc_p, mel_in = (T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=1))
    mel, att = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.isfinite(mel).all() and torch.isfinite(att).all()
    assert torch.allclose(att.sum(dim=2), torch.ones(B, S, device="cuda"), atol=2e-5)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    mel_p, att_p = dec.decode_teacher_forced(enc_q[perm].contiguous(), enc_p[perm].contiguous(),
                                             mel_in[perm].contiguous())
    assert torch.equal(mel_p, mel[perm]) and torch.equal(att_p, att[perm])
    # first 48 steps of 5 rows against the oracle
    sub = [0, 13, 31, 32, 63]
    m_ref, a_ref = O.run_teacher_forced(O.to_torch(sd), enc_q[sub].cpu(), enc_p[sub].cpu(),
                                        mel_in[sub][:, :, :48].cpu(), 1)
    assert G.rel_err(mel[sub][:, :, :48].cpu().numpy(), m_ref.numpy()) < TOL
    assert G.rel_err(att[sub][:, :48].cpu().numpy(), a_ref.numpy()) < TOL


def test_bad_arguments_raise():
    from wavernn_amd import Decoder
    with pytest.raises(RuntimeError, match="decoder_dims must be 256"):
        d = Decoder(80, 128, 512, 32).cuda()
        d._get_handle()
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    from wavernn_amd import synth
    dec = make_decoder(synth.decoder_state_dict(**dims), dims, 1)
    z = lambda *s: torch.zeros(*s, device="cuda")
    with pytest.raises(RuntimeError, match="before t == 0"):
        dec(z(2, 5, 256), z(2, 5, 256), z(2, 112), (z(2, 256), z(2, 512), z(2, 512)), (z(2, 512), z(2, 512)),
            z(2, 256), 3)
    with pytest.raises(TypeError):
        dec.decode_teacher_forced(z(2, 5, 256).double(), z(2, 5, 256), z(2, 112, 4))
    with pytest.raises(RuntimeError, match="encoder length"):
        dec.decode_teacher_forced(z(1, 4097, 256), z(1, 4097, 256), z(1, 112, 2))
