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
SOFTMAX_KEYS = ("scores", "cum", "att")   # element-wise |a-b| <= 1e-4 |b| + 1e-6
SUM_ATOL = 1e-5                           # signed sums: atol = 1e-5 * max|b| (golden_util.assert_close)
# Passes of hundreds of recurrent steps: two fp32 evaluations that differ in summation order (chain path vs launch path vs
# the CPU oracle) drift apart step by step (the reference itself moves 2e-6 from its fp64 evaluation in 1000 steps, SURVEY
# 8c), so the element-wise floor for mel frames there is the north star's own bar, 1e-4 of the tensor's scale; the
# max-norm bar (1e-4 relative) is unchanged and attention rows keep atol = 1e-6.
LONG_SUM_ATOL = 1e-4


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
            G.assert_close(val.cpu().numpy(), ref, (key, k), atol_scale=0.0 if key in SOFTMAX_KEYS else SUM_ATOL)


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
    G.assert_close(mel.cpu().numpy(), gold["mel"], "mel", atol_scale=SUM_ATOL)
    G.assert_close(att.cpu().numpy(), gold["att"], "att")


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
    dec.set_tf_path("launches")
    mel2, att2 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    G.assert_close(mel.cpu().numpy(), gold["mel"], "mel", atol_scale=SUM_ATOL)
    assert torch.equal(mel, mel2) and torch.equal(att, att2)   # same kernels, same order: bit-identical


@pytest.mark.parametrize("B,T,n", [(2, 24, 4), (3, 33, 5), (64, 128, 6), (2, 200, 3), (2, 300, 3)])
def test_deferred_combine_is_bit_identical(B, T, n, monkeypatch):
    """WRNN_DEFERRED_COMBINE=1 (the attention launch publishes its chunk records, the next step's GRU launch combines
    them; opt-in, it measured slower) gives the ticket combine's bits: same chunk scales, same fma chains.  T = 300 has
    more than 8 chunks and stays on the ticket path either way."""
    rec, _ = G.load("loop_full_r1")
    dims = rec["dims"]
    dec = make_decoder(G.weights(rec), dims, 1)
    dec.set_tf_path("launches")
    g = torch.Generator().manual_seed(B * 1000 + T)
    enc_q = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    enc_p = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    mel_in = torch.randn(B, dims["n_mels"] + dims["aligner_out_dims"], n, generator=g).cuda()
    monkeypatch.delenv("WRNN_DEFERRED_COMBINE", raising=False)
    mel0, att0 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    monkeypatch.setenv("WRNN_DEFERRED_COMBINE", "1")
    mel1, att1 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.equal(mel0, mel1) and torch.equal(att0, att1)


@pytest.mark.parametrize("B,T,n", [(1, 16, 5), (17, 40, 4), (33, 130, 4), (64, 128, 6)])
def test_in_launch_query_is_bit_identical(B, T, n, monkeypatch):
    """The query projection rides in the attention GRU's launch and waits there for the cell's tagged output (QueryJob);
    WRNN_SEPARATE_QUERY_LAUNCH=1 keeps it a launch of its own.  Same arithmetic, chunk for chunk: same bits, in the
    teacher-forced loop and in the free-running launch loop, with ragged last row groups."""
    rec, _ = G.load("loop_full_r1")
    dims = rec["dims"]
    dec = make_decoder(G.weights(rec), dims, 1)
    dec.set_free_path("launches")
    dec.set_tf_path("launches")
    g = torch.Generator().manual_seed(B * 1000 + T)
    enc_q = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    enc_p = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    mel_in = torch.randn(B, dims["n_mels"] + dims["aligner_out_dims"], n, generator=g).cuda()
    monkeypatch.delenv("WRNN_SEPARATE_QUERY_LAUNCH", raising=False)
    monkeypatch.delenv("WRNN_FREE_MERGE_ATTENTION", raising=False)
    monkeypatch.setenv("WRNN_FREE_NO_GRU_MERGE", "1")   # (the GRU in the first launch sums its K columns in another order)
    mel0, att0 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    fm0, fa0, _ = dec.decode_free(enc_q, enc_p, n + 2)
    monkeypatch.setenv("WRNN_SEPARATE_QUERY_LAUNCH", "1")
    mel1, att1 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    fm1, fa1, _ = dec.decode_free(enc_q, enc_p, n + 2)
    assert torch.isfinite(mel0).all() and torch.isfinite(fm0).all()
    assert torch.equal(mel0, mel1) and torch.equal(att0, att1)
    assert torch.equal(fm0, fm1) and torch.equal(fa0, fa1)


@pytest.mark.parametrize("mode", ["default", "fc2_only", "attention_too"])
@pytest.mark.parametrize("B,T,n,with_ali", [(5, 16, 6, False), (17, 40, 5, True), (33, 130, 5, False), (64, 128, 8, True)])
def test_free_merged_launches_vs_six(B, T, n, with_ali, mode, monkeypatch):
    """Free-running launch loop with in-launch hand-overs (LinJob's HAND source: tagged granules) against the six-launch
    schedule (WRNN_FREE_SIX_LAUNCHES=1).  default: fc1 || mel || fc2 || GRU || query in ONE launch (fc2 waits for the first
    prenet layer, the GRU for the prenet's output, the query projection for the new state; 4 launches per step);
    fc2_only (WRNN_FREE_NO_GRU_MERGE=1): the GRU and the query projection keep their launch (5 launches);
    attention_too (WRNN_FREE_MERGE_ATTENTION=1, measured no faster): rnn_input and LSTM1 also wait in the attention's launch.
    fc2 repeats the launch's arithmetic: same bits; the GRU, rnn_input and LSTM1 take their K columns in another order (the
    awaited operand last): fp32 rounding.  Ragged last row groups, several steps of feedback, and a second call on the same
    handle (tags start over) gives the first one's bits."""
    rec, _ = G.load("loop_full_r1")
    dims = rec["dims"]
    dec = make_decoder(G.weights(rec), dims, 1)
    dec.set_free_path("launches")
    g = torch.Generator().manual_seed(B * 1000 + T + 7)
    enc_q = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    enc_p = torch.randn(B, T, dims["decoder_dims"], generator=g).cuda()
    ali = torch.randn(B, dims["aligner_out_dims"], n, generator=g).cuda() if with_ali else None
    for name in ("WRNN_FREE_SIX_LAUNCHES", "WRNN_FREE_MERGE_ATTENTION", "WRNN_FREE_NO_GRU_MERGE", "WRNN_SEPARATE_QUERY_LAUNCH"):
        monkeypatch.delenv(name, raising=False)
    if mode == "attention_too":
        monkeypatch.setenv("WRNN_FREE_MERGE_ATTENTION", "1")
    if mode == "fc2_only":
        monkeypatch.setenv("WRNN_FREE_NO_GRU_MERGE", "1")
    fm0, fa0, _ = dec.decode_free(enc_q, enc_p, n, aligner_feats=ali)
    fm2, fa2, _ = dec.decode_free(enc_q, enc_p, n, aligner_feats=ali)
    monkeypatch.setenv("WRNN_FREE_SIX_LAUNCHES", "1")
    fm1, fa1, _ = dec.decode_free(enc_q, enc_p, n, aligner_feats=ali)
    assert torch.isfinite(fm0).all() and torch.isfinite(fa0).all()
    assert torch.equal(fm0, fm2) and torch.equal(fa0, fa2)
    if mode == "fc2_only":
        assert torch.equal(fm0, fm1) and torch.equal(fa0, fa1)
    else:
        G.assert_close(fm0.cpu().numpy(), fm1.cpu().numpy(), "mel, merged vs six launches", atol_scale=1e-5)
        G.assert_close(fa0.cpu().numpy(), fa1.cpu().numpy(), "attention, merged vs six launches")


@pytest.mark.parametrize("name", G.names("free_"))
def test_free_running_vs_reference_golden(name):
    rec, gold = G.load(name)
    dec = make_decoder(G.weights(rec), rec["dims"], rec["r"])
    enc_q, enc_p, _ = (T(a).cuda() for a in G.inputs(rec))
    mel, att, n = dec.decode_free(enc_q, enc_p, rec["steps"])
    assert n == rec["steps"]
    mel, att = mel.cpu().numpy(), att.cpu().numpy()
    assert G.rel_err(mel[:, :, ::rec["keep_every"]], gold["mel_sub"]) < TOL
    G.assert_close(mel[:, :, :32], gold["mel_head"], "mel_head", atol_scale=SUM_ATOL)
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
    G.assert_close(mel.cpu().numpy(), m_ref.numpy(), "mel", atol_scale=SUM_ATOL)
    G.assert_close(att.cpu().numpy(), a_ref.numpy(), "att")
    _, _, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), 15, stop_threshold=-1e9)
    assert n == 15


def free_case(B, Tn, steps, r, with_ali, seed):
    """Seeded inputs of a free-running case plus the oracle's run and a stop threshold that fires at a group
    k* >= 33, i.e. after the library's first 32-group poll of the stop votes: group k* is the first one past 32
    whose largest frame value undercuts every earlier group's (k > 10), the threshold lies between the two."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=seed, peaky=3.0, r=r)
    enc_q, enc_p, _ = (T(a) for a in synth.decoder_inputs(B, Tn, 1, seed=31))
    ali = T(synth.normalish(78, "ali", (B, 32, steps))) * 0.1 if with_ali else None
    m_ref, a_ref, n = O.run_free(O.to_torch(sd), enc_q, enc_p, steps, r, aligner_feats=ali)
    assert n == steps
    gm = m_ref.reshape(B, 80, steps, r).amax(dim=(0, 1, 3)).numpy()
    ks = [k for k in range(33, steps) if gm[k] < gm[11:k].min()]
    return dims, sd, enc_q, enc_p, ali, m_ref, a_ref, gm, ks


@pytest.mark.parametrize("B,Tn,steps,r,with_ali,seed", [(64, 128, 64, 1, False, 10), (64, 128, 64, 2, True, 10),
                                                       (64, 128, 64, 1, True, 11), (33, 130, 48, 1, False, 10),
                                                       (8, 128, 48, 1, True, 10), (1, 128, 48, 2, False, 10)])
def test_free_running_vs_oracle_seeded(B, Tn, steps, r, with_ali, seed):
    """The free-running loop at the shape bench.py times (B=64, T=128: four 16-row groups, four attention chunks)
    and at ragged / small-batch shapes, against the oracle on the same seeded inputs: its folded layers
    (LSTM1 through rnn_input, fc1 through the mel projection, LSTM2 over [y | h1 | h2]) exist only in this loop."""
    dims, sd, enc_q, enc_p, ali, m_ref, a_ref, gm, ks = free_case(B, Tn, steps, r, with_ali, seed)
    dec = make_decoder(sd, dims, r)
    alig = None if ali is None else ali.cuda()
    mel, att, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, aligner_feats=alig)
    assert n == steps
    G.assert_close(mel.cpu().numpy(), m_ref.numpy(), "mel", atol_scale=SUM_ATOL)
    G.assert_close(att.cpu().numpy(), a_ref.numpy(), "att")
    assert (att.cpu().argmax(dim=2) == a_ref.argmax(dim=2)).all()
    # stop rule: fires at group k* (after the first poll), exactly where the oracle's does
    assert ks, "no group past 32 undercuts the earlier ones: pick another seed"
    k = max(ks, key=lambda kk: float(gm[11:kk].min() - gm[kk]))   # the widest gap: far above fp32 noise
    lo, hi = float(gm[k]), float(gm[11:k].min())
    assert hi - lo > 1e-3
    thr = 0.5 * (lo + hi)
    _, _, n_ref = O.run_free(O.to_torch(sd), enc_q, enc_p, steps, r, aligner_feats=ali, stop_threshold=thr)
    mel2, att2, n2 = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, aligner_feats=alig, stop_threshold=thr)
    assert n2 == n_ref == k + 1
    assert torch.equal(mel2, mel[:, :, :n2 * r]) and torch.equal(att2, att[:, :n2])


@pytest.mark.parametrize("B,Tn,steps,r,with_ali,drop", [(1, 128, 64, 1, False, False), (1, 20, 40, 2, True, False),
                                                        (2, 128, 48, 1, True, True), (3, 37, 40, 3, False, False),
                                                        (4, 192, 40, 1, True, False), (4, 128, 40, 1, False, True),
                                                        (3, 200, 36, 5, True, False), (1, 832, 24, 1, False, False),
                                                        (1, 300, 30, 1, True, True), (1, 1, 20, 1, False, False)])
def test_free_running_resident_vs_oracle_and_launches(B, Tn, steps, r, with_ali, drop):
    """The small-batch free-running decode as ONE persistent weights-resident launch (taco_resident.h: in-launch stage
    hand-overs, context folded into E = W . enc_q^T) against the oracle and against the launch-per-stage loop, with
    and without aligner features and prenet dropout, ragged B and T, r up to 5, and with the stop rule."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=10, peaky=3.0, r=r)
    enc_q, enc_p, _ = (T(a) for a in synth.decoder_inputs(B, Tn, 1, seed=31))
    ali = T(synth.normalish(78, "ali", (B, 32, steps))) * 0.1 if with_ali else None
    nz = {k: T(synth.uniform(41, k, (steps, B, w))) for k, w in (("drop_u1", 256), ("drop_u2", 128))} if drop else None
    m_ref, a_ref, _ = O.run_free(O.to_torch(sd), enc_q, enc_p, steps, r, aligner_feats=ali,
                                 drop_u=None if nz is None else (nz["drop_u1"], nz["drop_u2"]))
    dec = make_decoder(sd, dims, r)
    if drop:
        dec.prenet.train()
    kw = dict(aligner_feats=None if ali is None else ali.cuda(), noise=None if nz is None else {k: v.cuda() for k, v in nz.items()})
    out = {}
    for path in ("resident", "launches"):
        dec.set_free_path(path)
        mel, att, n = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, **kw)
        assert n == steps
        G.assert_close(mel.cpu().numpy(), m_ref.numpy(), ("mel", path), atol_scale=SUM_ATOL)
        G.assert_close(att.cpu().numpy(), a_ref.numpy(), ("att", path))
        assert (att.cpu().argmax(dim=2) == a_ref.argmax(dim=2)).all()
        out[path] = (mel, att)
    # the stop rule ends both paths after the same group (threshold halfway between two groups' largest frames)
    gm = m_ref.reshape(B, 80, steps, r).amax(dim=(0, 1, 3)).numpy()
    ks = [k for k in range(12, steps) if gm[k] < gm[11:k].min()]
    if ks:
        k = max(ks, key=lambda kk: float(gm[11:kk].min() - gm[kk]))
        if float(gm[11:k].min() - gm[k]) > 1e-3:
            thr = 0.5 * float(gm[k] + gm[11:k].min())
            for path in ("resident", "launches"):
                dec.set_free_path(path)
                mel2, att2, n2 = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, stop_threshold=thr, **kw)
                assert n2 == k + 1, (path, n2, k + 1)
                assert torch.equal(mel2, out[path][0][:, :, :n2 * r]) and torch.equal(att2, out[path][1][:, :n2])
    # same call twice: bit-identical (the hand-over protocol is deterministic in its results)
    dec.set_free_path("resident")
    mel3, att3, _ = dec.decode_free(enc_q.cuda(), enc_p.cuda(), steps, **kw)
    assert torch.equal(mel3, out["resident"][0]) and torch.equal(att3, out["resident"][1])


def test_free_path_selection_errors():
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    dec = make_decoder(synth.decoder_state_dict(**dims), dims, 1)
    enc_q, enc_p, _ = (T(a).cuda() for a in synth.decoder_inputs(5, 16, 1, seed=31))
    dec.set_free_path("resident")
    with pytest.raises(RuntimeError, match="resident path needs"):
        dec.decode_free(enc_q, enc_p, 4)
    with pytest.raises(ValueError):
        dec.set_free_path("nope")
    dec.set_free_path("auto")
    assert dec.decode_free(enc_q, enc_p, 4)[2] == 4


@pytest.mark.parametrize("B,Tn,S,r", [(64, 128, 24, 1), (7, 37, 12, 3), (1, 1, 5, 1), (33, 130, 8, 2),
                                         (32, 256, 20, 1), (5, 129, 19, 2),   # chain path with eight slices per row
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
    G.assert_close(mel.cpu().numpy(), m_ref.numpy(), "mel", atol_scale=SUM_ATOL)
    G.assert_close(att.cpu().numpy(), a_ref.numpy(), "att")
    # softmax rows sum to one, attention is a distribution over ALL T positions (no mask)
    assert torch.allclose(att.sum(dim=2), torch.ones_like(att.sum(dim=2)), atol=1e-5)


@pytest.mark.parametrize("B,Tn,S,r", [(64, 128, 16, 1), (5, 40, 11, 2)])
def test_training_mode_loop_vs_oracle_and_step_api(B, Tn, S, r):
    """Training-mode forward (tacotron.py:215-226): prenet dropout and zoneout on both LSTM hidden states, with
    caller-supplied uniforms, through the one-call loop; against the oracle and, bit for bit, against the step API."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0, r=r)
    dec = make_decoder(sd, dims, r).train()
    enc_q, enc_p, mel_in = (T(a) for a in synth.decoder_inputs(B, Tn, S, seed=31))
    n = (S + r - 1) // r
    u = {k: T(synth.uniform(41, k, (n, B, w))) for k, w in (("drop_u1", 256), ("drop_u2", 128), ("zone_u1", 512), ("zone_u2", 512))}
    m_ref, a_ref = O.run_teacher_forced(O.to_torch(sd), enc_q, enc_p, mel_in, r, drop_u=(u["drop_u1"], u["drop_u2"]),
                                        zone_u=(u["zone_u1"], u["zone_u2"]))
    ug = {k: v.cuda() for k, v in u.items()}
    eq, ep, mi = enc_q.cuda(), enc_p.cuda(), mel_in.cuda()
    mel, att = dec.decode_teacher_forced(eq, ep, mi, noise=ug)
    G.assert_close(mel.cpu().numpy(), m_ref.numpy(), "mel", atol_scale=SUM_ATOL)
    G.assert_close(att.cpu().numpy(), a_ref.numpy(), "att")
    # zoneout really acted: without it the result differs
    mel_nz, _ = dec.decode_teacher_forced(eq, ep, mi, noise=dict(ug, zone_u1=torch.ones_like(ug["zone_u1"]),
                                                                 zone_u2=torch.ones_like(ug["zone_u2"])))
    assert not torch.equal(mel_nz, mel)
    dec.set_tf_path("launches")   # the launch loop runs the step API's kernels: same bits
    mel_l, att_l = dec.decode_teacher_forced(eq, ep, mi, noise=ug)
    G.assert_close(mel.cpu().numpy(), mel_l.cpu().numpy(), "mel chain vs launches", atol_scale=SUM_ATOL)
    mel, att = mel_l, att_l
    z = lambda w: torch.zeros(B, w, device="cuda")
    hidden, cell, ctx = (z(256), z(512), z(512)), (z(512), z(512)), z(256)
    mels, atts = [], []
    for k, t in enumerate(range(0, S, r)):
        pin = mi[:, :, t - 1] if t > 0 else z(112)
        m, sc, hidden, cell, ctx = dec(eq, ep, pin, hidden, cell, ctx, t, noise={kk: v[k] for kk, v in ug.items()})
        mels.append(m)
        atts.append(sc)
    assert torch.equal(torch.cat(mels, dim=2), mel) and torch.equal(torch.cat(atts, dim=1), att)
    with pytest.raises(ValueError, match="noise tensor has shape"):
        dec.decode_teacher_forced(eq, ep, mi, noise=dict(ug, zone_u1=ug["zone_u1"][:, :, :8]))


@pytest.mark.parametrize("B,Tn,S,r", [(1, 1, 3, 1), (1, 128, 40, 1), (16, 31, 9, 2), (17, 70, 33, 1), (33, 97, 70, 3),
                                         (64, 128, 100, 1), (48, 128, 65, 5),
                                         (1, 129, 20, 1), (5, 200, 37, 2), (17, 256, 120, 1), (32, 256, 70, 1), (32, 161, 33, 3),
                                         # long passes: every chunk size of the chain (16, 32, 64 ..., 32, 16, 8), the benchmarked shape included
                                         (64, 128, 640, 1), (64, 128, 200, 2), (32, 256, 300, 1)])
def test_chain_path_vs_launch_path(B, Tn, S, r):
    """Teacher-forced loop, chain path (the GRU -> query -> attention recurrence as one persistent launch, the LSTM stack's
    launches beside it in chunks of steps) against the two-launch schedule: same arithmetic except the summation order
    of the query projection (and, with eight slices per row at T > 128, of the energies), so fp32 rounding apart;
    deterministic (two runs give the same bits); ragged row groups, masked positions (T < 128, T < 256), chunk boundaries of
    the stack, r > 1."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0, r=r)
    dec = make_decoder(sd, dims, r)
    enc_q, enc_p, mel_in = (T(a).cuda() for a in synth.decoder_inputs(B, Tn, S, seed=31))
    dec.set_tf_path("launches")
    mel0, att0 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    dec.set_tf_path("chain")
    mel1, att1 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    mel2, att2 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.isfinite(mel1).all() and torch.isfinite(att1).all()
    assert torch.equal(mel1, mel2) and torch.equal(att1, att2)
    long_pass = (S + r - 1) // r >= 150
    G.assert_close(mel1.cpu().numpy(), mel0.cpu().numpy(), "mel", atol_scale=LONG_SUM_ATOL if long_pass else SUM_ATOL)
    G.assert_close(att1.cpu().numpy(), att0.cpu().numpy(), "att", atol=1e-5 if long_pass else 1e-6)
    assert torch.allclose(att1.sum(dim=2), torch.ones_like(att1.sum(dim=2)), atol=1e-5)
    # 'chain' runs the LSTM stack as persistent weights-resident launches (taco_stack.h); 'chain_launch_stack' keeps the chain
    # and runs the two cells as one launch per step: the attention recurrence does not depend on the stack (teacher
    # forcing), so the attention maps are the same bits, and the mel frames agree to the gate sums' summation order
    dec.set_tf_path("chain_launch_stack")
    mel3, att3 = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.equal(att3, att1)
    G.assert_close(mel1.cpu().numpy(), mel3.cpu().numpy(), "mel resident stack vs launch stack", atol_scale=LONG_SUM_ATOL if long_pass else SUM_ATOL)


def test_chain_path_selection_and_two_handles_in_flight():
    """'chain' refuses shapes it does not cover (B > 32 at T > 128; T > 256), 'auto' falls back to launches for them; two handles on two streams at
    once (their chain launches are run one after the other by the library: each needs every compute unit) both finish
    with the results they give alone."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0)
    dec = make_decoder(sd, dims, 1)
    enc_q, enc_p, mel_in = (T(a).cuda() for a in synth.decoder_inputs(40, 130, 4, seed=31))
    dec.set_tf_path("chain")
    with pytest.raises(RuntimeError, match="chain path needs"):
        dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    dec.set_tf_path("auto")
    m_auto, _ = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    dec.set_tf_path("launches")
    m_l, _ = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    assert torch.equal(m_auto, m_l)
    with pytest.raises(ValueError, match="unknown teacher-forced path"):
        dec.set_tf_path("resident")
    dec2 = make_decoder(sd, dims, 1)
    dec.set_tf_path("chain"), dec2.set_tf_path("chain")
    a = [T(x).cuda() for x in synth.decoder_inputs(64, 128, 96, seed=5)]
    b = [T(x).cuda() for x in synth.decoder_inputs(40, 100, 80, seed=6)]
    ra, rb = dec.decode_teacher_forced(*a), dec2.decode_teacher_forced(*b)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = []
    for _ in range(3):
        with torch.cuda.stream(s1):
            xa = dec.decode_teacher_forced(*a)
        with torch.cuda.stream(s2):
            xb = dec2.decode_teacher_forced(*b)
        outs.append((xa, xb))
    torch.cuda.synchronize()
    for xa, xb in outs:
        assert torch.equal(xa[0], ra[0]) and torch.equal(xa[1], ra[1])
        assert torch.equal(xb[0], rb[0]) and torch.equal(xb[1], rb[1])


def test_weights_refresh_after_in_place_edit():
    """The handle is reused across calls under no_grad until refresh_weights() / load_state_dict / .to()."""
    from wavernn_amd import synth
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    sd = synth.decoder_state_dict(**dims, seed=9, peaky=3.0)
    dec = make_decoder(sd, dims, 1)
    enc_q, enc_p, mel_in = (T(a).cuda() for a in synth.decoder_inputs(2, 9, 4, seed=31))
    with torch.no_grad():
        m0, _ = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
        h0 = dec._handle
        dec.mel_proj.weight.mul_(2.0)
        dec.refresh_weights()
        m1, _ = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
        assert dec._handle is not h0 or dec._handle_key is not None
        assert torch.allclose(m1, 2.0 * m0, rtol=1e-6, atol=0)
        dec.load_state_dict({k: T(v) for k, v in sd.items()})
        m2, _ = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
        assert torch.equal(m2, m0)


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
    # ALL 640 steps of 5 rows against the oracle: the pass bench.py times, every chunk size of the default path included
    sub = [0, 13, 31, 32, 63]
    m_ref, a_ref = O.run_teacher_forced(O.to_torch(sd), enc_q[sub].cpu(), enc_p[sub].cpu(), mel_in[sub].cpu(), 1)
    G.assert_close(mel[sub].cpu().numpy(), m_ref.numpy(), "mel", atol_scale=LONG_SUM_ATOL)
    G.assert_close(att[sub].cpu().numpy(), a_ref.numpy(), "att", atol=1e-5)
    # the first 48 steps to the short-pass bar
    G.assert_close(mel[sub][:, :, :48].cpu().numpy(), m_ref[:, :, :48].numpy(), "mel", atol_scale=SUM_ATOL)
    G.assert_close(att[sub][:, :48].cpu().numpy(), a_ref[:, :48].numpy(), "att")
    # ... and the whole pass against the launch schedule (the step API's kernels)
    dec.set_tf_path("launches")
    mel_l, att_l = dec.decode_teacher_forced(enc_q, enc_p, mel_in)
    G.assert_close(mel.cpu().numpy(), mel_l.cpu().numpy(), "mel vs launches", atol_scale=LONG_SUM_ATOL)
    G.assert_close(att.cpu().numpy(), att_l.cpu().numpy(), "att vs launches", atol=1e-5)


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
