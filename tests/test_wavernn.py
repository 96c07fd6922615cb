"""the WaveRNN generation loop.

PARITY UNPINNED: the WaveRNN vocoder is absent from the mounted reference tree, so these tests compare the
HIP path with the build's own CPU restatement (oracle/wavernn_oracle.py) and with stock torch modules
only; nothing here is a reference-parity claim.  Bars follow BASELINE.json's north star: teacher-forced
per-step outputs within 1e-4 relative, free-running samples identical for the first 1k steps under the
same uniforms."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import golden_util as G
from oracle import wavernn_oracle as WO
from wavernn_amd import synth

SMALL = dict(rnn_dims=32, fc_dims=32, bits=6, pad=2, upsample_factors=(2, 2, 3), feat_dims=8, compute_dims=16,
             res_out_dims=16, res_blocks=2, hop_length=12, sample_rate=22050)
FULL = dict(rnn_dims=512, fc_dims=512, bits=9, pad=2, upsample_factors=(4, 8, 8), feat_dims=80, compute_dims=128,
            res_out_dims=128, res_blocks=10, hop_length=256, sample_rate=22050)   # test_config.yaml:47-57


def make(cfg, seed=0, scale=1.0):
    from wavernn_amd.wavernn import WaveRNN
    m = WaveRNN(**cfg).eval()
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    sd = synth.fill_state_dict(shapes, seed=seed)
    for k in list(sd):
        if k.startswith("upsample.up_layers"):
            del sd[k]                      # keep the 1/(2s+1) smoothing kernels
        elif k.endswith("batch_norm.weight") or "batch_norm1.weight" in k or "batch_norm2.weight" in k:
            sd[k] = synth.uniform(seed, k, shapes[k], 0.8, 1.2)
    sd["fc3.weight"] = (sd["fc3.weight"] * scale).astype(np.float32)   # peakier class distribution
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    W = {k: v.detach().clone() for k, v in m.state_dict().items()}
    return m, W


def features(m, cfg, frames, seed, target, overlap):
    mels = torch.as_tensor(synth.uniform(seed, "voc_mel", (1, cfg["feat_dims"], frames), -4.0, 0.0))
    with torch.no_grad():
        mu, au = m.upsample(F.pad(mels, (cfg["pad"], cfg["pad"])))
    return mels, m.fold_with_overlap(mu, target, overlap), m.fold_with_overlap(au, target, overlap)


# ------------------------------------------------------------------ CPU: host logic vs the oracle
def test_upsample_and_fold_match_oracle():
    m, W = make(SMALL)
    mels, mu_f, au_f = features(m, SMALL, 30, 1, 100, 10)
    mo, ao = WO.upsample(W, F.pad(mels, (2, 2)), SMALL["upsample_factors"], SMALL["res_blocks"], SMALL["pad"])
    assert torch.equal(mu_f, WO.fold_with_overlap(mo, 100, 10))
    assert torch.equal(au_f, WO.fold_with_overlap(ao, 100, 10))
    assert mu_f.shape == (4, 120, 8) and au_f.shape == (4, 120, 16)


def test_fold_unfold_round_trip():
    """fold -> crossfade/unfold of a constant signal reproduces it where folds overlap (equal-power fades
    sum to >= 1 only at the ends of the fade; the plateau and the fold interiors are exact)."""
    from wavernn_amd.wavernn import WaveRNN
    x = torch.ones(1, 1000, 1)
    f = WaveRNN.fold_with_overlap(x, 200, 20)
    y = WaveRNN.xfade_and_unfold(f[:, :, 0].numpy().astype(np.float64), 20)
    assert np.array_equal(y, WO.xfade_and_unfold(f[:, :, 0].numpy(), 20))
    assert np.allclose(y[20:200], 1.0) and y.shape[0] == f.shape[0] * 220 + 20


def test_oracle_loop_matches_torch_gru_modules():
    m, W = make(SMALL)
    _, mu, au = features(m, SMALL, 30, 2, 100, 10)
    B, seq, a = mu.shape[0], 24, m.aux_dims
    u = torch.as_tensor(synth.uniform01(3, "u", B * seq).reshape(B, seq))
    s, lg = WO.generate_loop(W, mu[:, :seq], au[:, :seq], u, return_logits=True)
    with torch.no_grad():
        h1 = torch.zeros(1, B, 32); h2 = torch.zeros(1, B, 32); x = torch.zeros(B, 1)
        for t in range(seq):
            xx = m.I(torch.cat([x, mu[:, t], au[:, t, :a]], 1))
            o, h1 = m.rnn1(xx[:, None], h1); xx = xx + o[:, 0]
            o, h2 = m.rnn2(torch.cat([xx, au[:, t, a:2 * a]], 1)[:, None], h2); xx = xx + o[:, 0]
            xx = torch.relu(m.fc1(torch.cat([xx, au[:, t, 2 * a:3 * a]], 1)))
            xx = torch.relu(m.fc2(torch.cat([xx, au[:, t, 3 * a:]], 1)))
            assert G.rel_err(m.fc3(xx).numpy(), lg[:, t].numpy()) < 1e-5
            x = s[:, t][:, None]
    assert s.min() >= -1 and s.max() <= 1


def test_cpu_call_raises():
    m, _ = make(SMALL)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.generate(torch.zeros(1, 8, 12), batched=False)


# ------------------------------------------------------------------ GPU vs the build's oracle
@pytest.mark.gpu
@pytest.mark.parametrize("cfg,frames,target,overlap,seq", [(SMALL, 40, 100, 10, 120), (FULL, 12, 700, 60, 64)])
def test_gpu_teacher_forced_logits(cfg, frames, target, overlap, seq):
    m, W = make(cfg, seed=5)
    _, mu, au = features(m, cfg, frames, 6, target, overlap)
    mu, au = mu[:, :seq].contiguous(), au[:, :seq].contiguous()
    B = mu.shape[0]
    u = torch.as_tensor(synth.uniform01(7, "u", B * seq).reshape(B, seq))
    xf = torch.as_tensor(synth.uniform(8, "xf", (B, seq), -1.0, 1.0))
    s_ref, lg_ref = WO.generate_loop(W, mu, au, u, x_forced=xf, return_logits=True)
    s, lg = m.cuda().generate_loop(mu.cuda(), au.cuda(), u.cuda(), x_forced=xf.cuda(), return_logits=True)
    G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), "lg.cpu().numpy()", atol_scale=1e-5)
    assert (s.cpu() == s_ref).float().mean() > 0.999      # same logits, same uniforms: same classes


@pytest.mark.gpu
def test_gpu_free_running_sample_exact_1k_steps():
    m, W = make(FULL, seed=9, scale=4.0)
    _, mu, au = features(m, FULL, 12, 10, 1000, 60)
    mu, au = mu[:3, :1000].contiguous(), au[:3, :1000].contiguous()
    u = torch.as_tensor(synth.uniform01(11, "u", 3 * 1000).reshape(3, 1000))
    s_ref = WO.generate_loop(W, mu, au, u)
    s = m.cuda().generate_loop(mu.cuda(), au.cuda(), u.cuda()).cpu()
    assert torch.equal(s, s_ref)


@pytest.mark.gpu
def test_gpu_whole_generate_vs_oracle():
    cfg = SMALL
    m, W = make(cfg, seed=12, scale=3.0)
    mels = torch.as_tensor(synth.uniform(13, "voc_mel", (1, 8, 60), -4.0, 0.0))
    folds = m.fold_with_overlap(torch.zeros(1, 60 * 12, 1), 200, 20).shape[0]
    u = torch.as_tensor(synth.uniform01(14, "u", folds * 240).reshape(folds, 240))
    ref = WO.generate(W, mels, cfg, u, batched=True, target=200, overlap=20)
    out = m.cuda().generate(mels, batched=True, target=200, overlap=20, noise=u)
    assert out.shape == ref.shape == ((60 - 1) * 12,)
    assert np.array_equal(out, ref)
    one = m.generate(mels, batched=False, noise=u[:1].repeat(1, 3)[:, :720])
    assert one.shape == ((60 - 1) * 12,) and np.abs(one).max() <= 1.0


@pytest.mark.gpu
def test_gpu_row_independence_full_size():
    """BASELINE configs[1] shape: 10 s of audio = 20 folds of 12,100 steps (first 300 steps here): a
    permuted batch gives the permuted samples bit for bit."""
    m, _ = make(FULL, seed=15, scale=4.0)
    _, mu, au = features(m, FULL, 862, 16, 11000, 550)
    assert mu.shape[0] == 20 and mu.shape[1] == 12100
    mu, au = mu[:, :300].contiguous().cuda(), au[:, :300].contiguous().cuda()
    u = torch.as_tensor(synth.uniform01(17, "u", 20 * 300).reshape(20, 300)).cuda()
    m = m.cuda()
    s = m.generate_loop(mu, au, u)
    perm = torch.randperm(20, generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(m.generate_loop(mu[perm].contiguous(), au[perm].contiguous(), u[perm].contiguous()), s[perm])
    assert s.abs().max() <= 1.0


@pytest.mark.gpu
def test_gpu_config1_full_length_batched():
    """BASELINE configs[1] at its stated size: 10 s of audio, target 11000 / overlap 550 = 20 folds x the full 12,100
    steps (2.2 GB workspace, graph-replayed loop).  The first 1,000 steps of every fold are sample-exact against the
    build's oracle under the same uniforms (free-running: each sample feeds the next step); the whole output is in
    range and a permuted batch gives the permuted samples bit for bit; generate() on the same mel returns the
    cross-faded 10 s."""
    m, W = make(FULL, seed=15, scale=4.0)
    mels, mu, au = features(m, FULL, 862, 16, 11000, 550)
    assert mu.shape == (20, 12100, 80) and au.shape == (20, 12100, 128)
    u = torch.as_tensor(synth.uniform01(17, "u", 20 * 12100).reshape(20, 12100))
    s_ref = WO.generate_loop(W, mu[:, :1000], au[:, :1000], u[:, :1000])
    m = m.cuda()
    mug, aug, ug = mu.cuda(), au.cuda(), u.cuda()
    s = m.generate_loop(mug, aug, ug)
    assert s.shape == (20, 12100)
    assert torch.equal(s[:, :1000].cpu(), s_ref)
    assert torch.isfinite(s).all() and s.abs().max() <= 1.0
    sc = s.cpu()                                                     # (host arithmetic: torch's GPU division by a scalar
    cls = torch.round((sc + 1.0) * 0.5 * 511).long()                 # multiplies by the reciprocal)
    assert torch.equal(2.0 * cls.float() / 511.0 - 1.0, sc) and cls.unique().numel() >= 8   # one of the 512 class values
    perm = torch.randperm(20, generator=torch.Generator().manual_seed(2)).cuda()
    assert torch.equal(m.generate_loop(mug[perm].contiguous(), aug[perm].contiguous(), ug[perm].contiguous()), s[perm])
    wav = m.generate(mels, batched=True, target=11000, overlap=550, noise=u)
    assert wav.shape == (861 * 256,) and np.isfinite(wav).all() and np.abs(wav).max() <= 1.5   # crossfade sums two folds
    # a fold's interior (outside the cross-faded overlaps) is the loop's output untouched (generate() runs the upsampling
    # network on the GPU, whose features differ from the CPU's in the last bits: compare with the loop on ITS features)
    with torch.no_grad():
        mg, ag = m.upsample(F.pad(mels.cuda(), (2, 2)))
    sg = m.generate_loop(m.fold_with_overlap(mg, 11000, 550), m.fold_with_overlap(ag, 11000, 550), ug)
    assert np.array_equal(wav[550:11000], sg[0, 550:11000].cpu().numpy().astype(np.float64))


@pytest.mark.gpu
def test_gpu_config4_unbatched_5s():
    """BASELINE configs[4] at its stated size: unbatched (one row) sample-by-sample generation of 5 s of audio at
    rnn_dims = 512: 110,080 steps (431 frames x 256).  First 1,000 steps sample-exact against the oracle, the rest
    finite and in [-1, 1]; generate(batched=False) returns the same samples (plus its fade-out)."""
    m, W = make(FULL, seed=21, scale=4.0)
    frames = 431
    mels = torch.as_tensor(synth.uniform(22, "voc_mel", (1, 80, frames), -4.0, 0.0))
    with torch.no_grad():
        mu, au = m.upsample(F.pad(mels, (2, 2)))
    seq = frames * 256
    assert mu.shape == (1, seq, 80)
    u = torch.as_tensor(synth.uniform01(23, "u", seq).reshape(1, seq))
    s_ref = WO.generate_loop(W, mu[:, :1000], au[:, :1000], u[:, :1000])
    m = m.cuda()
    s = m.generate_loop(mu.cuda(), au.cuda(), u.cuda())
    assert torch.equal(s[:, :1000].cpu(), s_ref)
    assert torch.isfinite(s).all() and s.abs().max() <= 1.0 and s.unique().numel() >= 8
    wav = m.generate(mels, batched=False, noise=u)
    n = (frames - 1) * 256
    assert wav.shape == (n,)
    with torch.no_grad():                       # generate() upsamples on the GPU: compare with the loop on those features
        mg, ag = m.upsample(F.pad(mels.cuda(), (2, 2)))
    sg = m.generate_loop(mg, ag, u.cuda())
    assert np.array_equal(wav[:n - 20 * 256], sg[0, :n - 20 * 256].cpu().numpy().astype(np.float64))


@pytest.mark.gpu
def test_gpu_generate_batch_equals_generate():
    """Several utterances' folds as the rows of one loop call: each utterance's audio is what generate()
    gives it on its own with the same uniforms (rows are independent), incl. a split into two loop calls."""
    cfg = SMALL
    m, _ = make(cfg, seed=18, scale=3.0)
    m = m.cuda()
    mels = [torch.as_tensor(synth.uniform(19 + i, "voc_mel", (1, 8, fr), -4.0, 0.0)) for i, fr in enumerate((60, 25, 83))]
    noise = []
    for i, x in enumerate(mels):
        folds = m.fold_with_overlap(torch.zeros(1, x.shape[2] * 12, 1), 200, 20).shape[0]
        noise.append(torch.as_tensor(synth.uniform01(30 + i, "u", folds * 240).reshape(folds, 240)))
    single = [m.generate(x, batched=True, target=200, overlap=20, noise=n) for x, n in zip(mels, noise)]
    for max_rows in (256, 5):
        many = m.generate_batch(mels, target=200, overlap=20, noise=noise, max_rows=max_rows)
        assert len(many) == 3
        for a, b in zip(many, single):
            assert np.array_equal(a, b)


@pytest.mark.gpu
def test_gpu_empty_loop():
    m, _ = make(SMALL)
    m = m.cuda()
    out = m.generate_loop(torch.zeros(0, 5, 8).cuda(), torch.zeros(0, 5, 16).cuda(), torch.zeros(0, 5).cuda())
    assert out.shape == (0, 5)
    out = m.generate_loop(torch.zeros(2, 0, 8).cuda(), torch.zeros(2, 0, 16).cuda(), torch.zeros(2, 0).cuda())
    assert out.shape == (2, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,frames,target,overlap,seq", [(SMALL, 40, 100, 10, 120), (FULL, 12, 700, 60, 64)])
def test_gpu_mol_mode_loop_vs_oracle(cfg, frames, target, overlap, seq):
    """MOL mode: fc3 emits 30 mixture parameters and the step's sample comes from the sampler of the reference's
    utils/distribution.py:87-123 (pinned: tests/test_mol.py, tests/test_oracle_golden.py); the network around it is
    build-defined (unpinned).  Teacher-forced: the 30 mixture parameters within 1e-4 relative and every step's sample
    within 1e-4 of the oracle's (the sample is mean + exp(log_scale) * logit(u): at random init the scales are large, so
    the parameters' fp32 differences show up magnified in it); free-running (the continuous sample feeds the next
    step): within 1e-3 over the first 100 steps."""
    from wavernn_amd.distribution import U_HI, U_LO
    m, W = make(dict(cfg, mode="MOL"), seed=5)
    assert m.n_classes == 30 and W["fc3.weight"].shape[0] == 30
    _, mu, au = features(m, cfg, frames, 6, target, overlap)
    mu, au = mu[:, :seq].contiguous(), au[:, :seq].contiguous()
    B = mu.shape[0]
    u = torch.as_tensor((U_LO + (U_HI - U_LO) * synth.uniform01(7, "u", B * seq * 11).astype(np.float64)).astype(np.float32).reshape(B, seq, 11))
    xf = torch.as_tensor(synth.uniform(8, "xf", (B, seq), -1.0, 1.0))
    s_ref, lg_ref = WO.generate_loop(W, mu, au, u, x_forced=xf, return_logits=True, mode="MOL")
    mg = m.cuda()
    s, lg = mg.generate_loop(mu.cuda(), au.cuda(), u.cuda(), x_forced=xf.cuda(), return_logits=True)
    assert lg.shape == (B, seq, 30)
    G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), "lg.cpu().numpy()", atol_scale=1e-5)
    assert float((s.cpu() - s_ref).abs().max()) < 1e-4 and float(s.abs().max()) <= 1.0
    f_ref = WO.generate_loop(W, mu[:, :100], au[:, :100], u[:, :100], mode="MOL")
    f = mg.generate_loop(mu[:, :100].contiguous().cuda(), au[:, :100].contiguous().cuda(), u[:, :100].contiguous().cuda())
    assert float((f.cpu() - f_ref).abs().max()) < 1e-3 and f.unique().numel() > (50 if cfg is SMALL else 2)
    with pytest.raises(ValueError, match="uniforms must have shape"):
        mg.generate_loop(mu.cuda(), au.cuda(), u[:, :, 0].cuda())


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [9, 7, 6])
def test_gpu_resident_unbatched_vs_oracle_and_launches(bits):
    """One row (unbatched generation, BASELINE configs[4]) runs as ONE persistent weights-resident launch per call
    (wavernn_resident.h).  Teacher-forced: logits within 1e-4 of the oracle and of the launch loop; free-running: the first
    1,000 steps sample-exact against the oracle and identical to the launch loop's (both draw with the same arithmetic)."""
    cfg = dict(FULL, bits=bits)
    m, W = make(cfg, seed=31 + bits, scale=4.0)
    _, mu, au = features(m, cfg, 12, 32, 1000, 60)
    seq = 1000
    mu, au = mu[:1, :seq].contiguous(), au[:1, :seq].contiguous()
    u = torch.as_tensor(synth.uniform01(33, "u", seq).reshape(1, seq))
    xf = torch.as_tensor(synth.uniform(34, "xf", (1, seq), -1.0, 1.0))
    s_ref, lg_ref = WO.generate_loop(W, mu[:, :200], au[:, :200], u[:, :200], x_forced=xf[:, :200], return_logits=True)
    f_ref = WO.generate_loop(W, mu, au, u)
    m = m.cuda()
    mug, aug, ug, xfg = mu.cuda(), au.cuda(), u.cuda(), xf.cuda()
    res = {}
    for path in ("resident", "launches"):
        m.set_path(path)
        s, lg = m.generate_loop(mug[:, :200].contiguous(), aug[:, :200].contiguous(), ug[:, :200].contiguous(),
                                x_forced=xfg[:, :200].contiguous(), return_logits=True)
        G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), path, atol_scale=1e-5)
        assert (s.cpu() == s_ref).float().mean() > 0.995, path
        res[path] = (lg.cpu(), m.generate_loop(mug, aug, ug).cpu())
        assert torch.equal(res[path][1], f_ref), path
    G.assert_close(res["resident"][0].numpy(), res["launches"][0].numpy(), "res[resident][0].numpy()", atol_scale=1e-5)
    m.set_path("auto")
    assert torch.equal(m.generate_loop(mug, aug, ug).cpu(), f_ref)


@pytest.mark.gpu
def test_gpu_resident_mol_and_errors():
    from wavernn_amd.distribution import U_HI, U_LO
    cfg = dict(FULL, mode="MOL")
    m, W = make(cfg, seed=41)
    _, mu, au = features(m, FULL, 12, 42, 700, 60)
    seq = 100
    mu, au = mu[:1, :seq].contiguous(), au[:1, :seq].contiguous()
    u = torch.as_tensor((U_LO + (U_HI - U_LO) * synth.uniform01(43, "u", seq * 11).astype(np.float64)).astype(np.float32).reshape(1, seq, 11))
    xf = torch.as_tensor(synth.uniform(44, "xf", (1, seq), -1.0, 1.0))
    s_ref, lg_ref = WO.generate_loop(W, mu, au, u, x_forced=xf, return_logits=True, mode="MOL")
    f_ref = WO.generate_loop(W, mu, au, u, mode="MOL")
    m = m.cuda()
    m.set_path("resident")
    s, lg = m.generate_loop(mu.cuda(), au.cuda(), u.cuda(), x_forced=xf.cuda(), return_logits=True)
    G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), "lg.cpu().numpy()", atol_scale=1e-5)
    assert float((s.cpu() - s_ref).abs().max()) < 1e-4
    f = m.generate_loop(mu.cuda(), au.cuda(), u.cuda())
    assert float((f.cpu() - f_ref).abs().max()) < 1e-3
    # forced onto the persistent kernels, calls they cannot serve fail loudly: more than 32 rows, another model shape
    with pytest.raises(RuntimeError, match="at most 32 rows per call"):
        m.generate_loop(mu.repeat(33, 1, 1).cuda(), au.repeat(33, 1, 1).cuda(), u.repeat(33, 1, 1).cuda())
    small, _ = make(SMALL)
    small = small.cuda()
    small.set_path("resident")
    with pytest.raises(RuntimeError, match="persistent kernels need"):
        small.generate_loop(torch.zeros(1, 5, 8).cuda(), torch.zeros(1, 5, 16).cuda(), torch.rand(1, 5).cuda())
    with pytest.raises(ValueError, match="unknown path"):
        small.set_path("fast")


@pytest.mark.gpu
@pytest.mark.parametrize("bits,B", [(9, 20), (9, 2), (9, 16), (9, 32), (7, 17), (6, 5)])
def test_gpu_folds_resident_vs_oracle_and_launches(bits, B):
    """2 .. 32 rows (the folds of an utterance, BASELINE configs[1]) run as ONE persistent weights-resident launch per call
    (wavernn_folds.h: the matrix stages on the matrix cores, register-resident weights, tagged hand-overs).  Teacher-forced:
    logits within 1e-4 of the oracle and of the launch loop; free-running: the first 1,000 steps sample-exact against the
    oracle and identical to the launch loop's; a permuted batch gives the permuted samples bit for bit."""
    cfg = dict(FULL, bits=bits)
    m, W = make(cfg, seed=51 + bits, scale=4.0)
    _, mu, au = features(m, cfg, 862, 52, 11000, 550)
    assert mu.shape[0] == 20
    seq = 1000
    reps = (B + 19) // 20
    mu, au = mu.repeat(reps, 1, 1)[:B, :seq].contiguous(), au.repeat(reps, 1, 1)[:B, :seq].contiguous()
    u = torch.as_tensor(synth.uniform01(53, "u", B * seq).reshape(B, seq))
    xf = torch.as_tensor(synth.uniform(54, "xf", (B, seq), -1.0, 1.0))
    n = 120
    s_ref, lg_ref = WO.generate_loop(W, mu[:, :n], au[:, :n], u[:, :n], x_forced=xf[:, :n], return_logits=True)
    f_ref = WO.generate_loop(W, mu, au, u)
    m = m.cuda()
    mug, aug, ug, xfg = mu.cuda(), au.cuda(), u.cuda(), xf.cuda()
    res = {}
    for path in ("resident", "launches"):
        m.set_path(path)
        s, lg = m.generate_loop(mug[:, :n].contiguous(), aug[:, :n].contiguous(), ug[:, :n].contiguous(),
                                x_forced=xfg[:, :n].contiguous(), return_logits=True)
        G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), path, atol_scale=1e-5)
        assert (s.cpu() == s_ref).float().mean() > 0.995, path
        res[path] = (lg.cpu(), m.generate_loop(mug, aug, ug).cpu())
        assert torch.equal(res[path][1], f_ref), path
    G.assert_close(res["resident"][0].numpy(), res["launches"][0].numpy(), "res[resident][0].numpy()", atol_scale=1e-5)
    m.set_path("auto")                                     # AUTO: the persistent kernel for 2 .. 32 rows
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    sp = m.generate_loop(mug[perm].contiguous(), aug[perm].contiguous(), ug[perm].contiguous())
    assert torch.equal(sp.cpu(), res["resident"][1][perm.cpu()])
    assert torch.equal(m.generate_loop(mug, aug, ug).cpu(), res["resident"][1])   # and again: nothing left over from a call


@pytest.mark.gpu
def test_gpu_folds_resident_mol():
    from wavernn_amd.distribution import U_HI, U_LO
    cfg = dict(FULL, mode="MOL")
    m, W = make(cfg, seed=61)
    _, mu, au = features(m, FULL, 862, 62, 11000, 550)
    B, seq = 20, 100
    mu, au = mu[:, :seq].contiguous(), au[:, :seq].contiguous()
    u = torch.as_tensor((U_LO + (U_HI - U_LO) * synth.uniform01(63, "u", B * seq * 11).astype(np.float64)).astype(np.float32).reshape(B, seq, 11))
    xf = torch.as_tensor(synth.uniform(64, "xf", (B, seq), -1.0, 1.0))
    s_ref, lg_ref = WO.generate_loop(W, mu, au, u, x_forced=xf, return_logits=True, mode="MOL")
    f_ref = WO.generate_loop(W, mu, au, u, mode="MOL")
    m = m.cuda()
    m.set_path("resident")
    s, lg = m.generate_loop(mu.cuda(), au.cuda(), u.cuda(), x_forced=xf.cuda(), return_logits=True)
    G.assert_close(lg.cpu().numpy(), lg_ref.numpy(), "lg.cpu().numpy()", atol_scale=1e-5)
    assert float((s.cpu() - s_ref).abs().max()) < 1e-4
    f = m.generate_loop(mu.cuda(), au.cuda(), u.cuda())
    assert float((f.cpu() - f_ref).abs().max()) < 1e-3


@pytest.mark.gpu
def test_gpu_auto_splits_33_to_64_rows():
    """AUTO runs 33 .. 64 rows as two calls of the persistent kernel: the same samples as each half on its own, bit for bit,
    logits within 1e-4 of the launch loop's."""
    m, W = make(FULL, seed=71, scale=4.0)
    _, mu, au = features(m, FULL, 862, 72, 11000, 550)
    B, seq = 41, 160
    mu, au = mu.repeat(3, 1, 1)[:B, :seq].contiguous().cuda(), au.repeat(3, 1, 1)[:B, :seq].contiguous().cuda()
    u = torch.as_tensor(synth.uniform01(73, "u", B * seq).reshape(B, seq)).cuda()
    xf = torch.as_tensor(synth.uniform(74, "xf", (B, seq), -1.0, 1.0)).cuda()
    m = m.cuda()
    s_auto, lg_auto = m.generate_loop(mu, au, u, x_forced=xf, return_logits=True)
    f_auto = m.generate_loop(mu, au, u)
    m.set_path("resident")
    halves = [m.generate_loop(mu[a:b].contiguous(), au[a:b].contiguous(), u[a:b].contiguous()) for a, b in ((0, 21), (21, 41))]
    assert torch.equal(f_auto, torch.cat(halves))
    m.set_path("launches")
    s_l, lg_l = m.generate_loop(mu, au, u, x_forced=xf, return_logits=True)
    G.assert_close(lg_auto.cpu().numpy(), lg_l.cpu().numpy(), "lg_auto", atol_scale=1e-5)
    assert (s_auto == s_l).float().mean() > 0.995


@pytest.mark.gpu
@pytest.mark.parametrize("seq", [1, 2, 3, 7])
def test_gpu_folds_resident_very_short_calls(seq):
    """One to a few steps through the fold kernel (the first steps have no previous sample, state or hidden-half sums; the last
    sample is drawn after the loop): logits and samples as the launch loop's."""
    m, _ = make(FULL, seed=81, scale=4.0)
    m = m.cuda()
    B = 3
    mu = torch.as_tensor(synth.uniform(82, "mu", (B, seq, 80), -1.0, 1.0)).cuda()
    au = torch.as_tensor(synth.uniform(83, "au", (B, seq, 128), -1.0, 1.0)).cuda()
    u = torch.as_tensor(synth.uniform01(84, "u", B * seq).reshape(B, seq)).cuda()
    res = {}
    for path in ("resident", "launches"):
        m.set_path(path)
        res[path] = m.generate_loop(mu, au, u, return_logits=True)
    assert torch.equal(res["resident"][0], res["launches"][0])
    G.assert_close(res["resident"][1].cpu().numpy(), res["launches"][1].cpu().numpy(), "logits", atol_scale=1e-5)
