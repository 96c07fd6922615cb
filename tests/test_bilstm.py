"""the bidirectional LSTM that follows the length regulator
(models/forward_tacotron.py:96-99,147-152,224).  Fixtures: tensors captured with forward hooks while the
reference ForwardTacotron ran (tools/make_golden.py::case_forward_lstm)."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

import golden_util as G
from oracle import taco_oracle as O
from wavernn_amd import synth

TOL = 1e-4  # north-star relative tolerance (fp32)


def lstm_weights(rec):
    shapes = {"lstm." + k: tuple(v) for k, v in rec["lstm_shapes"].items()}
    sd = synth.fill_state_dict(shapes, seed=rec["seed"])
    return {k[len("lstm."):]: v for k, v in sd.items()}


def module(W, in_dims, H):
    from wavernn_amd.bilstm import BiLSTM
    m = BiLSTM(in_dims, H)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in W.items()})
    return m


def synth_case(B, M, in_dims, H, seed):
    names = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"]
    shapes = {}
    for sfx in ("", "_reverse"):
        shapes["weight_ih_l0" + sfx] = (4 * H, in_dims)
        shapes["weight_hh_l0" + sfx] = (4 * H, H)
        shapes["bias_ih_l0" + sfx] = (4 * H,)
        shapes["bias_hh_l0" + sfx] = (4 * H,)
    del names
    W = synth.fill_state_dict(shapes, seed=seed)
    x = synth.normalish(seed, "bilstm_x", (B, M, in_dims))
    return W, x


# ------------------------------------------------------------------ CPU: oracle vs reference fixtures
@pytest.mark.parametrize("name", G.names("fwdlstm_"))
def test_oracle_vs_reference_packed(name):
    rec, gold = G.load(name)
    W = O.to_torch(lstm_weights(rec))
    out, h_n, c_n = O.bilstm(W, torch.as_tensor(gold["lr_out"]), gold["mel_len"], rec["padding_value"])
    assert G.rel_err(out.numpy(), gold["packed_lstm_out"]) < 1e-5
    assert G.rel_err(h_n.numpy(), gold["packed_h_n"]) < 1e-5
    assert G.rel_err(c_n.numpy(), gold["packed_c_n"]) < 1e-5


@pytest.mark.parametrize("name", G.names("fwdlstm_"))
def test_oracle_vs_reference_padded(name):
    rec, gold = G.load(name)
    W = O.to_torch(lstm_weights(rec))
    out, h_n, c_n = O.bilstm(W, torch.as_tensor(gold["lr_out"]))
    assert G.rel_err(out.numpy(), gold["padded_lstm_out"]) < 1e-5
    assert G.rel_err(h_n.numpy(), gold["padded_h_n"]) < 1e-5
    assert G.rel_err(c_n.numpy(), gold["padded_c_n"]) < 1e-5


def test_parameter_names_match_torch_lstm():
    from wavernn_amd.bilstm import BiLSTM
    ref = torch.nn.LSTM(32, 20, batch_first=True, bidirectional=True)
    mine = BiLSTM(32, 20)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert list(mine.state_dict()) == list(ref.state_dict())


def test_cpu_call_raises():
    from wavernn_amd.bilstm import BiLSTM
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BiLSTM(8, 4)(torch.zeros(1, 3, 8))


# ------------------------------------------------------------------ GPU parity, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("name", G.names("fwdlstm_"))
def test_gpu_vs_reference_fixture(name):
    rec, gold = G.load(name)
    W = lstm_weights(rec)
    x = torch.as_tensor(gold["lr_out"]).cuda()
    m = module(W, x.shape[2], rec["cfg"]["rnn_dims"]).cuda()
    out, (h_n, c_n) = m.run_padded(x, torch.as_tensor(gold["mel_len"]), rec["padding_value"])
    for got, key in ((out, "packed_lstm_out"), (h_n, "packed_h_n"), (c_n, "packed_c_n")):
        G.assert_close(got.cpu().numpy(), gold[key], key, atol_scale=1e-5)
    out, (h_n, c_n) = m(x)
    for got, key in ((out, "padded_lstm_out"), (h_n, "padded_h_n"), (c_n, "padded_c_n")):
        G.assert_close(got.cpu().numpy(), gold[key], key, atol_scale=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,in_dims,H,ragged", [(16, 40, 512, 512, True), (1, 7, 64, 36, False),
                                                  (33, 19, 128, 256, True), (5, 1, 32, 16, False)])
def test_gpu_vs_oracle(B, M, in_dims, H, ragged):
    W, x = synth_case(B, M, in_dims, H, seed=600 + B)
    lens = None
    if ragged:
        lens = 1 + (synth.uniform01(7, "lens", B) * M).astype(np.int64)
        lens[0], lens[-1] = M, 1
    ref_out, ref_h, ref_c = O.bilstm(O.to_torch(W), torch.as_tensor(x), lens, -11.5129)
    m = module(W, in_dims, H).cuda()
    out, (h_n, c_n) = m.run_padded(torch.as_tensor(x).cuda(), None if lens is None else torch.as_tensor(lens), -11.5129)
    G.assert_close(out.cpu().numpy(), ref_out.numpy(), "out.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(h_n.cpu().numpy(), ref_h.numpy(), "h_n.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(c_n.cpu().numpy(), ref_c.numpy(), "c_n.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,ragged", [(1, 1, False), (7, 12, True), (17, 9, False), (33, 21, True), (49, 64, True),
                                        (64, 130, True), (64, 33, False), (24, 470, True)])
def test_gpu_resident_path_vs_oracle_and_launches(B, M, ragged):
    """The persistent weights-in-registers launch (in = H = 512, B <= 64; one to four 16-row tiles, ragged lengths, both
    directions; M = 470: three launches of at most 224 frames, the state and the tagged h ring crossing the boundaries)
    against the oracle, and against the launch loop to fp32 summation order plus the tag bit (the resident path's hidden state
    carries the frame's tag in its last mantissa bit: at most one ulp per frame); two runs give the same bits."""
    W, x = synth_case(B, M, 512, 512, seed=640 + B)
    lens = None
    if ragged:
        lens = 1 + (synth.uniform01(9, "lens", B) * M).astype(np.int64)
        lens[0], lens[-1] = M, 1
    tl = None if lens is None else torch.as_tensor(lens)
    ref = O.bilstm(O.to_torch(W), torch.as_tensor(x), lens, 0.5)
    m = module(W, 512, 512).cuda()
    xg = torch.as_tensor(x).cuda()
    got = {}
    for path in ("resident", "launches"):
        m.set_path(path)
        out, (h_n, c_n) = m.run_padded(xg, tl, 0.5)
        got[path] = (out, h_n, c_n)
        for g, r in zip(got[path], ref):
            G.assert_close(g.cpu().numpy(), r.numpy(), path, atol_scale=1e-5)
    for a, b_ in zip(got["resident"], got["launches"]):
        assert float((a - b_).abs().max()) < 5e-6
    m.set_path("resident")
    out2, (h2, c2) = m.run_padded(xg, tl, 0.5)
    assert torch.equal(out2, got["resident"][0]) and torch.equal(h2, got["resident"][1]) and torch.equal(c2, got["resident"][2])


@pytest.mark.gpu
def test_gpu_flag_kernels_in_a_subprocess():
    """WRNN_BILSTM_FLAGS=1 (read once per process) selects round 2's flag hand-over kernels, kept for comparison: they must
    still agree with the launch loop.  Run in a child process so that the variable is seen."""
    import os
    import subprocess
    import sys
    code = (
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import test_bilstm as T\n"
        "for B, M in ((7, 12), (33, 21), (64, 40)):\n"
        "    W, x = T.synth_case(B, M, 512, 512, seed=3)\n"
        "    m = T.module(W, 512, 512).cuda(); xg = torch.as_tensor(x).cuda(); got = []\n"
        "    for path in ('resident', 'launches'):\n"
        "        m.set_path(path); out, (h, c) = m.run_padded(xg, None, 0.0); got.append(out)\n"
        "    d = float((got[0] - got[1]).abs().max()); assert d < 5e-6, d\n"
        "print('ok')\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WRNN_BILSTM_FLAGS="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_gpu_resident_calls_in_flight_on_several_streams():
    """Two BiLSTM handles, a Decoder on the chain path and one on the free-running persistent kernel, enqueued on four streams: the persistent
    grids (each needs all its workgroups resident) are queued behind each other on the device's stream for them, and every
    result equals the one of the same call made alone, bit for bit."""
    from wavernn_amd import Decoder
    W1, x1 = synth_case(64, 60, 512, 512, seed=660)
    W2, x2 = synth_case(20, 90, 512, 512, seed=661)
    m1, m2 = module(W1, 512, 512).cuda(), module(W2, 512, 512).cuda()
    xg1, xg2 = torch.as_tensor(x1).cuda(), torch.as_tensor(x2).cuda()
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    dec = Decoder(**dims)
    dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in synth.decoder_state_dict(**dims, seed=3).items()})
    dec = dec.cuda().eval()
    dec.set_tf_path("chain")
    ins = [torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in synth.decoder_inputs(16, 40, 48, seed=5)]
    m1.set_path("resident"), m2.set_path("resident")
    dec1 = Decoder(**dims)   # a second decoder for the free-running persistent kernel (one utterance)
    dec1.load_state_dict(dec.state_dict())
    dec1 = dec1.cuda().eval()
    dec1.set_free_path("resident")
    alone = (m1.run_padded(xg1)[0], m2.run_padded(xg2)[0], dec.decode_teacher_forced(*ins)[0],
             dec1.decode_free(ins[0][:1].contiguous(), ins[1][:1].contiguous(), 30)[0])
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(4)]
    got = [None] * 4
    for rep in range(3):
        with torch.cuda.stream(streams[0]):
            got[0] = m1.run_padded(xg1)[0]
        with torch.cuda.stream(streams[2]):
            got[2] = dec.decode_teacher_forced(*ins)[0]
        with torch.cuda.stream(streams[1]):
            got[1] = m2.run_padded(xg2)[0]
        with torch.cuda.stream(streams[3]):
            got[3] = dec1.decode_free(ins[0][:1].contiguous(), ins[1][:1].contiguous(), 30)[0]
    torch.cuda.synchronize()
    for g, a_ in zip(got, alone):
        assert torch.equal(g, a_)


@pytest.mark.gpu
def test_gpu_resident_paths_beside_a_long_running_kernel():
    """Every persistent path (bidirectional LSTM, teacher-forced chain + resident LSTM stack, free-running decode of one
    utterance, one-row and fold-batched WaveRNN) enqueued while another stream keeps every compute unit busy for a few hundred milliseconds
    (a queue of large matrix products): their workgroups become resident one by one, so a hand-over wait can last as long
    as the other stream's kernels do.  The waits are bounded in seconds with a back-off, not in milliseconds: the calls
    finish with the bits they give on a quiet device."""
    from test_wavernn import FULL, make
    from wavernn_amd import Decoder
    W1, x1 = synth_case(64, 40, 512, 512, seed=670)
    m1 = module(W1, 512, 512).cuda()
    m1.set_path("resident")
    xg1 = torch.as_tensor(x1).cuda()
    dims = dict(n_mels=80, decoder_dims=256, lstm_dims=512, aligner_out_dims=32)
    dec = Decoder(**dims)
    dec.load_state_dict({k: torch.as_tensor(np.ascontiguousarray(v)) for k, v in synth.decoder_state_dict(**dims, seed=3).items()})
    dec = dec.cuda().eval()
    dec.set_tf_path("chain"), dec.set_free_path("resident")
    ins = [torch.as_tensor(np.ascontiguousarray(a)).cuda() for a in synth.decoder_inputs(64, 128, 80, seed=5)]
    eq1, ep1 = ins[0][:1].contiguous(), ins[1][:1].contiguous()
    voc, _ = make(FULL, seed=3)
    voc = voc.cuda()
    voc.set_path("resident")
    mu = torch.as_tensor(synth.normalish(9, "mu_busy", (1, 400, 80))).cuda()
    au = torch.as_tensor(synth.normalish(9, "au_busy", (1, 400, 128))).cuda()
    uu = torch.as_tensor(synth.uniform01(9, "uu_busy", 400).reshape(1, 400)).cuda()

    mu20, au20, uu20 = mu.repeat(20, 1, 1).contiguous(), au.repeat(20, 1, 1).contiguous(), uu.repeat(20, 1).contiguous()

    def calls():
        return (m1.run_padded(xg1)[0], dec.decode_teacher_forced(*ins)[0], dec.decode_free(eq1, ep1, 60)[0],
                voc.generate_loop(mu, au, uu), voc.generate_loop(mu20, au20, uu20))

    quiet = calls()
    torch.cuda.synchronize()
    a = torch.randn(8192, 8192, device="cuda")
    busy, side = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(busy):
        for _ in range(30):             # about 10 ms each: a few hundred milliseconds of full-chip work
            a @ a
    with torch.cuda.stream(side):
        loud = calls()
    torch.cuda.synchronize()
    for q, g in zip(quiet, loud):
        assert torch.isfinite(g).all() and torch.equal(q, g)


@pytest.mark.gpu
def test_gpu_cabi_lengths_are_clamped_on_both_paths():
    """Through the C ABI directly: lens outside [0, M] are clamped (0 = the row never runs: padding everywhere, zero final
    state), the same on the launch loop and on the persistent path."""
    from wavernn_amd import _lib
    B, M = 35, 11
    W, x = synth_case(B, M, 512, 512, seed=670)
    m = module(W, 512, 512).cuda()
    xg = torch.as_tensor(x).cuda()
    lens = torch.randint(1, M + 1, (B,), dtype=torch.int32)
    lens[0], lens[1], lens[2], lens[34] = 0, M + 5, -3, 0
    lg = lens.cuda()
    h = m._get_handle()
    res = {}
    for path in ("launches", "resident"):
        _lib.check(_lib.load().wrnn_bilstm_set_path(h, _lib.FREE_PATHS[path]), "set_path")
        out = torch.full((B, M, 1024), 7.0, device="cuda")
        h_n = torch.full((2, B, 512), 7.0, device="cuda")
        c_n = torch.full((2, B, 512), 7.0, device="cuda")
        _lib.check(_lib.load().wrnn_bilstm_forward(h, xg.data_ptr(), lg.data_ptr(), B, M, -2.5, out.data_ptr(), h_n.data_ptr(),
                                                   c_n.data_ptr(), torch.cuda.current_stream().cuda_stream), "forward")
        torch.cuda.synchronize()
        res[path] = (out, h_n, c_n)
        for row in (0, 2, 34):                      # clamped to 0 frames
            assert bool((out[row] == -2.5).all()) and bool((h_n[:, row] == 0).all()) and bool((c_n[:, row] == 0).all())
        full, _ = m.run_padded(xg[1:2].contiguous())   # lens[1] > M: clamped to M, the row runs all frames
        assert float((out[1] - full[0]).abs().max()) < 5e-6
    for a_, b_ in zip(res["launches"], res["resident"]):
        assert float((a_ - b_).abs().max()) < 5e-6


@pytest.mark.gpu
def test_gpu_path_selection():
    """'resident' refuses shapes it does not cover; 'auto' runs the launch loop for them."""
    W, x = synth_case(3, 5, 64, 36, seed=650)
    m = module(W, 64, 36).cuda()
    xg = torch.as_tensor(x).cuda()
    base = m.run_padded(xg)[0]
    m.set_path("resident")
    with pytest.raises(RuntimeError, match="resident path needs"):
        m.run_padded(xg)
    m.set_path("launches")
    assert torch.equal(m.run_padded(xg)[0], base)
    with pytest.raises(ValueError):
        m.set_path("chain")
    W, x = synth_case(65, 3, 512, 512, seed=651)
    m = module(W, 512, 512).cuda()
    m.set_path("resident")
    with pytest.raises(RuntimeError, match="resident path needs"):
        m.run_padded(torch.as_tensor(x).cuda())
    m.set_path("auto")
    assert torch.isfinite(m.run_padded(torch.as_tensor(x).cuda())[0]).all()


@pytest.mark.gpu
def test_gpu_packed_sequence_call_convention():
    """Called the way forward_tacotron.py:147-152 calls nn.LSTM; checked against torch's own CPU nn.LSTM."""
    B, M, in_dims, H = 6, 23, 64, 48
    W, x = synth_case(B, M, in_dims, H, seed=610)
    lens = torch.tensor([23, 4, 17, 1, 23, 9])
    ref = torch.nn.LSTM(in_dims, H, batch_first=True, bidirectional=True)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in W.items()})
    xt = torch.as_tensor(x)
    with torch.no_grad():
        ro, (rh, rc) = ref(pack_padded_sequence(xt, lens, batch_first=True, enforce_sorted=False))
        ro, _ = pad_packed_sequence(ro, padding_value=-11.5129, batch_first=True)
    m = module(W, in_dims, H).cuda()
    go, (gh, gc) = m(pack_padded_sequence(xt.cuda(), lens, batch_first=True, enforce_sorted=False))
    go, glens = pad_packed_sequence(go, padding_value=-11.5129, batch_first=True)
    assert glens.tolist() == lens.tolist()
    G.assert_close(go.cpu().numpy(), ro.numpy(), "go.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(gh.cpu().numpy(), rh.numpy(), "gh.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(gc.cpu().numpy(), rc.numpy(), "gc.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
def test_gpu_full_size_properties():
    """BASELINE config-4 shape after the length regulator (B=64, C=512, M~830): rows are independent
    (a permuted batch gives the permuted output bit for bit), outputs finite, plus an oracle slice."""
    B, M, in_dims, H = 64, 830, 512, 512
    W, x = synth_case(B, M, in_dims, H, seed=620)
    m = module(W, in_dims, H).cuda()
    xg = torch.as_tensor(x).cuda()
    lens = torch.as_tensor(M - (synth.uniform01(8, "lens", B) * 200).astype(np.int64))
    out, (h_n, _) = m.run_padded(xg, lens, 0.0)
    assert torch.isfinite(out).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    out_p, (h_p, _) = m.run_padded(xg[perm.cuda()], lens[perm], 0.0)
    assert torch.equal(out_p, out[perm.cuda()])
    assert torch.equal(h_p, h_n[:, perm.cuda()])
    rows = [0, 17, 63]
    ref_out, _, _ = O.bilstm(O.to_torch(W), torch.as_tensor(x[rows, :60]), None)
    sub, _ = m.run_padded(xg[rows, :60].contiguous())
    G.assert_close(sub[:, :, :H].cpu().numpy(), ref_out[:, :, :H].numpy(), "sub[:", atol_scale=1e-5)   # forward half is causal
    G.assert_close(sub.cpu().numpy(), ref_out.numpy(), "sub.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
def test_gpu_bad_lengths_raise():
    W, x = synth_case(2, 5, 32, 16, seed=630)
    m = module(W, 32, 16).cuda()
    with pytest.raises(ValueError):
        m.run_padded(torch.as_tensor(x).cuda(), torch.tensor([6, 2]))
    with pytest.raises(ValueError):
        m.run_padded(torch.as_tensor(x).cuda(), torch.tensor([0, 2]))
