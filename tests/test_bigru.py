"""the bidirectional GRU that ends the reference's CBHG block
(models/common_layers.py:89,123).  Fixtures: tensors captured with a forward hook while the reference CBHG
ran (tools/make_golden.py::case_cbhg_gru)."""
import numpy as np
import pytest
import torch
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

import golden_util as G
from oracle import taco_oracle as O
from wavernn_amd import synth

TOL = 1e-4  # north-star relative tolerance (fp32)


def gru_weights(rec):
    shapes = {"rnn." + k: tuple(v) for k, v in rec["rnn_shapes"].items()}
    # the fixture filled the whole CBHG state dict; regenerate it the same way and keep the rnn.* entries
    from wavernn_amd.tacotron_model import CBHG
    ref = CBHG(K=rec["K"], in_channels=rec["in_channels"], channels=rec["channels"], proj_channels=rec["proj"],
               num_highways=rec["highways"])
    all_shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items() if v.is_floating_point() and v.dim() > 0}
    assert {k: v for k, v in all_shapes.items() if k.startswith("rnn.")} == shapes
    sd = synth.fill_state_dict(all_shapes, seed=rec["seed"])
    return {k[len("rnn."):]: v for k, v in sd.items() if k.startswith("rnn.")}, sd


def module(W, in_dims, H):
    from wavernn_amd.bigru import BiGRU
    m = BiGRU(in_dims, H)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in W.items()})
    return m


def synth_case(B, M, in_dims, H, seed):
    shapes = {}
    for sfx in ("", "_reverse"):
        shapes["weight_ih_l0" + sfx] = (3 * H, in_dims)
        shapes["weight_hh_l0" + sfx] = (3 * H, H)
        shapes["bias_ih_l0" + sfx] = (3 * H,)
        shapes["bias_hh_l0" + sfx] = (3 * H,)
    W = synth.fill_state_dict(shapes, seed=seed)
    x = synth.normalish(seed, "bigru_x", (B, M, in_dims))
    return W, x


# ------------------------------------------------------------------ CPU: oracle vs reference fixtures
@pytest.mark.parametrize("name", G.names("cbhggru_"))
def test_oracle_vs_reference(name):
    rec, gold = G.load(name)
    W, _ = gru_weights(rec)
    out, h_n = O.bigru(O.to_torch(W), torch.as_tensor(gold["rnn_in"]))
    assert G.rel_err(out.numpy(), gold["rnn_out"]) < 1e-5
    assert G.rel_err(h_n.numpy(), gold["h_n"]) < 1e-5


def test_oracle_vs_torch_packed():
    B, M, in_dims, H = 4, 11, 12, 8
    W, x = synth_case(B, M, in_dims, H, seed=700)
    lens = torch.tensor([11, 3, 7, 1])
    ref = torch.nn.GRU(in_dims, H, batch_first=True, bidirectional=True)
    ref.load_state_dict({k: torch.as_tensor(v) for k, v in W.items()})
    with torch.no_grad():
        ro, rh = ref(pack_padded_sequence(torch.as_tensor(x), lens, batch_first=True, enforce_sorted=False))
        ro, _ = pad_packed_sequence(ro, padding_value=-3.0, batch_first=True)
    out, h_n = O.bigru(O.to_torch(W), torch.as_tensor(x), lens, -3.0)
    assert G.rel_err(out.numpy(), ro.numpy()) < 1e-5
    assert G.rel_err(h_n.numpy(), rh.numpy()) < 1e-5


def test_parameter_names_match_torch_gru():
    from wavernn_amd.bigru import BiGRU
    ref = torch.nn.GRU(32, 20, batch_first=True, bidirectional=True)
    mine = BiGRU(32, 20)
    assert {k: tuple(v.shape) for k, v in mine.state_dict().items()} == \
           {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    assert list(mine.state_dict()) == list(ref.state_dict())


def test_cpu_call_raises():
    from wavernn_amd.bigru import BiGRU
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        BiGRU(8, 4)(torch.zeros(1, 3, 8))


# ------------------------------------------------------------------ GPU parity, through the C ABI
@pytest.mark.gpu
@pytest.mark.parametrize("name", G.names("cbhggru_"))
def test_gpu_vs_reference_fixture(name):
    rec, gold = G.load(name)
    W, _ = gru_weights(rec)
    x = torch.as_tensor(gold["rnn_in"]).cuda()
    m = module(W, x.shape[2], rec["channels"]).cuda()
    out, h_n = m(x)
    G.assert_close(out.cpu().numpy(), gold["rnn_out"], "out.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(h_n.cpu().numpy(), gold["h_n"], "h_n.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", G.names("cbhggru_"))
def test_gpu_whole_cbhg_vs_reference_fixture(name):
    """The CBHG mirror with the HIP recurrence inside against the reference CBHG's output."""
    from wavernn_amd.tacotron_model import CBHG
    rec, gold = G.load(name)
    _, sd = gru_weights(rec)
    m = CBHG(K=rec["K"], in_channels=rec["in_channels"], channels=rec["channels"], proj_channels=rec["proj"],
             num_highways=rec["highways"])
    m.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()}, strict=False)
    m = m.cuda().eval()
    x = torch.as_tensor(synth.normalish(rec["seed"], "cbhg_x", (rec["B"], rec["in_channels"], rec["M"]))).cuda()
    with torch.no_grad():
        out = m(x)
    G.assert_close(out.cpu().numpy(), gold["rnn_out"], "out.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("B,M,in_dims,H,ragged", [(32, 60, 128, 128, False), (1, 7, 64, 36, False),
                                                  (33, 19, 128, 256, True), (5, 1, 32, 16, False),
                                                  # the weights-in-registers path (H % 32 == 0, H <= 128): ragged rows,
                                                  # chunk tails (M % 16 != 0) and whole chunks
                                                  (9, 37, 64, 64, True), (3, 16, 32, 32, False), (4, 50, 96, 96, True),
                                                  (130, 33, 128, 128, True),
                                                  # a pair of workgroups per (row, direction) (H = 192, 256): ragged rows, more
                                                  # rows than one launch holds (64), odd and even frame counts
                                                  (64, 50, 256, 256, True), (5, 33, 256, 256, False), (70, 21, 128, 256, True),
                                                  (9, 17, 64, 192, True), (1, 1, 256, 256, False)])
def test_gpu_vs_oracle(B, M, in_dims, H, ragged):
    W, x = synth_case(B, M, in_dims, H, seed=700 + B)
    lens = None
    if ragged:
        lens = 1 + (synth.uniform01(7, "lens", B) * M).astype(np.int64)
        lens[0], lens[-1] = M, 1
    ref_out, ref_h = O.bigru(O.to_torch(W), torch.as_tensor(x), lens, -11.5129)
    m = module(W, in_dims, H).cuda()
    out, h_n = m.run_padded(torch.as_tensor(x).cuda(), None if lens is None else torch.as_tensor(lens), -11.5129)
    G.assert_close(out.cpu().numpy(), ref_out.numpy(), "out.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(h_n.cpu().numpy(), ref_h.numpy(), "h_n.cpu().numpy()", atol_scale=1e-5)


@pytest.mark.gpu
def test_gpu_full_size_properties():
    """The postnet recurrence at the BASELINE decoder shape (B=64, S=640 frames, 128 -> 128): rows are
    independent (a permuted batch gives the permuted output bit for bit), outputs finite, plus an oracle slice."""
    B, M, in_dims, H = 64, 640, 128, 128
    W, x = synth_case(B, M, in_dims, H, seed=720)
    m = module(W, in_dims, H).cuda()
    xg = torch.as_tensor(x).cuda()
    out, h_n = m(xg)
    assert torch.isfinite(out).all()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1))
    out_p, h_p = m(xg[perm.cuda()])
    assert torch.equal(out_p, out[perm.cuda()])
    assert torch.equal(h_p, h_n[:, perm.cuda()])
    rows = [0, 17, 63]
    ref_out, _ = O.bigru(O.to_torch(W), torch.as_tensor(x[rows, :60]), None)
    sub, _ = m(xg[rows, :60].contiguous())
    G.assert_close(sub.cpu().numpy(), ref_out.numpy(), "sub.cpu().numpy()", atol_scale=1e-5)
    G.assert_close(out[rows, :60, :H].cpu().numpy(), ref_out[:, :, :H].numpy(), "out[rows", atol_scale=1e-5)   # forward half is causal


@pytest.mark.gpu
def test_gpu_empty_inputs():
    W, _ = synth_case(1, 1, 32, 16, seed=730)
    m = module(W, 32, 16).cuda()
    out, h_n = m(torch.zeros(0, 4, 32).cuda())
    assert out.shape == (0, 4, 32) and h_n.shape == (2, 0, 16)
    out, h_n = m(torch.zeros(3, 0, 32).cuda())
    assert out.shape == (3, 0, 32) and torch.equal(h_n, torch.zeros(2, 3, 16).cuda())
