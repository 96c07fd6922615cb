"""GPU parity of the HIP length regulator (through the C ABI): bit-exact (pure copy)."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import taco_oracle as O

pytestmark = pytest.mark.gpu


def T(a):
    return torch.as_tensor(np.ascontiguousarray(a))


def test_reference_golden_cases():
    from wavernn_amd import LengthRegulator, synth
    rec, g = G.load("lr_cases")
    lr = LengthRegulator()
    d0 = T(g["dur0"].copy()).cuda()
    y0 = lr(T(g["x0"]).cuda(), d0)
    assert np.array_equal(y0.cpu().numpy(), g["y0"])
    assert np.array_equal(d0.cpu().numpy(), g["dur0_after"])     # in-place clamp, common_layers.py:18
    c = rec["case1"]
    x1, _ = synth.lr_inputs(c["B"], c["T"], c["C"], seed=c["seed"], dur_lo=c["dur_lo"], dur_hi=c["dur_hi"])
    d1 = T(g["dur1"].copy()).cuda()
    y1 = lr(T(x1).cuda(), d1)
    assert np.array_equal(y1.cpu().numpy(), g["y1"])
    assert np.array_equal(d1.cpu().numpy(), g["dur1_after"])
    y2 = lr(T(g["x2"]).cuda(), T(g["dur2"].copy()).cuda())
    assert list(y2.shape) == list(g["y2_shape"])


@pytest.mark.parametrize("B,Tn,C,lo,hi", [(64, 128, 512, 1, 11), (64, 128, 256, 1, 11), (3, 1000, 768, 0, 3),
                                          (1, 1, 4, 5, 5), (5, 300, 512, 0, 0),
                                          (112, 128, 512, 1, 11)])  # > 160 MiB of output: the streaming tuning
def test_vs_oracle_seeded(B, Tn, C, lo, hi):
    from wavernn_amd import LengthRegulator, synth
    x, dur = synth.lr_inputs(B, Tn, C, seed=4, dur_lo=lo, dur_hi=hi)
    ref = O.length_regulate(T(x), T(dur.copy()))
    out = LengthRegulator()(T(x).cuda(), T(dur.copy()).cuda())
    assert tuple(out.shape) == tuple(ref.shape)
    assert torch.equal(out.cpu(), ref)


def test_fractional_and_noncontiguous():
    from wavernn_amd import LengthRegulator, synth
    x = T(synth.normalish(3, "x", (4, 50, 64)))
    dur = T(synth.uniform(3, "d", (4, 50), -1.0, 4.0))
    big = torch.zeros(4, 100)
    big[:, ::2] = dur
    view = big.cuda()[:, ::2]            # non-contiguous durations are clamped in place too
    ref_d = dur.clone()
    ref = O.length_regulate(x, ref_d)
    out = LengthRegulator()(x.cuda().transpose(0, 1).contiguous().transpose(0, 1), view)
    assert torch.equal(out.cpu(), ref)
    assert torch.equal(view.cpu(), ref_d)


def test_full_size_properties():
    """BASELINE config 4 shape (B=64, T=128, C=512, dur ~ U{1..11}): every output row equals the
    token it came from, padding is zero, per-item length = sum of durations."""
    from wavernn_amd import LengthRegulator, synth
    x, dur = synth.lr_inputs(64, 128, 512, seed=4)
    xg, dg = T(x).cuda(), T(dur).cuda()
    out = LengthRegulator()(xg, dg)
    lens = dg.long().sum(dim=1)
    assert out.shape[1] == int(lens.max())
    idx = torch.repeat_interleave(torch.arange(128, device="cuda").repeat(64, 1).reshape(-1), dg.long().reshape(-1))
    assert idx.numel() == int(lens.sum())
    for b in (0, 17, 63):
        n = int(lens[b])
        ib = torch.repeat_interleave(torch.arange(128, device="cuda"), dg[b].long())
        assert torch.equal(out[b, :n], xg[b, ib])
        assert not out[b, n:].any()
