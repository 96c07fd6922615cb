"""The mixture-of-logistics sampler (reference utils/distribution.py:87-123) on the GPU, through the C ABI, against the
fixture that tools/make_golden.py produced by running the reference function itself with the same uniforms, and against
the pinned oracle restatement on a larger seeded input.  Bar: the mixture index bit-exact (an argmax of fp32 scores whose
logs come from different libms: a flip needs two scores within an ulp or two, which the seeded inputs do not hold),
the sample within 1e-6 absolute (outputs live in [-1, 1])."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import wavernn_oracle as WO
from wavernn_amd import synth


def test_cpu_call_raises():
    from wavernn_amd.distribution import sample_from_discretized_mix_logistic
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        sample_from_discretized_mix_logistic(torch.zeros(1, 30, 4))


@pytest.mark.gpu
def test_gpu_vs_reference_golden():
    from wavernn_amd.distribution import sample_from_discretized_mix_logistic
    rec, g = G.load("mol_sampler")
    y, um, ul = (torch.as_tensor(g[k]).cuda() for k in ("y", "u_mix", "u_log"))
    x, mix = sample_from_discretized_mix_logistic(y, noise=(um, ul), return_mixture=True)
    assert np.array_equal(mix.cpu().numpy(), g["mix"])
    assert np.abs(x.cpu().numpy().astype(np.float64) - g["x"]).max() <= 1e-6
    x7 = sample_from_discretized_mix_logistic(y, log_scale_min=-7.0, noise=(um, ul))
    assert np.abs(x7.cpu().numpy().astype(np.float64) - g["x_min7"]).max() <= 1e-6
    assert float(x.abs().max()) == 1.0                                   # the output clamp acted


@pytest.mark.gpu
@pytest.mark.parametrize("B,nr_mix,T", [(20, 10, 12100), (1, 10, 1), (3, 4, 1000), (2, 32, 77)])
def test_gpu_vs_oracle_seeded(B, nr_mix, T):
    from wavernn_amd.distribution import U_HI, U_LO, sample_from_discretized_mix_logistic
    y = synth.normalish(70, "y", (B, 3 * nr_mix, T))
    y[:, 2 * nr_mix:] = y[:, 2 * nr_mix:] * 1.5 - 3.0
    um = (U_LO + (U_HI - U_LO) * synth.uniform01(71, "um", B * T * nr_mix).astype(np.float64)).astype(np.float32).reshape(B, T, nr_mix)
    ul = (U_LO + (U_HI - U_LO) * synth.uniform01(72, "ul", B * T).astype(np.float64)).astype(np.float32).reshape(B, T)
    x_ref, mix_ref = WO.sample_mol(torch.as_tensor(y), torch.as_tensor(um), torch.as_tensor(ul))
    x, mix = sample_from_discretized_mix_logistic(torch.as_tensor(y).cuda(), noise=(torch.as_tensor(um).cuda(), torch.as_tensor(ul).cuda()),
                                                  return_mixture=True)
    assert torch.equal(mix.cpu().long(), mix_ref)
    assert float((x.cpu().double() - x_ref.double()).abs().max()) <= 1e-6
    drawn = sample_from_discretized_mix_logistic(torch.as_tensor(y).cuda())   # internal uniforms: in range, finite
    assert torch.isfinite(drawn).all() and float(drawn.abs().max()) <= 1.0
