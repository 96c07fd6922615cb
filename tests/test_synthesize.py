"""BASELINE configs[3]: ForwardTacotron forward + WaveRNN vocode end to end for 1,000 synthetic utterances sharded by
utterance over 8 ranks (reference per-sentence loop: gen_forward.py:88-107; the vocoder is build-defined, PARITY
UNPINNED).  The host logic (slices, noise, gather) is checked on CPU with stand-in models; on
the GPU the eight shards of the full workload run one after the other on the one card and a sample of utterances
must be bit-equal to the single-utterance path (tts.generate -> WaveRNN.generate under the same uniforms)."""
import numpy as np
import pytest
import torch

from wavernn_amd import sharding, synth
from wavernn_amd import synthesize as SY


class _FakeTTS(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(1))

    def generate(self, x, alpha=1.0):
        M = 3 * x.shape[1]
        return {'mel_post': (x.float().repeat_interleave(3, dim=1)[:, None, :] * torch.ones(1, 4, 1))[:, :, :M]}


class _FakeVoc:
    hop_length = 4

    def generate_batch(self, mels, target, overlap, noise, max_rows):
        return [np.full(((m.shape[2] - 1) * 4,), float(n.sum())) for m, n in zip(mels, noise)]


def test_shards_cover_every_utterance_once_and_noise_is_per_utterance():
    toks = [torch.as_tensor(t) for t in synth.utterance_tokens(37)]
    whole = SY.synthesize(_FakeTTS(), _FakeVoc(), toks, target=20, overlap=2, seed=3)
    parts = []
    for r in range(8):
        lo, hi = sharding.my_slice(len(toks), r, 8)
        parts += SY.synthesize(_FakeTTS(), _FakeVoc(), toks[lo:hi], first_index=lo, target=20, overlap=2, seed=3)
    assert len(parts) == len(whole) == 37
    for a, b in zip(parts, whole):
        assert np.array_equal(a['wav'], b['wav']) and torch.equal(a['mel_post'], b['mel_post'])
    n0 = SY.utterance_noise(_FakeVoc(), 36, 5, 3, 20, 2)
    assert n0.shape == (7, 24) and torch.equal(n0, SY.utterance_noise(_FakeVoc(), 36, 5, 3, 20, 2))
    assert not torch.equal(n0, SY.utterance_noise(_FakeVoc(), 36, 6, 3, 20, 2))


def build_models():
    from wavernn_amd import ForwardTacotron, WaveRNN
    torch.manual_seed(0)
    ft = ForwardTacotron(**synth.FORWARD_TACOTRON_CFG).eval()
    with torch.no_grad():
        for p in ft.dur_pred.parameters():
            p.mul_(0.02)                      # durations = 6 frames per token, exactly, for every utterance:
        ft.dur_pred.lin.bias.fill_(6.0)       # few distinct mel lengths, so MIOpen searches few conv shapes
    voc = WaveRNN(**synth.WAVERNN_CFG).eval()
    return ft, voc


@pytest.mark.gpu
def test_config3_1k_utterances_sharded_over_8():
    ft, voc = build_models()
    ft, voc = ft.cuda(), voc.cuda()
    toks = [torch.as_tensor(t) for t in synth.utterance_tokens(1000)]
    wavs = []
    for r in range(8):                         # the 8 ranks' shards, one after the other on the one GPU
        lo, hi = sharding.my_slice(1000, r, 8)
        assert hi - lo == 125
        wavs += [u['wav'] for u in SY.synthesize(ft, voc, toks[lo:hi], first_index=lo, seed=11, max_rows=192)]
    assert len(wavs) == 1000
    for i, w in enumerate(wavs):
        assert w.shape == ((6 * len(toks[i]) - 1) * 256,) and np.isfinite(w).all() and np.abs(w).max() <= 1.5
    for i in (0, 1, 2, 124, 125, 501, 873, 999):      # shard boundaries included
        g = ft.generate(x=toks[i].cuda()[None])
        noise = SY.utterance_noise(voc, g['mel_post'].shape[2], i, 11, 11000, 550)
        one = voc.generate(g['mel_post'], batched=True, target=11000, overlap=550, noise=noise)
        assert np.array_equal(one, wavs[i]), i
