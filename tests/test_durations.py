"""attention -> durations (duration_extraction/duration_extractor.py:23-130).

Pinned by the reference's own known answers (tests/test_duration_extractor.py:21-57, restated below)
and by tests/golden/durations_cases.npz (outputs of the reference DurationExtractor).  Parity statement:
path cost bit-identical to scipy's Dijkstra on every input; durations identical wherever the shortest
path is unique and on the reference's known answers; on exact ties the HIP path follows
oracle.extract_dp's documented preference (scipy's tie choice is heap-order dependent)."""
import numpy as np
import pytest
import torch

import golden_util as G
from oracle import duration_oracle as DO
from wavernn_amd import synth


def cases():
    rec, d = G.load("durations_cases")
    for n, meta in enumerate(rec["cases"]):
        yield n, meta, {k: d[f"c{n}_{k}"] for k in ("x", "mel", "att", "dur", "score")}


CASES = list(cases())


def diagonal_attention(S, T):  # the reference test's helper (:9-13)
    att = np.zeros((S, T), dtype=np.float32)
    for i in range(S):
        att[i, i // 2] = 1
    return att


def known_answers():
    """(x, mel, att, thr, shift, expected durations) of tests/test_duration_extractor.py."""
    x = np.array([15, 16, 10, 17, 18])
    mel = np.full((80, 10), -10.0, dtype=np.float32)
    yield x, mel, diagonal_attention(10, 5), -11.0, 0.0, [2, 2, 2, 2, 2]
    mel2 = mel.copy()
    mel2[:, 4:6] = -11.51
    att = diagonal_attention(10, 5)
    att[3:5, :] = 1.0 / 5
    yield x, mel2, att, -11.0, 0.0, [2, 3, 1, 2, 2]
    yield x, mel2, att, -11.0, 0.25, [2, 2, 2, 2, 2]


# ------------------------------------------------------------------ CPU: oracle pinned
@pytest.mark.parametrize("fn", [DO.extract_dijkstra, DO.extract_dp])
def test_reference_known_answers(fn):
    for x, mel, att, thr, shift, want in known_answers():
        dur, score, _ = fn(x, mel, att, thr, shift)
        assert dur.tolist() == want
        assert 0.0 <= score <= 1.0


@pytest.mark.parametrize("n,meta,c", CASES, ids=[f"{m['kind']}{n}" for n, m, _ in CASES])
def test_oracle_dijkstra_vs_reference(n, meta, c):
    dur, score, _ = DO.extract_dijkstra(c["x"], c["mel"], c["att"], meta["thr"], meta["shift"])
    assert dur.tolist() == c["dur"].tolist()          # ties included: same scipy routine, same graph
    assert score == pytest.approx(float(c["score"]), abs=1e-12)


@pytest.mark.parametrize("n,meta,c", CASES, ids=[f"{m['kind']}{n}" for n, m, _ in CASES])
def test_oracle_dp_vs_reference(n, meta, c):
    dur, score, cost = DO.extract_dp(c["x"], c["mel"], c["att"], meta["thr"], meta["shift"])
    _, _, cost_ref = DO.extract_dijkstra(c["x"], c["mel"], c["att"], meta["thr"], meta["shift"])
    assert cost == cost_ref                           # bit-identical float64 distance
    assert dur.sum() == meta["S"]
    if meta["kind"] in ("unique", "onesil"):
        assert dur.tolist() == c["dur"].tolist()
        assert score == pytest.approx(float(c["score"]), abs=1e-12)


def test_cpu_call_raises():
    from wavernn_amd.duration_extractor import DurationExtractor
    ex = DurationExtractor(-11.0, 0.25)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ex.extract_batch(torch.zeros(1, 4, 2), torch.zeros(1, 80, 4), torch.zeros(1, 2, dtype=torch.long),
                         torch.tensor([4]), torch.tensor([2]))


def test_attention_file_format(tmp_path):
    """att_pred/<id>.npy as duration_extraction_pipe.py:118-125 writes it: float32 [mel_len, x_len]."""
    from wavernn_amd.duration_pipe import save_attention_batch
    att = torch.rand(2, 7, 5)
    save_attention_batch(att, torch.tensor([5, 3]), torch.tensor([7, 4]), ["a", "b"], tmp_path)
    a, b = np.load(tmp_path / "a.npy", allow_pickle=False), np.load(tmp_path / "b.npy", allow_pickle=False)
    assert a.dtype == np.float32 and a.shape == (7, 5) and b.shape == (4, 3)
    assert np.array_equal(b, att[1, :4, :3].numpy())


# ------------------------------------------------------------------ GPU parity, through the C ABI
def gpu_single(ex, x, mel, att):
    return ex(x=torch.as_tensor(x), mel=torch.as_tensor(mel), attention=torch.as_tensor(att))


@pytest.mark.gpu
def test_gpu_reference_known_answers():
    from wavernn_amd.duration_extractor import DurationExtractor
    for x, mel, att, thr, shift, want in known_answers():
        att_in = torch.as_tensor(att.copy())
        dur, score = DurationExtractor(thr, shift)(x=torch.as_tensor(x), mel=torch.as_tensor(mel), attention=att_in)
        assert dur.tolist() == want
        assert isinstance(score, float)
        assert score == pytest.approx(DO.extract_dp(x, mel, att, thr, shift)[1], abs=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,meta,c", CASES, ids=[f"{m['kind']}{n}" for n, m, _ in CASES])
def test_gpu_vs_golden_and_dp(n, meta, c):
    from wavernn_amd.duration_extractor import DurationExtractor
    ex = DurationExtractor(meta["thr"], meta["shift"])
    dur, score, cost = ex.extract_batch(torch.as_tensor(c["att"]).cuda()[None], torch.as_tensor(c["mel"]).cuda()[None],
                                        torch.as_tensor(c["x"]).cuda()[None], torch.tensor([meta["S"]]),
                                        torch.tensor([meta["T"]]), return_cost=True)
    d_dp, s_dp, c_dp = DO.extract_dp(c["x"], c["mel"], c["att"], meta["thr"], meta["shift"])
    assert dur[0].cpu().tolist() == d_dp.tolist()                      # integer work: exact
    assert float(cost[0]) == c_dp == DO.extract_dijkstra(c["x"], c["mel"], c["att"], meta["thr"], meta["shift"])[2]
    assert float(score[0]) == pytest.approx(s_dp, abs=1e-12)
    if meta["kind"] in ("unique", "onesil"):
        assert dur[0].cpu().tolist() == c["dur"].tolist()              # the reference's own output
        assert float(score[0]) == pytest.approx(float(c["score"]), abs=1e-12)


@pytest.mark.gpu
def test_gpu_ragged_batch_vs_dp():
    """Padded batch with ragged mel / token lengths, silent stretches and the probability shift."""
    from wavernn_amd.duration_extractor import DurationExtractor
    B, S, T = 9, 70, 21
    rng = np.random.default_rng(5)
    att = torch.softmax(torch.as_tensor(synth.normalish(9, "att", (B, S, T))) * 4, dim=2).numpy()
    mel = synth.uniform(9, "mel", (B, 80, S), -12.0, -9.5)
    tok = rng.integers(0, 40, size=(B, T))
    ml = rng.integers(2, S + 1, size=B)
    xl = rng.integers(1, T + 1, size=B)
    ml[0], xl[0], ml[1], xl[1] = S, T, 1, 1
    ex = DurationExtractor(-10.7, 0.25)
    dur, score, cost = ex.extract_batch(torch.as_tensor(att).cuda(), torch.as_tensor(mel).cuda(), torch.as_tensor(tok).cuda(),
                                        torch.as_tensor(ml), torch.as_tensor(xl), return_cost=True)
    dur, score, cost = dur.cpu().numpy(), score.cpu().numpy(), cost.cpu().numpy()
    for b in range(B):
        s, t = int(ml[b]), int(xl[b])
        if s * t == 1:
            assert dur[b, 0] == 1 and dur[b, 1:].sum() == 0   # the reference itself cannot run a 1x1 matrix
            continue
        try:
            d_dp, s_dp, c_dp = DO.extract_dp(tok[b, :t], mel[b, :, :s], att[b, :s, :t], -10.7, 0.25)
        except ZeroDivisionError:       # every frame silent: the reference raises, the batch entry point flags NaN
            assert np.isnan(score[b])
            continue
        assert dur[b, :t].tolist() == d_dp.tolist(), b
        assert dur[b, t:].sum() == 0
        assert cost[b] == c_dp
        assert score[b] == pytest.approx(s_dp, abs=1e-12)


@pytest.mark.gpu
def test_gpu_full_size_from_decoder_shapes():
    """B=64 LJSpeech-length utterances (S=640 frames, T=128 tokens, the bench workload's attention
    shape): frames are conserved, nothing leaks past x_len, cost equals scipy's Dijkstra."""
    from wavernn_amd.duration_extractor import DurationExtractor
    B, S, T = 64, 640, 128
    centre = np.linspace(0, T - 1, S)[None, :, None]
    logits = -0.5 * ((np.arange(T)[None, None, :] - centre) / 1.5) ** 2 + synth.normalish(11, "jit", (B, S, T)) * 0.7
    att = torch.softmax(torch.as_tensor(logits, dtype=torch.float32), dim=2)
    mel = torch.as_tensor(synth.uniform(11, "mel", (B, 80, S), -12.0, -8.0))
    tok = torch.as_tensor((synth.uniform01(11, "tok", B * T) * 60).astype(np.int64).reshape(B, T))
    ml = torch.as_tensor(S - (synth.uniform01(12, "ml", B) * 200).astype(np.int64))
    xl = torch.as_tensor(T - (synth.uniform01(12, "xl", B) * 40).astype(np.int64))
    ex = DurationExtractor(-10.2, 0.25)
    dur, score, cost = ex.extract_batch(att.cuda(), mel.cuda(), tok.cuda(), ml, xl, return_cost=True)
    dur = dur.cpu()
    assert torch.equal(dur.sum(dim=1).long(), ml)
    for b in range(B):
        assert dur[b, int(xl[b]):].sum() == 0
    assert torch.isfinite(score).all() and (score >= 0).all() and (score <= 1).all()
    for b in (0, 31, 63):
        s, t = int(ml[b]), int(xl[b])
        d_ref, s_ref, c_ref = DO.extract_dijkstra(tok[b, :t].numpy(), mel[b, :, :s].numpy(), att[b, :s, :t].numpy(), -10.2, 0.25)
        assert float(cost[b]) == c_ref
        d_dp, s_dp, _ = DO.extract_dp(tok[b, :t].numpy(), mel[b, :, :s].numpy(), att[b, :s, :t].numpy(), -10.2, 0.25)
        assert dur[b, :t].tolist() == d_dp.tolist()
        assert float(score[b]) == pytest.approx(s_dp, abs=1e-12)


@pytest.mark.gpu
def test_gpu_all_silent_raises_like_reference():
    from wavernn_amd.duration_extractor import DurationExtractor
    x, mel, att, _, _, _ = next(known_answers())
    with pytest.raises(ZeroDivisionError):
        gpu_single(DurationExtractor(0.0, 0.25), x, mel, att)   # every frame below the threshold


@pytest.mark.gpu
def test_gpu_bad_lengths_raise():
    from wavernn_amd.duration_extractor import DurationExtractor
    ex = DurationExtractor(-11.0, 0.0)
    a, m, t = torch.zeros(1, 4, 3).cuda(), torch.zeros(1, 80, 4).cuda(), torch.zeros(1, 3, dtype=torch.long).cuda()
    with pytest.raises(ValueError):
        ex.extract_batch(a, m, t, torch.tensor([5]), torch.tensor([3]))
    with pytest.raises(ValueError):
        ex.extract_batch(a, m, t, torch.tensor([4]), torch.tensor([0]))


@pytest.mark.gpu
def test_gpu_duration_files(tmp_path):
    """alg/<id>.npy as duration_extraction_pipe.py:183 writes it, from a device-resident batch."""
    from wavernn_amd.duration_extractor import DurationExtractor
    from wavernn_amd.duration_pipe import extract_durations_batch
    x, mel, att, thr, shift, want = list(known_answers())[1]
    res = extract_durations_batch(DurationExtractor(thr, shift), torch.as_tensor(att).cuda()[None],
                                  torch.as_tensor(mel).cuda()[None], torch.as_tensor(x).cuda()[None],
                                  torch.tensor([10]), torch.tensor([5]), ["utt0"], tmp_path)
    saved = np.load(tmp_path / "utt0.npy", allow_pickle=False)
    assert saved.dtype.kind == "i" and saved.tolist() == want == res["utt0"][0].tolist()
