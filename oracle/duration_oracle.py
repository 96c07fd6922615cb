"""CPU oracle for the attention -> duration extraction.

TEST INFRASTRUCTURE ONLY: imported by tests/ (and bench.py's cpu-baseline leg), never by the product
path.  Pinned: against the reference's own known answers (tests/test_duration_extractor.py:21-57) and
against tests/golden/durations_cases.npz, written by running the reference DurationExtractor
(tools/make_golden.py::case_durations).

`extract_dijkstra` restates duration_extraction/duration_extractor.py:23-130 with the same third-party
routine the reference calls, scipy.sparse.csgraph.dijkstra (scipy 1.15.3 in this image; the reference
pins no version): identical graph (same CSR), so identical distances, predecessors and tie choices.
`extract_dp` is the dynamic-programme form the HIP kernel implements: identical distances; on ties it
prefers down-right, then down, then right, which need not be scipy's heap-order choice.
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import dijkstra

SILENT_IDS = tuple(range(11))  # utils/text/symbols.py:27: indices of '_' + "!'(),.:;? " in `phonemes`


def _prepare(x, mel, att, thr, shift, silent_ids):
    """:36-53 -> (clamped attention [mel_len rows used], silence mask, weights float32)."""
    att = np.array(att, dtype=np.float32, copy=True)
    mel = np.asarray(mel, dtype=np.float32)
    mel_len = mel.shape[-1]
    sil = mel.mean(axis=0, dtype=np.float32) < np.float32(thr)
    idx = np.nonzero(sil)[0]
    if idx.size >= 2:   # a single silent frame squeezes to a 0-dim tensor and is skipped (:43-44)
        stok = np.isin(np.asarray(x), np.asarray(silent_ids)).astype(np.float32)
        att_shift = stok * np.float32(shift) * np.float32(2) - np.float32(shift)
        att[idx, :] = att[idx, :] + att_shift
    att = np.clip(att, np.float32(0.0), np.float32(1.0))
    w = (np.float32(1.0) - att[:mel_len, :]).astype(np.float32)
    return att, sil, w


def _collect(path, att, sil, n_tok):
    """:66-84: last column per row -> durations; mean attention along the path over voiced frames."""
    mel_text = {}
    scores = []
    for i, j in path:
        mel_text[i] = j
        if not sil[i]:
            scores.append(float(att[i, j]))
    dur = np.zeros(n_tok, dtype=np.float32)
    for j in mel_text.values():
        dur[j] += 1
    return dur, sum(scores) / len(scores)


def extract_dijkstra(x, mel, att, thr, shift, silent_ids=SILENT_IDS):
    att, sil, w = _prepare(x, mel, att, thr, shift, silent_ids)
    S, T = w.shape
    node = np.arange(S * T).reshape(S, T)
    rows = [node[:, :-1].ravel(), node[:-1, :].ravel(), node[:-1, :-1].ravel()]          # :108-127
    cols = [node[:, 1:].ravel(), node[1:, :].ravel(), node[1:, 1:].ravel()]
    data = [w[:, 1:].ravel(), w[1:, :].ravel(), w[1:, 1:].ravel()]
    adj = coo_matrix((np.concatenate(data), (np.concatenate(rows), np.concatenate(cols))),
                     shape=(S * T, S * T)).tocsr()
    adj.sort_indices()
    dist, pred = dijkstra(csgraph=adj, directed=True, indices=0, return_predecessors=True)
    path = []
    p = pred[-1]
    while p != 0:
        path.append(p)
        p = pred[p]
    path.reverse()
    path = [0] + path + [dist.size - 1]
    dur, score = _collect([divmod(int(n), T) for n in path], att, sil, len(x))
    return dur, score, float(dist[-1])


def extract_dp(x, mel, att, thr, shift, silent_ids=SILENT_IDS):
    att, sil, w32 = _prepare(x, mel, att, thr, shift, silent_ids)
    w = w32.astype(np.float64)
    S, T = w.shape
    D = np.full((S, T), np.inf)
    P = np.zeros((S, T), dtype=np.int8)
    D[0, 0] = 0.0
    P[0, 0] = 3
    for i in range(S):
        for j in range(T):
            if i == 0 and j == 0:
                continue
            best, code = (D[i - 1, j - 1], 2) if (i > 0 and j > 0) else (np.inf, 2)
            if i > 0 and D[i - 1, j] < best:
                best, code = D[i - 1, j], 1
            if j > 0 and D[i, j - 1] < best:
                best, code = D[i, j - 1], 0
            D[i, j] = best + w[i, j]
            P[i, j] = code
    i, j = S - 1, T - 1
    path = [(i, j)]
    while P[i, j] != 3:
        c = P[i, j]
        if c == 2:
            i, j = i - 1, j - 1
        elif c == 1:
            i -= 1
        else:
            j -= 1
        path.append((i, j))
    path.reverse()
    dur, score = _collect(path, att, sil, len(x))
    return dur, score, float(D[-1, -1])
