"""Load tests/golden/*.npz and re-create the inputs their recipes describe."""
import glob
import json
import os

import numpy as np

from wavernn_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(kind_prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, kind_prefix + "*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    d = {k: z[k] for k in z.files if k != "recipe"}
    return json.loads(str(z["recipe"])), d


def weights(recipe):
    return synth.decoder_state_dict(**recipe["dims"], seed=recipe["seed"], peaky=recipe["peaky"],
                                    r=recipe["r"])


def inputs(recipe):
    """(enc_q, enc_p, mel_in) exactly as tools/make_golden.py drew them."""
    dims = recipe["dims"]
    kind = recipe["kind"]
    S = {"steps": 3 * recipe["r"] + 1, "loop": recipe.get("S"), "free": 1}[kind]
    return synth.decoder_inputs(recipe["B"], recipe["T"], S, dims["n_mels"], dims["decoder_dims"],
                                dims["aligner_out_dims"], seed=recipe["seed"] + 10)


def noise(recipe, n_steps):
    B, l = recipe["B"], recipe["dims"]["lstm_dims"]
    s = recipe["seed"] + 20
    return dict(du1=synth.uniform(s, "du1", (n_steps, B, synth.PRENET_FC1)),
                du2=synth.uniform(s, "du2", (n_steps, B, synth.PRENET_FC2)),
                zu1=synth.uniform(s, "zu1", (n_steps, B, l)),
                zu2=synth.uniform(s, "zu2", (n_steps, B, l)))


def rel_err(a, b):
    """max |a-b| / max(|b|_inf, tiny): the 'relative error' of BASELINE.json's north_star."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def assert_close(a, b, what="", rtol=1e-4, atol=1e-6, atol_scale=0.0, tol=1e-4):
    """The parity bar of the decoder path, both ways: max-norm relative error < tol (BASELINE.json's
    'within 1e-4 relative') AND element-wise |a - b| <= rtol * |b| + atol, so that small-magnitude entries
    (attention weights of 1e-3 next to a row maximum of 0.9) are held to the same relative bar.
    Softmax outputs (scores, attention state) use the plain atol = 1e-6; signed sums with cancellation (mel
    frames, hidden states: an fp32 dot product's absolute error follows the size of its terms, not of its
    result) pass atol_scale = 1e-5, i.e. atol = 1e-5 * max|b|, still ten times tighter than the max-norm bar."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = rel_err(a, b)
    assert err < tol, (what, "max-norm relative error", err)
    atol = max(atol, atol_scale * float(np.abs(b).max()) if b.size else 0.0)
    bad = np.abs(a - b) > rtol * np.abs(b) + atol
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(a - b) - rtol * np.abs(b)), a.shape)
        raise AssertionError((what, "element-wise", int(bad.sum()), "of", a.size, "worst at", i, float(a[i]), float(b[i])))
