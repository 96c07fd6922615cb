"""The C-ABI library loads and exports every symbol include/wrnn_mi355.h declares (no GPU needed)."""
import os
import re

import pytest

from wavernn_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "wrnn_mi355.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(wrnn_[a-z0-9_]+)\s*\(", src)))


def test_library_is_built():
    assert os.path.exists(build.LIB_PATH), "run `python -m wavernn_amd.build` (or __graft_entry__.build())"


def test_header_and_binding_agree():
    decl = _declared_symbols()
    assert decl, "no declarations parsed"
    assert sorted(_lib.SYMBOLS) == decl


def test_every_declared_symbol_is_exported():
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.wrnn_version() >= 100


def test_weight_struct_matches_reference_state_dict():
    from wavernn_amd import synth
    keys = set(synth.decoder_param_shapes())
    assert {k for _, k in _lib.WEIGHT_FIELDS} == keys


def test_errors_without_gpu_are_loud():
    import torch
    from wavernn_amd import Decoder, LengthRegulator
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    dec = Decoder(80, 256, 512, 32)
    B, T = 2, 5
    z = torch.zeros
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        dec(z(B, T, 256), z(B, T, 256), z(B, 112), (z(B, 256), z(B, 512), z(B, 512)), (z(B, 512), z(B, 512)),
            z(B, 256), 0)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        LengthRegulator()(z(1, 2, 4), z(1, 2))


def test_decoder_state_dict_keys_match_reference():
    # key list probed from the reference
    from wavernn_amd import Decoder, synth
    dec = Decoder(80, 256, 512, 32)
    sd = dec.state_dict()
    want = set(synth.decoder_param_shapes()) | {"r"}
    assert set(sd) == want
    for k, shape in synth.decoder_param_shapes().items():
        assert tuple(sd[k].shape) == tuple(shape), k
    assert Decoder.max_r == 20 and sd["r"].dtype.is_floating_point is False


def test_argument_validation_precedes_any_device_work():
    """Bad sizes and null tensors are rejected with a message before any HIP call (so this runs without a GPU)."""
    lib = _lib.load()
    assert lib.wrnn_aligner_distance(None, None, -1, 1, 1, 1, None, None) != 0
    assert b"negative size" in lib.wrnn_last_error()
    assert lib.wrnn_aligner_distance(None, None, 1, 1, 1, 4, None, None) != 0
    assert b"null pointer" in lib.wrnn_last_error()
    assert lib.wrnn_aligner_distance(None, None, 0, 5, 5, 4, None, None) == 0     # empty batch: nothing to do
    assert lib.wrnn_lr_expand(None, None, 1, 1, 6, None, 1, None) != 0
    assert b"multiple of 4" in lib.wrnn_last_error()
    assert lib.wrnn_lr_prepare(None, -1, 1, None, None, None) != 0
    assert lib.wrnn_lr_prepare(None, 0, 1, None, None, None) == 0
    assert lib.wrnn_bigru_forward(None, None, None, 1, 1, 0.0, None, None, None) != 0
    assert b"null bigru handle" in lib.wrnn_last_error()
    assert lib.wrnn_decoder_destroy(None) == 0 and lib.wrnn_bigru_destroy(None) == 0


def test_decoder_pickles_and_copies_without_its_handle():
    """torch.save(model) / spawn-based multiprocessing pickle the module: the load_state_dict hook is a module-level
    function (a local lambda in the module's __dict__ cannot be pickled) and the library handle never travels."""
    import copy
    import pickle
    import torch
    from wavernn_amd import Decoder, Tacotron
    dec = Decoder(80, 256, 512, 32)
    dec2 = pickle.loads(pickle.dumps(dec))
    assert dec2._handle is None and dec2._dirty
    assert all(torch.equal(a, b) for a, b in zip(dec.state_dict().values(), dec2.state_dict().values()))
    assert copy.deepcopy(dec)._handle is None
    tac = Tacotron(embed_dims=16, num_chars=20, encoder_dims=16, decoder_dims=256, n_mels=80, postnet_dims=16, encoder_k=2,
                   lstm_dims=32, postnet_k=2, num_highways=1, dropout=0.5, stop_threshold=-11.0, aligner_hidden_dims=16,
                   aligner_out_dims=32, speaker_emb_dim=0)
    pickle.loads(pickle.dumps(tac))
