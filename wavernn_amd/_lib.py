"""ctypes binding of libwrnn_mi355.so (C ABI declared in include/wrnn_mi355.h).

There is no CPU fallback: if the library is missing or a call fails, a RuntimeError is
raised.  Build the library with ``python -m wavernn_amd.build`` (or
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os

from .build import LIB_PATH

c_float_p = C.c_void_p  # raw device/host addresses are passed as integers


class DecoderDims(C.Structure):
    _fields_ = [("n_mels", C.c_int32), ("decoder_dims", C.c_int32),
                ("lstm_dims", C.c_int32), ("aligner_out_dims", C.c_int32)]


# order == field order of wrnn_decoder_weights; value == reference state_dict key
WEIGHT_FIELDS = [
    ("prenet_fc1_w", "prenet.fc1.weight"), ("prenet_fc1_b", "prenet.fc1.bias"),
    ("prenet_fc2_w", "prenet.fc2.weight"), ("prenet_fc2_b", "prenet.fc2.bias"),
    ("lsa_conv_w", "attn_net.conv.weight"), ("lsa_L_w", "attn_net.L.weight"),
    ("lsa_L_b", "attn_net.L.bias"), ("lsa_W_w", "attn_net.W.weight"),
    ("lsa_W_b", "attn_net.W.bias"), ("lsa_v_w", "attn_net.v.weight"),
    ("gru_w_ih", "attn_rnn.weight_ih"), ("gru_w_hh", "attn_rnn.weight_hh"),
    ("gru_b_ih", "attn_rnn.bias_ih"), ("gru_b_hh", "attn_rnn.bias_hh"),
    ("rnn_input_w", "rnn_input.weight"), ("rnn_input_b", "rnn_input.bias"),
    ("lstm1_w_ih", "res_rnn1.weight_ih"), ("lstm1_w_hh", "res_rnn1.weight_hh"),
    ("lstm1_b_ih", "res_rnn1.bias_ih"), ("lstm1_b_hh", "res_rnn1.bias_hh"),
    ("lstm2_w_ih", "res_rnn2.weight_ih"), ("lstm2_w_hh", "res_rnn2.weight_hh"),
    ("lstm2_b_ih", "res_rnn2.bias_ih"), ("lstm2_b_hh", "res_rnn2.bias_hh"),
    ("mel_proj_w", "mel_proj.weight"),
]


class DecoderWeights(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in WEIGHT_FIELDS]


class DecoderState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("attn_h", "h1", "h2", "c1", "c2", "context", "lsa_cum", "lsa_att")]


# order == field order of wrnn_bilstm_weights; value == torch.nn.LSTM parameter name
BILSTM_FIELDS = [
    ("w_ih", "weight_ih_l0"), ("w_hh", "weight_hh_l0"), ("b_ih", "bias_ih_l0"), ("b_hh", "bias_hh_l0"),
    ("w_ih_reverse", "weight_ih_l0_reverse"), ("w_hh_reverse", "weight_hh_l0_reverse"),
    ("b_ih_reverse", "bias_ih_l0_reverse"), ("b_hh_reverse", "bias_hh_l0_reverse"),
]


class BiLstmWeights(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in BILSTM_FIELDS]


# torch.nn.GRU has the same parameter names; field order of wrnn_bigru_weights
BIGRU_FIELDS = BILSTM_FIELDS


class BiGruWeights(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in BIGRU_FIELDS]


class WaveRnnDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("rnn_dims", "fc_dims", "feat_dims", "aux_dims", "n_classes", "mode")]


WAVERNN_MODES = {"RAW": 0, "MOL": 1}   # WRNN_WAVERNN_RAW / WRNN_WAVERNN_MOL


# order == field order of wrnn_wavernn_weights; value == (module, parameter) of the WaveRNN mirror
WAVERNN_FIELDS = [
    ("I_w", "I.weight"), ("I_b", "I.bias"),
    ("rnn1_w_ih", "rnn1.weight_ih_l0"), ("rnn1_w_hh", "rnn1.weight_hh_l0"),
    ("rnn1_b_ih", "rnn1.bias_ih_l0"), ("rnn1_b_hh", "rnn1.bias_hh_l0"),
    ("rnn2_w_ih", "rnn2.weight_ih_l0"), ("rnn2_w_hh", "rnn2.weight_hh_l0"),
    ("rnn2_b_ih", "rnn2.bias_ih_l0"), ("rnn2_b_hh", "rnn2.bias_hh_l0"),
    ("fc1_w", "fc1.weight"), ("fc1_b", "fc1.bias"), ("fc2_w", "fc2.weight"), ("fc2_b", "fc2.bias"),
    ("fc3_w", "fc3.weight"), ("fc3_b", "fc3.bias"),
]


class WaveRnnWeights(C.Structure):
    _fields_ = [(name, C.c_void_p) for name, _ in WAVERNN_FIELDS]


# every symbol include/wrnn_mi355.h declares: name -> (restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
SYMBOLS = {
    "wrnn_last_error": (C.c_char_p, []),
    "wrnn_version": (_I, []),
    "wrnn_decoder_create": (_I, [C.POINTER(DecoderDims), C.POINTER(DecoderWeights), _I, C.POINTER(_P)]),
    "wrnn_decoder_destroy": (_I, [_P]),
    "wrnn_decoder_step": (_I, [_P, _P, _P, _P, C.POINTER(DecoderState), C.POINTER(DecoderState),
                               _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "wrnn_decoder_run_teacher_forced": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "wrnn_decoder_run_free": (_I, [_P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _P, _P, _P,
                                   C.POINTER(_I), _P]),
    "wrnn_decoder_set_free_path": (_I, [_P, _I]),
    "wrnn_decoder_set_tf_path": (_I, [_P, _I]),
    "wrnn_decoder_set_profiling": (_I, [_P, _I]),
    "wrnn_decoder_get_profile": (_I, [_P, C.POINTER(C.c_float), C.POINTER(_I)]),
    "wrnn_lr_prepare": (_I, [_P, _I, _I, _P, _P, _P]),
    "wrnn_lr_expand": (_I, [_P, _P, _I, _I, _I, _P, _I, _P]),
    "wrnn_lr_regulate_static": (_I, [_P, _P, _I, _I, _I, _P, _I, _P, _P]),
    "wrnn_aligner_distance": (_I, [_P, _P, _I, _I, _I, _I, _P, _P]),
    "wrnn_durations_workspace_bytes": (C.c_size_t, [_I, _I, _I]),
    "wrnn_durations_extract": (_I, [_P, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _I, _I, _P, _I, _F, _F, _P, _P, _I,
                                    _P, _P, _P]),
    "wrnn_wavernn_create": (_I, [C.POINTER(WaveRnnDims), C.POINTER(WaveRnnWeights), _I, C.POINTER(_P)]),
    "wrnn_wavernn_destroy": (_I, [_P]),
    "wrnn_wavernn_generate": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "wrnn_wavernn_set_path": (_I, [_P, _I]),
    "wrnn_mol_sample": (_I, [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P]),
    "wrnn_bilstm_create": (_I, [_I, _I, C.POINTER(BiLstmWeights), _I, C.POINTER(_P)]),
    "wrnn_bilstm_destroy": (_I, [_P]),
    "wrnn_bilstm_forward": (_I, [_P, _P, _P, _I, _I, _F, _P, _P, _P, _P]),
    "wrnn_bilstm_set_path": (_I, [_P, _I]),
    "wrnn_bigru_create": (_I, [_I, _I, C.POINTER(BiGruWeights), _I, C.POINTER(_P)]),
    "wrnn_bigru_destroy": (_I, [_P]),
    "wrnn_bigru_forward": (_I, [_P, _P, _P, _I, _I, _F, _P, _P, _P]),
}

FREE_PATHS = {"auto": 0, "launches": 1, "resident": 2}   # WRNN_FREE_PATH_* and WRNN_WAVERNN_PATH_*
TF_PATHS = {"auto": 0, "launches": 1, "chain": 2, "chain_launch_stack": 3}         # WRNN_TF_PATH_*
# (tick_query has launches only under WRNN_SEPARATE_QUERY_LAUNCH=1: the projection rides in the GRU launch otherwise)
PROFILE_KINDS = ["prenet_hoisted", "tick_gru+query+lstm1", "tick_query", "tick_attention+rnn_input+lstm2", "mel_hoisted",
                 "chain_gru+query+attention", "lstm_stack", "rnn_input_chunks"]

_lib = None


def lib_path() -> str:
    return os.environ.get("WRNN_MI355_LIB", LIB_PATH)


def load():
    """Load the shared library (once) and type every entry point.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64.so.7; it must be the one HIP runtime of the
    # process, so it is loaded before the library (whose NEEDED entry then resolves to it).
    import torch  # noqa: F401
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"libwrnn_mi355.so not found at {path}: build it with `python -m wavernn_amd.build`. "
            "wavernn_amd has no CPU or PyTorch fallback for its HIP kernels.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().wrnn_last_error()
        raise RuntimeError(f"{what} failed: {msg.decode() if msg else 'unknown error'}")
