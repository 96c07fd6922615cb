"""wavernn_amd: MI355X-native (gfx950 HIP) hot path of as-ideas/WaveRNN (ForwardTacotron).

Only the autoregressive Tacotron decoder loop, the length-regulator expand and the callers either side
of them (SURVEY.md section 8, rows a-e and f1-f4) live here; everything runs through libwrnn_mi355.so (include/wrnn_mi355.h).
Importing the package does not need a GPU; calling an op without one, or without the built
library, raises.
"""
__all__ = ["Decoder", "LengthRegulator", "Tacotron", "BiLSTM", "BiGRU", "DurationExtractor", "ForwardTacotron", "FastPitch", "MultiFastPitch", "MultiForwardTacotron", "WaveRNN", "synth"]


def __getattr__(name):  # lazy: `import wavernn_amd.synth` must not need torch
    if name == "Decoder":
        from .tacotron import Decoder
        return Decoder
    if name == "Tacotron":
        from .tacotron_model import Tacotron
        return Tacotron
    if name == "LengthRegulator":
        from .length_regulator import LengthRegulator
        return LengthRegulator
    if name == "BiLSTM":
        from .bilstm import BiLSTM
        return BiLSTM
    if name == "BiGRU":
        from .bigru import BiGRU
        return BiGRU
    if name == "ForwardTacotron":
        from .forward_tacotron import ForwardTacotron
        return ForwardTacotron
    if name in ("FastPitch", "MultiFastPitch", "MultiForwardTacotron"):
        from . import forward_models
        return getattr(forward_models, name)
    if name == "WaveRNN":
        from .wavernn import WaveRNN
        return WaveRNN
    if name == "DurationExtractor":
        from .duration_extractor import DurationExtractor
        return DurationExtractor
    raise AttributeError(name)
