"""The allocation-free entry point wrnn_lr_regulate_static (include/wrnn_mi355.h; no counterpart in the reference) against
the existing pair wrnn_lr_prepare + wrnn_lr_expand, through the C ABI, bit for bit (pure integer / copy work): two calls
in a row on one GPU, the second with a smaller input (the scratch keeps the first call's prefix sums), plus a cap below
the true length and the size limit."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def pair(lib, L, x, dur, M):
    B, T, C = x.shape
    st = torch.cuda.current_stream().cuda_stream
    cs = torch.empty(B, T, dtype=torch.int32, device="cuda")
    lens = torch.empty(B, dtype=torch.int32, device="cuda")
    out = torch.full((B, M, C), float("nan"), device="cuda")
    L.check(lib.wrnn_lr_prepare(dur.data_ptr(), B, T, cs.data_ptr(), lens.data_ptr(), st), "wrnn_lr_prepare")
    L.check(lib.wrnn_lr_expand(x.data_ptr(), cs.data_ptr(), B, T, C, out.data_ptr(), M, st), "wrnn_lr_expand")
    return out, lens


def static(lib, L, x, dur, M):
    B, T, C = x.shape
    st = torch.cuda.current_stream().cuda_stream
    lens = torch.empty(B, dtype=torch.int32, device="cuda")
    out = torch.full((B, M, C), float("nan"), device="cuda")     # allocated by the caller, as the inputs are
    L.check(lib.wrnn_lr_regulate_static(x.data_ptr(), dur.data_ptr(), B, T, C, out.data_ptr(), M, lens.data_ptr(), st),
            "wrnn_lr_regulate_static")
    return out, lens


def test_static_entry_matches_prepare_expand_two_calls_in_a_row():
    from wavernn_amd import _lib as L, synth
    lib = L.load()
    cases = [(64, 128, 512, -1.0, 11.0, 900),     # BASELINE configs[3] shape, negative durations to clamp, cap > longest item
             (5, 50, 64, 0.0, 4.0, 120),          # smaller: most of the scratch still holds the first call's sums
             (7, 33, 8, 0.0, 6.0, 40)]            # cap below the longest item: rows past the cap are dropped
    for B, T, C, lo, hi, M in cases:
        x = torch.as_tensor(synth.normalish(3, f"x{B}", (B, T, C))).cuda()
        d = torch.as_tensor(synth.uniform(3, f"d{B}", (B, T), lo, hi))
        d1, d2 = d.clone().cuda(), d.clone().cuda()
        o_ref, l_ref = pair(lib, L, x, d1, M)
        o, l = static(lib, L, x, d2, M)
        torch.cuda.synchronize()
        assert torch.equal(l, l_ref) and torch.equal(d2, d1) and float(d2.min()) >= 0.0
        assert torch.equal(o, o_ref) and not torch.isnan(o).any()
        if M == 40:
            assert int(l.max()) > M                       # the cap really cut
    # no tokens at all: lens = 0 and zero rows, from both entry points
    x0 = torch.zeros(3, 0, 8, device="cuda")
    d0 = torch.zeros(3, 0, device="cuda")
    o_ref, l_ref = pair(lib, L, x0, d0, 5)
    o, l = static(lib, L, x0, d0, 5)
    torch.cuda.synchronize()
    assert torch.equal(l, l_ref) and int(l.abs().max()) == 0 and torch.equal(o, o_ref) and not o.any()
    # the mirror's no-sync path is the same call
    from wavernn_amd import LengthRegulator
    x = torch.as_tensor(synth.normalish(3, "xm", (3, 20, 16))).cuda()
    d = torch.as_tensor(synth.uniform(3, "dm", (3, 20), 0.0, 5.0)).cuda()
    ref = LengthRegulator()(x, d.clone())
    out, lens = LengthRegulator().forward_capped(x, d.clone(), 128)
    assert int(lens.max()) == ref.shape[1] and torch.equal(out[:, :ref.shape[1]], ref) and not out[:, ref.shape[1]:].any()


def test_static_entry_rejects_oversize():
    from wavernn_amd import _lib as L
    lib = L.load()
    x = torch.zeros(1, 4, 4, device="cuda")
    rc = lib.wrnn_lr_regulate_static(x.data_ptr(), x.data_ptr(), 1024, 257, 4, x.data_ptr(), 1, x.data_ptr(), None)
    assert rc != 0 and b"exceeds the static scratch" in lib.wrnn_last_error()
