"""The host path (CPU tensors -> libmispec's plain C++ loops, include/mispec.h mispec_*_host_f32)
against the reference's own outputs and the pinned oracle on the golden cases: no GPU needed.
BASELINE configs[0] (STFT n_fft=512 hop=128, batch 1, 1 s @ 16 kHz on CPU) is one of them."""
import warnings

import numpy as np
import pytest
import torch

from tests import _golden
from tests._golden import assert_parity, assert_phase_parity, build_module, is_phase, oracle_forward

HOST_CLASSES = ("STFT", "MelSpectrogram", "Gammatonegram", "CQT1992v2", "CQT", "CQT2010v2", "VQT")


def _host_cases():
    g = _golden.Golden()
    names = []
    for name in _golden.case_names(forward_only=True):
        case = g.cases[name]
        if case["cls"] in HOST_CLASSES and not _golden.is_inverse(case) and case.get("method", "forward") == "forward":
            # the plumbing-sized ones: a second of audio or so per clip
            x = g.inputs[case["input"]]
            if x.size <= 4 * 25000:
                names.append(name)
    return names


@pytest.mark.parametrize("name", _host_cases())
def test_host_path_matches_reference_and_oracle(golden, name):
    case = golden.cases[name]
    x = golden.inputs[case["input"]]
    ref = golden.forward[name]
    mod = build_module(case)  # on the CPU
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        y = mod(torch.as_tensor(x), **case["fwd"])
    assert y.device.type == "cpu"
    y = y.numpy()
    assert y.dtype == ref.dtype and list(y.shape) == case["out_shape"]
    orc = oracle_forward(build_module(case), case, x)
    if is_phase(case):
        mcase = dict(case, ctor=dict(case["ctor"], output_format="Magnitude"), fwd={})
        mag = oracle_forward(build_module(mcase), mcase, x)
        assert_phase_parity(y, ref, mag, what=name + " vs reference")
        assert_phase_parity(y, orc, mag, what=name + " vs oracle")
    else:
        assert_parity(y, ref, rel=1e-4, what=name + " vs reference")
        assert_parity(y, orc, rel=1e-4, what=name + " vs oracle")


def test_baseline_config0_on_the_host():
    """BASELINE.json configs[0]: STFT n_fft=512 hop=128, batch 1, 1 s @ 16 kHz, CPU."""
    from nnaudio_amd import features
    from oracle import spectral_oracle as O

    m = features.STFT(n_fft=512, hop_length=128, verbose=False)
    x = torch.randn(1, 16000, generator=torch.Generator().manual_seed(0))
    y = m(x).numpy()
    assert y.shape == (1, 257, 126, 2)
    want = O.stft(x.numpy(), m.wsin.numpy(), m.wcos.numpy(), 128, output_format="Complex")
    assert_parity(y, want, rel=1e-5, what="configs[0]")


def _host_cases_next():
    """MFCC (power_to_db + DCT on the host) and the inverse STFT cases of the manifest."""
    g = _golden.Golden()
    names = []
    for name in _golden.case_names(forward_only=True):
        case = g.cases[name]
        if case["cls"] == "MFCC" or _golden.is_inverse(case):
            if g.inputs[case["input"]].size <= 4 * 600000:
                names.append(name)
    return names


@pytest.mark.parametrize("name", _host_cases_next())
def test_host_path_mfcc_and_inverse_stft(golden, name):
    """CPU tensors through MFCC.forward (mel.py:309-326) and STFT.inverse / iSTFT.forward (stft.py:15-63) run the
    library's host loops (mispec_power_to_db_host_f32, mispec_istft_host_f32): the reference's outputs."""
    case = golden.cases[name]
    x = golden.inputs[case["input"]]
    ref = golden.forward[name]
    mod = build_module(case)  # on the CPU
    with torch.no_grad(), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        method = case.get("method", "forward")
        fn = mod if method == "forward" else getattr(mod, method)
        y = fn(torch.as_tensor(x), **case["fwd"])
    assert y.device.type == "cpu"
    y = y.numpy()
    assert y.dtype == ref.dtype and list(y.shape) == case["out_shape"]
    if _golden.is_inverse(case):
        y_r, _ = _golden.well_conditioned(case, mod, x, y, ref)
        assert_parity(y_r, ref, rel=1e-4, what=name + " vs reference")
    else:
        assert_parity(y, ref, rel=2e-4, what=name + " vs reference")


def test_host_power_to_db_takes_the_magnitude_of_ref():
    """the reference (mel.py:276) and the device entries use |ref|; the host entry must too."""
    from nnaudio_amd import engine

    spec = torch.rand(2, 5, 7) + 1e-3
    a = engine.power_to_db(spec, 1e-10, 2.5, 80.0)
    b = engine.power_to_db(spec, 1e-10, -2.5, 80.0)
    assert torch.equal(a, b)
    want = 10.0 * torch.log10(torch.clamp(spec, min=1e-10)) - 10.0 * np.log10(2.5)
    want = torch.maximum(want, want.amax(dim=(1, 2), keepdim=True) - 80.0)
    assert float((a - want).abs().max()) < 1e-4


def test_frame_major_output_is_refused_off_the_fft_path():
    """mispec.h out_frame_major is served by the FFT path of the device entry only: the host loops (and through them every
    CPU tensor) refuse it instead of writing a (bins x frames) spectrogram into a frame-major buffer."""
    import numpy as np
    import pytest
    import torch

    from nnaudio_amd import engine

    x = torch.randn(2, 4096)
    n = np.arange(1024)
    wr = torch.tensor(np.cos(2 * np.pi * np.outer(np.arange(513), n) / 1024), dtype=torch.float32)
    wi = torch.tensor(np.sin(2 * np.pi * np.outer(np.arange(513), n) / 1024), dtype=torch.float32)
    with pytest.raises(RuntimeError, match="frame-major|out_frame_major"):
        engine.framed_gemm(x, wr, wi, hop=256, pad=512, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_POWER, power=2.0,
                           out_frame_major=544)
