"""torch.compile / torch.export see the feature modules as opaque custom ops (nnaudio_amd.ops):
traced here with fake tensors, no GPU needed (the ops' fake implementations give the shapes)."""
import warnings

import pytest
import torch

from tests._golden import build_module

CASES = [
    ("STFT", dict(n_fft=256, hop_length=64, output_format="Magnitude"), (3, 126, 63)),
    ("STFT", dict(n_fft=256, hop_length=64, output_format="Complex"), (3, 129, 63, 2)),
    ("MelSpectrogram", dict(sr=16000, n_fft=256, n_mels=32, hop_length=64), (3, 32, 63)),
    ("MFCC", dict(sr=16000, n_mfcc=13, n_fft=256, n_mels=32, hop_length=64), (3, 13, 63)),
    ("CQT1992v2", dict(sr=16000, hop_length=64, fmin=110, n_bins=48, output_format="Magnitude"), (3, 48, 63)),
    ("CQT2010v2", dict(sr=16000, hop_length=64, fmin=110, n_bins=48, output_format="Complex",
                       earlydownsample=False), (3, 48, 63, 2)),
    ("VQT", dict(sr=16000, hop_length=64, fmin=110, n_bins=48, gamma=5, earlydownsample=False), (3, 48, 63)),
    # the frequency-domain CQT2010 calls the octave loop with its own scale / imaginary sign: it must
    # NOT take the single-op route (whose schema carries neither)
    ("CQT2010", dict(sr=16000, hop_length=64, fmin=110, n_bins=48, output_format="Magnitude",
                     earlydownsample=False), None),
]


@pytest.mark.parametrize("cls,ctor,shape", CASES)
def test_modules_export_without_graph_breaks(cls, ctor, shape):
    ctor = dict(ctor)
    if cls == "STFT" and shape[1] == 126:
        ctor["freq_bins"] = 126
    mod = build_module(dict(cls=cls, ctor=ctor, fwd={}))
    x = torch.randn(3, 4000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ep = torch.export.export(mod, (x,))  # strict tracing: a graph break raises
    targets = [str(n.target) for n in ep.graph.nodes if n.op == "call_function"]
    if cls in ("MelSpectrogram", "MFCC"):
        # the fused-epilogue path is one op (it picks fused / two kernels at run time, on the real filterbank)
        assert any("mispec.stft_filterbank" in t for t in targets), targets
    elif cls == "CQT2010":
        assert not any("mispec.octave_recursion" in t for t in targets), targets
        assert any("mispec.framed_gemm" in t for t in targets) and any("mispec.fir_decimate" in t for t in targets)
    elif cls in ("CQT2010v2", "VQT"):
        # the whole octave recursion is one op (the fused pyramid kernel in bf16x3)
        assert any("mispec.octave_recursion" in t for t in targets), targets
        assert not any("mispec.fir_decimate" in t for t in targets)
    else:
        assert any("mispec.framed_gemm" in t for t in targets), targets
    if cls == "MFCC":
        assert any("mispec.power_to_db" in t for t in targets) and any("mispec.filterbank" in t for t in targets)
    out = [n for n in ep.graph.nodes if n.op == "output"][0].args[0][0]
    if shape is not None:
        assert tuple(out.meta["val"].shape) == shape


def test_ops_are_registered_with_fake_implementations():
    from nnaudio_amd import ops  # noqa: F401

    x = torch.empty(2, 1000, device="meta")
    w = torch.empty(33, 1, 64, device="meta")
    y = torch.ops.mispec.framed_gemm(x, w, w, 16, 32, 2, 0, -1.0, 0.0, 2.0, None, False, "bf16x3")
    assert y.shape == (2, 33, 63, 2) and y.device.type == "meta"
    assert torch.ops.mispec.fir_decimate(x, torch.empty(256, device="meta"), 2).shape == (2, 500)


@pytest.mark.parametrize("trainable,x_grad", [(True, False), (False, True)])
def test_cqt1992v2_compile_with_a_graph_reaches_the_autograd_function(trainable, x_grad):
    """Grad mode on + trainable kernels (or a differentiable input) under torch.compile: the
    ``support`` hint of the compile branch must be consumed before the autograd function sees the
    keyword arguments (it used to reach ``_framed_args`` and raise TypeError, which dynamo does not
    turn into a graph break).  On CPU the call must get as far as the device check."""
    mod = build_module(dict(cls="CQT1992v2", ctor=dict(sr=16000, hop_length=64, fmin=110, n_bins=24,
                                                       trainable=trainable), fwd={}))
    x = torch.randn(2, 4000, requires_grad=x_grad)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(RuntimeError, match="GPU only"):
            torch.compile(mod, backend="eager")(x)
    torch._dynamo.reset()
