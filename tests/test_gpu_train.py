"""A training step keeps the forward's Complex values for its backward (round 6): the module's output in grad mode is the
pointwise epilogue of those values (mispec_framed_epilogue_fwd_f32) and must carry the SAME BITS as the fused launch of the
no-grad forward; the gradients are those of the recomputing backward."""
import pytest
import torch

from nnaudio_amd import engine, features

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt", ["Magnitude", "Complex", "Phase"])
@pytest.mark.parametrize("precision", [None, "fp32"])
def test_grad_mode_forward_has_the_fused_launch_bits(fmt, precision):
    dev = "cuda:0"
    m = features.STFT(n_fft=512, hop_length=128, window="hann", output_format=fmt, trainable=True, verbose=False).to(dev)
    m.precision = precision
    x = torch.randn(3, 9000, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    with torch.no_grad():
        ref = m(x)
    out = m(x)
    assert out.requires_grad
    assert torch.equal(out.detach(), ref)


def test_epilogue_fwd_matches_every_fused_epilogue():
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn(2, 5000, device=dev, generator=g)
    wr = torch.randn(37, 256, device=dev, generator=g)
    wi = torch.randn(37, 256, device=dev, generator=g)
    scale = torch.rand(37, device=dev, generator=g) + 0.5
    kw = dict(hop=64, pad=128, pad_mode=engine.PAD_REFLECT, im_sign=-1.0, precision="fp32", row_scale=scale)
    z = engine.framed_gemm(x, wr, wi, epilogue=engine.EPI_COMPLEX, **kw)
    for epi, extra in ((engine.EPI_MAGNITUDE, {}), (engine.EPI_MAGNITUDE, {"eps": 1e-8}), (engine.EPI_POWER, {"power": 2.0}),
                       (engine.EPI_POWER, {"power": 1.0}), (engine.EPI_POWER, {"power": 0.6, "eps": 1e-10}),
                       (engine.EPI_PHASE_ATAN2, {}), (engine.EPI_PHASE_COSSIN, {}), (engine.EPI_COMPLEX, {})):
        fused = engine.framed_gemm(x, wr, wi, epilogue=epi, **kw, **extra)
        apart = engine.epilogue_fwd(z, epi, **extra)
        assert torch.equal(fused, apart), (epi, extra)


def test_saved_values_and_recomputation_give_the_same_gradients(monkeypatch):
    dev = "cuda:0"
    x = torch.randn(4, 12000, device=dev, generator=torch.Generator(device=dev).manual_seed(3))

    def grads():
        torch.manual_seed(0)
        m = features.STFT(n_fft=512, hop_length=128, window="hann", output_format="Magnitude", trainable=True,
                          verbose=False).to(dev)
        m(x).pow(2).mean().backward()
        return m.wsin.grad.clone(), m.wcos.grad.clone()

    kept = grads()
    monkeypatch.setenv("MISPEC_SAVE_Z_MAX_BYTES", "0")  # the recomputing backward
    redone = grads()
    for a, b in zip(kept, redone):
        assert torch.equal(a, b)
