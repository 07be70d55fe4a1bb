import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    from tests import _golden

    return _golden.Golden()


@pytest.fixture(autouse=True)
def _stft_route(request):
    """The STFT family has two routes on the GPU: the FFT path (csrc/stft_fft.inl; the default for window x DFT
    kernels) and the contraction kernels.  The suites written for the contraction kernels keep testing them
    (FFT switched off); tests/test_gpu_fft.py runs the FFT path -- kernel-level cases, the golden cases of the
    STFT family and the BASELINE-sized checks once more."""
    try:
        from nnaudio_amd import engine
    except Exception:  # (the library is not built: the tests that need it fail on their own)
        yield
        return
    old = engine.set_fft(request.node.module.__name__.endswith("test_gpu_fft"))
    yield
    engine.set_fft(old)


@pytest.fixture(params=[False, True], ids=["contraction", "fft"])
def both_stft_routes(request):
    """For the suites that are about what sits AROUND the STFT kernels (autograd, MFCC, the inverse, sharding):
    once on the contraction kernels and once on the FFT route, the shipped default."""
    from nnaudio_amd import engine

    old = engine.set_fft(request.param)
    yield request.param
    engine.set_fft(old)
