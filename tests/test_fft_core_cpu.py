"""nnaudio_amd/csrc/fft_core.h (the lane-level arithmetic of the STFT's FFT path) compiled for the host:
the 64 lanes of a wave run one after the other and the result is compared with a float64 DFT -- radix
plans, Stockham index arithmetic, padded exchange buffer and the real-input post-processing, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clang():
    for cand in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++"), shutil.which("amdclang++")):
        if cand and os.path.exists(cand):
            return cand
    return None


def test_fft_core_against_float64_dft(tmp_path):
    cxx = _clang()
    if cxx is None:
        pytest.skip("no clang++ (fft_core.h uses ext_vector_type)")
    exe = str(tmp_path / "fft_core_harness")
    subprocess.run([cxx, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "nnaudio_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "fft_core_harness.cpp"), "-o", exe, "-lm"], check=True)
    res = subprocess.run([exe], stdout=subprocess.PIPE, text=True)
    print(res.stdout)
    assert res.returncode == 0, res.stdout
