"""bench.py's host-side pieces that need no GPU: the tiling model behind the executed-flop figure, the
bounded roofline fraction, and the self-launch of `--gpus N` (which must fail loudly, not hang, when
the GPUs are not there)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_fold_geometry_mirrors_the_library():
    # cfg2: 1025 bins -> 512 even (+ the Nyquist bin in the pre-pass) + 512 odd, n_fft / 4 taps
    assert bench.fold_geometry(1025, 2048) == (1024, 512)
    # cfg3 with the filterbank fused: the single fold, 512 bins in tiles (+ Nyquist), n_fft / 2 taps
    assert bench.fold_geometry(513, 1024, True) == (512, 512)
    assert bench.fold_geometry(513, 1024, False) == (512, 256)
    # a kernel the second fold does not take (not a multiple of 64)
    assert bench.fold_geometry(201, 400) == (256, 208)


def test_roofline_fraction_is_bounded_and_labelled():
    meta = dict(flops=4.632e11, bytes=3.559e8, bound="mfma", executed=2.0 * 2048 * 512 * 55168)
    blk = bench.roofline_block(meta, 0.5e-3, "f16x3")
    assert 0 < blk["frac"] <= 1 and blk["peak"] == 2500.0
    assert blk["frac"] == pytest.approx(blk["executed_frac_of_raw_mfma_peak"])
    assert blk["algorithmic_frac"] > 1  # the folds skip work: that is why it is not `frac`
    assert blk["executed_source"] == "tiling"
    # a workload without a tiling model still reports a fraction that cannot exceed 1
    blk = bench.roofline_block(dict(flops=1e13, bytes=1e8, bound="mfma", executed=None), 1e-3, "fp32")
    assert blk["frac"] == 1.0
    blk = bench.roofline_block(dict(flops=6e10, bytes=4.02e8, bound="hbm", executed=None), 0.5e-3, "f16x3")
    assert blk["unit"] == "GB/s" and blk["frac"] == pytest.approx(4.02e8 / 0.5e-3 / 8e12)


def test_gpus_n_self_launch_fails_loudly_without_the_gpus():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr + r.stdout
