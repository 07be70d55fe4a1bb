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


def _canned(n_gpus=1):
    """A full result dict of the shape main() builds, with the longest strings the script writes."""
    prose = "x" * 400
    pk = {"k%d" % i: {"FETCH_SIZE": 1.234567e8, "WRITE_SIZE": 7.654321e7, "SQ_INSTS_VALU_MFMA_MOPS_F16": 0.0}
          for i in range(8)}
    blk = dict(bench.roofline_block(dict(flops=4.632e11, bytes=3.559e8, bound="hbm", executed=None), 0.159e-3, "fft",
                                    traffic=3.6e8, kernel=prose),
               traffic_detail={"fetch_bytes": 1.2e8, "write_bytes": 2.4e8, "per_kernel": pk, "method": prose},
               dominant_kernel={"name": "stft_fft_kernel<1024, 1, false> (" + prose + ")", "avg_ms": 0.159})
    path = {"frames_per_s": 1.0687613104232763e8, "ms_per_step": 0.5161863501416519, "step_device_ms": 0.515954475402832,
            "algorithmic_tflops": 897.82, "algorithmic_frac": 1.0773849658848194, "mfma_frac": 0.2690834646502573,
            "hbm_frac_on_algorithmic_bytes": 0.08621845941983243, "dominant_kernel": dict(name=prose, avg_ms=0.5),
            "what": prose, "dynamic_range_db": -152.9}
    names = ["cqt", "cqt_f16x3", "cqt_bf16x3", "stft_trainable_fwd_bwd", "mel", "gammatone", "cqt2010", "vqt", "cqt2010_bf16x3",
             "cqt2010_fp32", "mel_f16x3", "gammatone_f16x3", "cqt_f16x3_cfg4_shard", "istft", "mfcc", "stft256", "stft4096"]
    out = {"metric": "spectrogram frames/sec", "value": 347220123.456, "unit": "frames/s", "n_gpus": n_gpus, "steps": 200,
           "warmup": 50, "ms_per_step": 0.158881234, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic", "prewarm_ms": 300.1, "kernel_source_sha": "0123456789abcdef",
           "config": {"workload": "STFT n_fft=2048 hop=512 hann, B=64 x 10 s @ 44.1 kHz, Magnitude (configs[1])",
                      "global_batch": 64 * n_gpus, "precision": prose, "precision_short": "fp32 FFT per frame",
                      "parallelism": "batch-sharded x%d, no data-path collective" % n_gpus},
           "roofline": blk, "paths": {k: dict(path) for k in ("fft", "f16x3", "bf16x3", "fp32")},
           "roofline_cqt84": dict(blk, precision="fp32", default_module=True, ms_per_step=1.37, frames_per_s=4.0e7, workload=prose, same_bits_as_torch_conv1d=1.0),
           "roofline_cqt84_f16x3": dict(blk, precision="f16x3", default_module=False, ms_per_step=0.4, frames_per_s=1.3e8, workload=prose),
           "extra": {k: dict(path, roofline=dict(blk), workload=prose, precision="f16x3") for k in names},
           "cpu_baseline": {"value": 46194.8, "unit": "frames/s", "cores": 128, "kind": "port", "sample": prose,
                            "librosa_equivalent": {"value": 92590.7, "what": prose}, "cqt84": {"value": 8737.9, "sample": prose}}}
    if n_gpus > 1:
        g = {"with_gather_ms_per_step": 1.234, "without_gather_ms_per_step": 0.159, "bytes_per_rank": 2.26e8}
        out["gather"] = dict(g, what=prose, cqt_cfg4_shard=dict(g), cqt2010_cfg5_shard=dict(g))
        out["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": None, "kind": "port", "sample": "measured at N=1 only"}
    return out


@pytest.mark.parametrize("n_gpus", [1, 8])
def test_line_is_small(n_gpus, tmp_path, capsys):
    """The driver keeps 8 KB of stdout: the line must be ONE line, the LAST one, well under that, and
    still carry the contract's fields."""
    import json

    out = _canned(n_gpus)
    assert len(json.dumps(out)) > 20000  # (the full record is what did not fit in r3)
    print("some earlier chatter")
    line = bench.emit(out, detail_dir=str(tmp_path))
    printed = capsys.readouterr().out.rstrip("\n").split("\n")
    assert printed[-1] == line and "\n" not in line
    assert len(line) < 4096
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("STFT n_fft=2048 hop=512")
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(d["roofline"])
    assert d["roofline"]["dominant_kernel"]["name"] == "stft_fft_kernel<1024, 1, false>"
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(d["cpu_baseline"])
    assert set(d["paths"]) == {"fft", "f16x3", "bf16x3", "fp32"} and "mfma_frac" in d["paths"]["f16x3"]
    assert "cqt2010" in d["extra"] and "ms_per_step" in d["extra"]["cqt2010"]
    # CQT84, the other half of the metric: the module as it ships, and the opt-in arithmetic named as such
    # ONE CQT84 headline block, the module as it ships; the opt-in arithmetic's block stays in bench_detail.json (round 6)
    assert d["roofline_cqt84"]["default_module"] is True and "roofline_cqt84_f16x3" not in d
    assert d["roofline_cqt84"]["same_bits_as_torch_conv1d"] == 1.0
    # the side file holds the full record and the line names it
    name = "bench_detail.json" if n_gpus == 1 else "bench_detail_n8.json"
    full = json.load(open(tmp_path / name))
    assert full["extra"]["mel"]["roofline"]["traffic_detail"]["per_kernel"]
    assert d["detail"].endswith(name) and len(d["detail_sha16"]) == 16
    if n_gpus > 1:
        assert d["gather"]["cqt2010_cfg5_shard"]["with_gather_ms_per_step"] == pytest.approx(1.234)


def test_same_bits_probe_runs_on_the_host_path():
    """bench.py's live comparison of the CQT1992v2 module with torch's conv1d (`same_bits_as_torch_conv1d` on the line) runs
    and returns a fraction (on the CPU oneDNN blocks short kernels differently from the host loops' FMA chain: any value;
    on the MI355X the line shows 1.0, which tests/test_reference_order.py asserts against the library's reference kernel)."""
    import torch

    import bench
    from nnaudio_amd import features

    m = features.CQT1992v2(sr=8000, hop_length=256, fmin=220, n_bins=24, bins_per_octave=12, verbose=False)
    frac = bench.same_bits_as_conv1d(m, torch.randn(1, 8000))
    assert 0.0 <= frac <= 1.0, frac


def test_reference_operator_sequence_probe_runs_on_the_host_path():
    """bench.py's `reference_ops_on_this_gpu` block (the reference's pad + conv1d + sqrt on the bench device) -- here on the CPU,
    against the module's host path: the two agree to float32 rounding."""
    import torch

    import bench

    r = bench.reference_ops_on_gpu(torch.randn(2, 6000))
    assert r["unit"] == "frames/s" and r["value"] > 0 and r["max_diff_of_peak"] <= 1e-5, r
