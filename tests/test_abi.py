"""The C-ABI shared library loads and exports every symbol include/mispec.h declares;
host-side guards fail loudly (no GPU needed: no compute call is made here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "mispec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mispec_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from nnaudio_amd import _abi, build

    build.build(verbose=False)
    lib = ctypes.CDLL(_abi.LIB_PATH)
    names = _declared_functions()
    assert set(names) == set(_abi.EXPORTS)
    for n in names:
        assert hasattr(lib, n), n
    # ... and nothing else: every exported function is one the header declares (no stray C++ symbol)
    import subprocess

    nm = subprocess.run(["nm", "-D", "--defined-only", _abi.LIB_PATH], capture_output=True, text=True).stdout
    exported = {ln.split()[-1] for ln in nm.splitlines() if len(ln.split()) == 3 and ln.split()[1] == "T"}  # (weak STL instantiations aside)
    exported = {e for e in exported if not e.startswith(("_init", "_fini", "__"))}
    assert exported <= set(names), sorted(exported - set(names))
    assert _abi.load().mispec_version() == _abi.ABI_VERSION
    assert isinstance(_abi.load().mispec_last_error(), bytes)  # (the text of this thread's last failure, b"" if none yet)


def test_args_struct_layout_matches_header(tmp_path):
    """sizeof / offsetof as the C compiler sees include/mispec.h == the ctypes mirror."""
    import shutil
    import subprocess

    from nnaudio_amd import _abi

    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no C compiler")
    for cname, mirror in (("mispec_framed_gemm_args", _abi.FramedGemmArgs),
                          ("mispec_planar_args", _abi.PlanarArgs),
                          ("mispec_octave_level", _abi.OctaveLevel),
                          ("mispec_octave_args", _abi.OctaveArgs),
                          ("mispec_octave_stream_args", _abi.OctaveStreamArgs),
                          ("mispec_octave_stream_plan", _abi.OctaveStreamPlan)):
        fields = [f[0] for f in mirror._fields_]
        prog = ['#include <stdio.h>', '#include <stddef.h>', '#include "mispec.h"', 'int main(void){',
                'printf("%%zu\\n", sizeof(%s));' % cname]
        for f in fields:
            prog.append('printf("%%zu\\n", offsetof(%s, %s));' % (cname, f))
        prog.append("return 0;}")
        src = tmp_path / ("layout_%s.c" % cname)
        src.write_text("\n".join(prog))
        exe = tmp_path / ("layout_%s" % cname)
        subprocess.check_call([gcc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
        assert vals[0] == ctypes.sizeof(mirror)
        for f, off in zip(fields, vals[1:]):
            assert getattr(mirror, f).offset == off, (cname, f)


def test_integration_stub_matches_the_header(tmp_path):
    """INTEGRATION.md's ctypes stub, executed: its Args block must be the struct the shipped
    library checks (`struct_size`), field for field."""
    import shutil
    import subprocess

    from nnaudio_amd import _abi, build

    build.build(verbose=False)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch.*?)```", text, flags=re.S).group(1)
    code = code.replace('ctypes.CDLL("libmispec.so")', "ctypes.CDLL(%r)" % _abi.LIB_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    Args = ns["Args"]
    assert [f[0] for f in Args._fields_] == [f[0] for f in _abi.FramedGemmArgs._fields_]
    assert ctypes.sizeof(Args) == ctypes.sizeof(_abi.FramedGemmArgs)
    for name, _ in Args._fields_:
        assert getattr(Args, name).offset == getattr(_abi.FramedGemmArgs, name).offset, name
    gcc = shutil.which("gcc")
    if gcc is not None:
        src = tmp_path / "sz.c"
        src.write_text('#include <stdio.h>\n#include "mispec.h"\nint main(void){printf("%zu\\n", '
                       'sizeof(mispec_framed_gemm_args));return 0;}\n')
        exe = tmp_path / "sz"
        subprocess.check_call([gcc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert int(subprocess.check_output([str(exe)])) == ctypes.sizeof(Args)
    # the library accepts the stub's struct_size (and then stops at the NULL pointers)
    a = Args(struct_size=ctypes.sizeof(Args))
    assert ns["_lib"].mispec_framed_gemm_f32(ctypes.byref(a), None) == -1
    assert b"NULL device pointer" in _abi.load().mispec_last_error()
    assert {"split_basis", "fold_basis", "framed"} <= set(ns)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from nnaudio_amd import _abi

    lib = _abi.load()
    assert lib.mispec_framed_gemm_f32(None, None) == -1
    assert b"NULL" in lib.mispec_last_error()
    a = _abi.FramedGemmArgs()
    a.struct_size = 4
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1
    assert b"struct_size" in lib.mispec_last_error()
    assert lib.mispec_fir_decimate_f32(None, 0, 1, 1, None, 1, 1, 0, None, 0, 1, None, 0, None) == -1
    assert lib.mispec_filterbank_f32(None, 1, 1, None, 1, 1, None, None) == -1
    # argument validation happens before any device work: every entry refuses NULL / bad sizes
    assert lib.mispec_power_to_db_f32(None, 1, 1, 1e-10, 1.0, 80.0, None, None, 0, None) == -1
    assert lib.mispec_power_to_db_bwd_f32(None, None, 1, 1, 1e-10, 80.0, None, None, 0, None) == -1
    assert lib.mispec_mfcc_tail_f32(None, 1, 128, 10, 1e-10, 1.0, 80.0, None, 20, None, None) == -1
    assert lib.mispec_mfcc_tail_f32(4096, 1, 128, 10, 0.0, 1.0, 80.0, 8192, 20, 12288, None) == -1 and b"amin" in lib.mispec_last_error()
    assert lib.mispec_mfcc_tail_f32(4096, 1, 128, 10, 1e-10, 1.0, 80.0, 8192, 20, 4096, None) == -1 and b"alias" in lib.mispec_last_error()
    assert lib.mispec_mfcc_tail_f32(4096, 1, 300, 10, 1e-10, 1.0, 80.0, 8192, 20, 12288, None) == -2 and b"256 mel bands" in lib.mispec_last_error()
    assert lib.mispec_mfcc_tail_f32(4096, 1, 16, 10, 1e-10, 1.0, 80.0, 8192, 20, 12288, None) == -2  # n_mfcc > n_mels
    assert lib.mispec_istft_grad_signal_f32(None, 0, 1, 1, 1, None, 1, 0, 1, None, None) == -1
    assert lib.mispec_overlap_add_f32(None, 1, 1, 1, None, 1, 0, None, 0, 1, None) == -1
    assert lib.mispec_basis_split_bytes(0, 16, 1) == -1 and lib.mispec_basis_split_bytes(4, 48, 0) == 2 * 4 * 64 * 2
    # complex banks: four planes + the fragment-order copy of one 16-bin tile + its block of zeros
    assert lib.mispec_basis_split_bytes(4, 48, 1) == 4 * 4 * 64 * 2 + 1 * 64 * 128 + 4096
    # the fp32 fragment-order copy alone (MISPEC_PREC_F32 on the strip kernel): 17 bins = 2 tiles
    assert lib.mispec_basis_frag_bytes(17, 48) == 2 * 64 * 128 + 4096 and lib.mispec_basis_frag_bytes(0, 48) == -1
    assert lib.mispec_basis_frag_bytes(1025, 64) == -2
    assert lib.mispec_frag_basis_f32(None, None, 0, 4, 48, None, 0, None) == -1
    # the fused filterbank fields are validated with the rest of the block (fake non-NULL pointers:
    # nothing is dereferenced on the host)
    a = _abi.FramedGemmArgs()
    a.struct_size = ctypes.sizeof(_abi.FramedGemmArgs)
    a.x = a.basis_re = a.basis_im = a.out = 4096
    a.n_clips, a.n_samples, a.n_frames, a.n_bins, a.kernel, a.hop = 1, 1024, 5, 129, 256, 64
    a.pad, a.pad_mode, a.epilogue, a.power, a.im_sign = 128, 2, 2, 2.0, -1.0
    a.fb, a.n_fb = 4096, 8  # no fb_support
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1 and b"fb_support" in lib.mispec_last_error()
    a.fb_support, a.power = 4096, 1.5
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1 and b"power 1 or 2" in lib.mispec_last_error()
    a.power, a.n_fb = 2.0, 300
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -2 and b"256 filters" in lib.mispec_last_error()
    a.n_fb, a.tile = 8, 1
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -2 and b"automatic tile" in lib.mispec_last_error()
    a.tile, a.reserved4 = 0, 1
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1 and b"reserved" in lib.mispec_last_error()
    # the ablation bits of `reserved` exist only in the benchmarking build: the product library
    # refuses them instead of computing wrong spectrograms
    a.reserved4, a.out_frame_major = 0, 2
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1 and b"out_frame_major" in lib.mispec_last_error()
    a.out_frame_major, a.reserved = 0, 16
    assert lib.mispec_framed_gemm_f32(ctypes.byref(a), None) == -1 and b"reserved must be 0" in lib.mispec_last_error()
    assert lib.mispec_framed_gemm_workspace_bytes(ctypes.byref(a)) == -1


def test_no_environment_switch_reaches_the_kernels():
    """MISPEC_DEBUG used to be OR-ed into every call's ablation bits; nothing reads it now."""
    import inspect

    from nnaudio_amd import _abi, engine

    assert "MISPEC_DEBUG" not in inspect.getsource(engine) + inspect.getsource(_abi)


def test_needs_grad_sees_dataparallel_replicas():
    """nn.DataParallel replicas hold trainable parameters as plain attributes (replica.parameters()
    is empty); the graph / no-graph decision must still see them."""
    from nnaudio_amd import engine, features

    m = features.MelSpectrogram(sr=16000, n_fft=256, n_mels=20, hop_length=64, trainable_mel=True,
                                verbose=False)
    x = torch.zeros(2, 1024)
    assert engine.needs_grad(m, x)
    # what torch.nn.parallel.replicate does to every sub-module (torch/nn/parallel/replicate.py)
    rep = m._replicate_for_data_parallel()
    rep.stft = m.stft._replicate_for_data_parallel()
    rep._modules["stft"] = rep.stft
    for key, p in m._parameters.items():
        setattr(rep, key, p.detach().clone().requires_grad_(True) * 1.0)  # non-leaf copy
    assert list(rep.parameters()) == []
    assert engine.needs_grad(rep, x)
    with torch.no_grad():
        assert not engine.needs_grad(rep, x)
    frozen = features.MelSpectrogram(sr=16000, n_fft=256, n_mels=20, hop_length=64, verbose=False)
    assert not engine.needs_grad(frozen, x)
    assert engine.needs_grad(frozen, x.clone().requires_grad_(True))


def test_cpu_tensors_take_the_host_path_or_fail_loudly():
    """CPU tensors run on libmispec's own host loops (the forward of the six hot-path modules, like
    the reference's conv1d: wherever the input lives); operations without a host implementation
    -- and everything once the host path is switched off -- raise."""
    from nnaudio_amd import engine, features

    m = features.STFT(n_fft=64, hop_length=16, verbose=False)
    y = m(torch.zeros(1, 256))
    assert tuple(y.shape) == (1, 33, 17, 2) and y.device.type == "cpu" and float(y.abs().max()) == 0.0
    with pytest.raises(ValueError):
        m(torch.zeros(1, 1, 1, 256))
    with pytest.raises(AssertionError):
        m(torch.zeros(1, 16))
    with pytest.raises(RuntimeError, match="GPU only"):  # backward needs the HIP kernels
        features.STFT(n_fft=64, hop_length=16, trainable=True, verbose=False)(torch.zeros(1, 256))
    # MFCC and the inverse STFT have host loops too (round 4): forward of CPU tensors, no gradients
    c = features.MFCC(sr=16000, n_mfcc=13, n_fft=256, n_mels=32, hop_length=64, verbose=False)(torch.ones(1, 4000))
    assert tuple(c.shape) == (1, 13, 63) and c.device.type == "cpu" and bool(torch.isfinite(c).all())
    assert engine.mfcc_tail(torch.ones(1, 32, 63), 1e-10, 1.0, 80.0, torch.ones(13, 32)) is None  # (the one-launch tail is a device kernel)
    inv = features.STFT(n_fft=64, hop_length=16, iSTFT=True, verbose=False)
    xr = torch.randn(2, 640, generator=torch.Generator().manual_seed(0))
    back = inv.inverse(inv(xr), length=640)
    assert back.device.type == "cpu" and float((back - xr).abs().max()) < 1e-4
    engine.set_host_path(False)
    try:
        with pytest.raises(RuntimeError, match="GPU only"):
            m(torch.zeros(1, 256))
    finally:
        engine.set_host_path(True)


def test_inverse_stft_needs_the_gpu():
    from nnaudio_amd import features

    # (its backward, trainable window included, runs on the HIP kernels: tests/test_gpu_parity.py)
    m = features.iSTFT(n_fft=64, hop_length=16, trainable_window=True, verbose=False)
    with pytest.raises(RuntimeError, match="GPU only"):
        m(torch.zeros(1, 33, 8, 2), onesided=True)


def test_legacy_import_shim_warns():
    import importlib
    import warnings

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        mod = importlib.import_module("nnaudio_amd.Spectrogram")
        importlib.reload(mod)
    assert any("deprecated" in str(i.message) for i in w)
    assert hasattr(mod, "STFT") and hasattr(mod, "CQT2010v2") and hasattr(mod, "VQT")


def test_filterbank_support_tables():
    """Host plumbing of the fused filterbank epilogue: [first, last + 1) of every filter row and
    the average number of filters covering a bin; the fusion rule follows include/mispec.h."""
    import numpy as np

    from nnaudio_amd import engine
    from nnaudio_amd.basis import mel_filterbank

    fb = torch.zeros(5, 40)
    fb[0, 3:7] = 1.0
    fb[1, 6] = 0.5
    fb[3, 0:40] = 2.0
    fb[4, 39] = 1.0
    sup, cov = engine.filterbank_support(fb)
    assert sup.dtype == torch.int32 and sup.tolist() == [[3, 7], [6, 7], [0, 0], [0, 40], [39, 40]]
    assert abs(cov - (4 + 1 + 0 + 40 + 1) / 40.0) < 1e-6
    mel = torch.from_numpy(np.asarray(mel_filterbank(22050, 1024, n_mels=128), dtype=np.float32))
    sup, cov = engine.filterbank_support(mel)
    assert 1.5 < cov < 2.6  # triangular filters: every bin under about two of them
    assert engine.fused_filterbank_ok(2.0, cov, 128) and engine.fused_filterbank_ok(1.0, cov, 128)
    assert not engine.fused_filterbank_ok(1.5, cov, 128)    # exponent 1 or 2 only
    assert not engine.fused_filterbank_ok(2.0, 64.0, 64)    # dense (gammatone)
    assert not engine.fused_filterbank_ok(2.0, cov, 300)    # > 256 filters


def _plan_args(m, B, L, keep, precision="bf16x3"):
    """Argument block of CQT1992v2's contraction with fake device pointers (nothing is dereferenced
    by the host-only planning query) and the real supports."""
    from nnaudio_amd import _abi, engine
    from nnaudio_amd.features._cqt_common import SupportCache

    kr = m.cqt_kernels_real.reshape(m.cqt_kernels_real.shape[0], -1)
    ki = m.cqt_kernels_imag.reshape(kr.shape)
    sup = SupportCache._build(kr, ki)
    F, K = kr.shape
    hop = m.hop_length
    a = _abi.FramedGemmArgs()
    a.struct_size = ctypes.sizeof(_abi.FramedGemmArgs)
    a.x = a.basis_re = a.basis_im = a.out = a.row_support = a.basis_split = 4096
    a.n_clips, a.n_samples, a.x_clip_stride = B, L, L
    a.hop, a.pad, a.pad_mode, a.n_frames = hop, K // 2, 2, L // hop + 1
    a.basis_row_stride, a.n_bins, a.kernel = K, F, K
    a.epilogue, a.im_sign = engine.EPI_MAGNITUDE, -1.0
    a.out_clip_stride, a.out_row_stride = F * a.n_frames, a.n_frames
    if precision == "bf16x3":
        a.precision = engine.PREC_BF16X3
        a.basis_split_bytes = _abi.load().mispec_basis_split_bytes(F, K, 1)
    else:  # fp32: the fragment-order copy of the bank in basis_split
        a.precision = engine.PREC_F32
        a.basis_split_bytes = _abi.load().mispec_basis_frag_bytes(F, K)
    a.row_support_host = sup.host_copy.ctypes.data
    keep.append(sup)
    return a, sup.host_copy


@pytest.mark.parametrize("cfg", [
    dict(sr=44100, hop_length=512, n_bins=84, bins_per_octave=12, B=64, L=441000),   # cfg4 / the bench
    dict(sr=44100, hop_length=512, n_bins=84, bins_per_octave=12, B=16, L=441000),   # one rank's shard
    dict(sr=22050, hop_length=512, n_bins=96, bins_per_octave=24, B=3, L=70000),
    dict(sr=22050, hop_length=512, n_bins=120, bins_per_octave=36, fmin=110.0, B=2, L=80000),
    dict(sr=16000, hop_length=128, n_bins=70, bins_per_octave=12, fmin=55.0, B=5, L=33000),
])
@pytest.mark.parametrize("precision", ["bf16x3", "fp32"])
def test_strip_plan_covers_every_row_tile_once(cfg, precision):
    """Host logic of the strip kernel's launch plan (mispec_strip_plan, no GPU): every 16-bin row
    tile sits in exactly one pass, its waves cut its super-stages into consecutive runs, the
    reduction masks of a group cover the four frame tiles once, the slab holds the pass."""
    from nnaudio_amd import _abi, features

    cfg = dict(cfg)
    B, L = cfg.pop("B"), cfg.pop("L")
    m = features.CQT1992v2(verbose=False, **cfg)
    keep = []
    a, sup = _plan_args(m, B, L, keep, precision)
    lib = _abi.load()
    buf = (ctypes.c_int32 * (1 + 8 * 36))()
    n_pass = lib.mispec_strip_plan(ctypes.byref(a), 256, buf, len(buf))
    assert 1 <= n_pass <= 8, _abi.load().mispec_last_error()
    hop, K, F = a.hop, a.kernel, a.n_bins
    # jobs of 128 frames, or of 64 when the 128-frame ones would not fill the workgroup slots
    bn = 128 if buf[0] == -(-B * a.n_frames // 128) else 64
    assert buf[0] == -(-B * a.n_frames // bn)
    seen = {}
    for i in range(n_pass):
        cost, jbase, span, rows = buf[1 + 36 * i:5 + 36 * i]
        assert rows % 16 == 0 and bn + 2 * (span - 1) <= rows <= 288
        waves = [buf[5 + 36 * i + 8 * w:13 + 36 * i + 8 * w] for w in range(4)]
        groups = {}
        for w, (tile, kb, ke, ja, jb, g0, gsize, fmask) in enumerate(waves):
            if tile < 0:
                continue
            assert g0 <= w < g0 + gsize
            groups.setdefault(tile, []).append((w, kb, ke, ja, jb, g0, gsize, fmask))
        for tile, ws in groups.items():
            assert tile not in seen
            seen[tile] = i
            assert [w for w, *_ in ws] == list(range(ws[0][5], ws[0][5] + ws[0][6]))
            kb, ke = ws[0][1], ws[0][2]
            lo = min(int(s) for s, e in sup[16 * tile:16 * tile + 16] if e > s)
            hi = max(int(e) for s, e in sup[16 * tile:16 * tile + 16] if e > s)
            assert kb % 32 == 0 and kb <= lo and ke >= hi and kb > lo - 32
            # consecutive runs of super-stages from the first to the last one the taps touch
            assert ws[0][3] == kb // hop and ws[-1][4] == (ke - 1) // hop + 1
            for (_, _, _, _, jb0, *_), (_, _, _, ja1, *_) in zip(ws, ws[1:]):
                assert jb0 == ja1
            assert all(jbase <= ja <= jb <= jbase + span for _, _, _, ja, jb, *_ in ws)
            masks = [fm for *_, fm in ws]
            assert sum(masks) == 15 and all(x & y == 0 for k, x in enumerate(masks) for y in masks[k + 1:])
    assert sorted(seen) == list(range(-(-F // 16)))


def test_strip_plan_declines_what_the_kernel_does_not_cover():
    from nnaudio_amd import _abi, features

    lib = _abi.load()
    buf = (ctypes.c_int32 * 300)()
    keep = []
    m = features.CQT1992v2(sr=44100, hop_length=512, n_bins=84, verbose=False)
    a, _ = _plan_args(m, 4, 441000, keep)
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) > 0
    a.row_support_host = None  # no host copy of the supports: the narrow-tile kernel
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) == 0
    a, _ = _plan_args(m, 4, 441000, keep)
    a.hop = 500  # not a multiple of 32 taps
    a.n_frames = 441000 // 500 + 1
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) == 0
    a, _ = _plan_args(m, 1, 30000, keep)  # 59 frames: even a 64-frame job would straddle several clips
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) == 0
    # kernels of 90 hops: the slab of a 128-frame job would need more than 288 rows -- 64-frame jobs fit
    m2 = features.CQT1992v2(sr=22050, hop_length=256, n_bins=96, bins_per_octave=24, verbose=False)
    a, _ = _plan_args(m2, 3, 70000, keep)
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) > 0 and buf[0] == -(-3 * a.n_frames // 64)
    # ... and kernels of 180 hops do not
    m3 = features.CQT1992v2(sr=22050, hop_length=128, n_bins=96, bins_per_octave=24, verbose=False)
    a, _ = _plan_args(m3, 3, 70000, keep)
    assert lib.mispec_strip_plan(ctypes.byref(a), 256, buf, 300) == 0
    a, _ = _plan_args(m, 4, 441000, keep)
    assert lib.mispec_strip_plan(ctypes.byref(a), 0, buf, 300) == -1


def test_documented_build_line_names_every_unit():
    """the hand-written hipcc line must compile every translation unit build.py compiles (since
    ABI 11 there are two: a library built from mispec.hip alone lacks mispec_octave_stream_*), and the units' objects
    together must define every function include/mispec.h declares."""
    import subprocess

    from nnaudio_amd import build

    build.build(verbose=False)
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = text[text.index("## 1. Build"):text.index("## 2.")]
    line = " ".join(ln.rstrip("\\").strip() for ln in block.splitlines() if "hipcc" in ln or ln.strip().startswith("nnaudio_amd/csrc/"))
    for src, _ in build.UNITS:
        assert os.path.relpath(src, ROOT) in line, (src, line)
    assert "-shared" in line and "gfx950" in line and "-I include" in line
    defined = set()
    objdir = os.path.join(os.path.dirname(build.SRC), "_obj")
    for src, _ in build.UNITS:
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        nm = subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True, check=True).stdout
        defined |= {ln.split()[-1] for ln in nm.splitlines() if ln.split()[1:2] == ["T"]}
    missing = [n for n in _declared_functions() if n not in defined]
    assert not missing, missing


def test_scratch_guard_sees_cached_objects(tmp_path, monkeypatch):
    """the refusal of a spilling build must not depend on the object having been compiled in THIS call.  The
    remarks travel with the object (obj.remarks.json); an object without a record, or one compiled with other flags, is
    stale; a refused build takes the offending objects off the disk."""
    import json

    from nnaudio_amd import build

    # the allow-list: framed_* / octave_stream* never, stft_fft_* / istft_* only the listed instances within their bytes
    r = {"framed_fold_kernelENS_7KParamsE": 8, "octave_stream_kernelILi6ELb1EEvNS_3OSPE": 4,
         "stft_fft_kernelILi512ELi2ELb0ELin1EEEvNS_7KParamsEi": 20, "stft_fft_kernelILi1024ELi1ELb0ELin1EEEvNS_7KParamsEi": 4,
         "istft_ola_fft_kernelILi1024EEEvPKfiiS2_iiiPfxiii": 12, "fold2_frames_kernel": 64, "clean": 0}
    bad = build.refused_scratch(r, ablate=False)
    assert set(bad) == {"framed_fold_kernelENS_7KParamsE", "octave_stream_kernelILi6ELb1EEvNS_3OSPE",
                        "stft_fft_kernelILi1024ELi1ELb0ELin1EEEvNS_7KParamsEi"}
    assert "octave_stream_kernelILi6ELb1EEvNS_3OSPE" not in build.refused_scratch(r, ablate=True)  # (benchmarking build: warned)
    # the record next to an object
    obj = str(tmp_path / "unit.o")
    assert build._cached_remarks(obj, False) is None             # no object, no record
    open(obj, "w").write("x")
    assert build._cached_remarks(obj, False) is None             # an object without a record is stale
    json.dump({"flags": build._flags_key(False), "scratch": {"framed_x": 16}}, open(build._sidecar(obj), "w"))
    assert build._cached_remarks(obj, False) == {"framed_x": 16}  # ... seen again by every later build() call
    assert build._cached_remarks(obj, True) is None              # other defines: stale whatever its date
    build._discard(obj)
    assert not os.path.exists(obj) and not os.path.exists(build._sidecar(obj))
