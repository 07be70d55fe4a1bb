"""The streaming octave kernel's geometry, without a GPU: the plan the library computes
(mispec_octave_stream_plan_of: look-aheads, ring sizes, blocks, segments) is run through the executable
model of the kernel's schedule (scripts/octave_stream_model.py: rings full of stale NaNs, absolute
positions, the edge patches) and must reproduce the plain recursion -- zero-padded FIR decimation
(utils.py:73-124) + reflect- / zero-padded frames (utils.py:498-521) -- to rounding."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import octave_stream_model as M  # noqa: E402

from nnaudio_amd import _abi  # noqa: E402


def library_plan(L0, hop0, K, n_frames, n_seg, n_cus=256, n_clips=1, pad_reflect=True, precision=2):
    lib = _abi.load()
    a = _abi.OctaveStreamArgs()
    a.struct_size = ctypes.sizeof(_abi.OctaveStreamArgs)
    a.n_levels = len(K)
    a.x, a.out, a.taps = 1 << 20, 1 << 20, 1 << 20  # (never dereferenced by the plan)
    a.x_clip_stride, a.n_clips, a.n_samples = L0, n_clips, L0
    a.hop, a.n_frames, a.n_taps, a.epilogue = hop0, n_frames, 256, 1
    for l, k in enumerate(K):
        if k:
            a.level[l].bank_split, a.level[l].bank_split_bytes = 1 << 20, 1 << 30
            a.level[l].n_bins, a.level[l].kernel = 12, k
            a.level[l].pad_mode = 2 if pad_reflect else 1
    a.precision, a.n_segments = precision, n_seg
    p = _abi.OctaveStreamPlan()
    rc = lib.mispec_octave_stream_plan_of(ctypes.byref(a), n_cus, ctypes.byref(p))
    return rc, p


def as_model_plan(p, L0, hop0, K, n_frames):
    D = p.n_levels
    segs, b = [], 0
    while b < p.n_blocks:
        segs.append((b, min(p.n_blocks, b + p.blocks_per_segment)))
        b += p.blocks_per_segment
    assert len(segs) == p.n_segments
    return dict(D=D, nf=p.frames_per_step, chunk=4096, hop=[hop0 >> l for l in range(D)],
                L=list(p.length[:D]), K=list(K), blk=[4096 >> l for l in range(D)], c=list(p.lookahead[:D]),
                rows=list(p.ring_rows[:D]), n_blocks=p.n_blocks, segs=segs, warm=p.warm_steps,
                n_frames=n_frames, n_taps=256, tiles_span=p.blocks_per_tile)


CASES = [
    # cfg5's first launch in small: four octaves at hop 512, 192-tap (trimmed) banks
    dict(L0=40000, hop0=512, K=(192, 192, 192, 192), n_seg=2),
    dict(L0=33000, hop0=512, K=(192, 192, 192, 192), n_seg=0),
    # the follow-up launch: level 0 = x_3 without a bank, octaves 4-7 below it
    dict(L0=30000, hop0=64, K=(0, 192, 192, 192, 192), n_seg=3),
    # VQT-like widths, zero padding, untrimmed 256-tap banks, other hops
    dict(L0=50000, hop0=256, K=(256, 128, 64, 32), n_seg=2, pad_reflect=False),
    dict(L0=44100, hop0=512, K=(256, 256, 256, 256), n_seg=2),
    dict(L0=20479, hop0=128, K=(64, 64, 32), n_seg=1),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "L%d_h%d_D%d" % (c["L0"], c["hop0"], len(c["K"])))
def test_library_plan_runs_the_model(case):
    L0, hop0, K = case["L0"], case["hop0"], case["K"]
    n_frames = L0 // hop0 + 1
    rc, p = library_plan(L0, hop0, K, n_frames, case["n_seg"], pad_reflect=case.get("pad_reflect", True))
    assert rc == 0, _abi.load().mispec_last_error()
    plan = as_model_plan(p, L0, hop0, K, n_frames)
    # the same geometry as the model derives on its own
    own = M.plan_stream(L0, hop0, list(K), n_frames, n_seg=max(1, p.n_segments))
    assert plan["c"] == own["c"] and plan["rows"] == own["rows"] and plan["L"] == own["L"]
    assert p.lds_bytes <= 160 * 1024
    err, _ = M.check(L0=L0, hop0=hop0, K=K, pad_reflect=case.get("pad_reflect", True), plan=plan)
    assert err < 1e-12


def test_plan_of_the_cfg5_shard():
    """64 x 30 s @ 44.1 kHz, hop 512: one workgroup per CU, four segments of a clip, 145 KB of LDS."""
    L0 = 1323000
    rc, p = library_plan(L0, 512, (192,) * 4, L0 // 512 + 1, 0, n_cus=256, n_clips=64)
    assert rc == 0
    assert (p.frames_per_step, p.blocks_per_tile, p.n_segments) == (8, 2, 4)
    assert list(p.ring_rows[:4]) == [256, 128, 64, 32] and list(p.lookahead[:4]) == [1152, 512, 192, 32]
    assert list(p.contract_wave[:4]) == [4, 5, 6, 7]
    rc, p = library_plan(165375, 64, (0, 192, 192, 192, 192), L0 // 512 + 1, 0, n_cus=256, n_clips=64)
    assert rc == 0 and p.frames_per_step == 64 and p.lds_bytes <= 160 * 1024


@pytest.mark.parametrize("kw,why", [
    (dict(L0=40000, hop0=1024, K=(192, 192)), "hop"),           # hop > 512
    (dict(L0=40000, hop0=96, K=(192, 192)), "hop"),             # 4096 % hop
    (dict(L0=40000, hop0=512, K=(192, 192, 192, 192, 192)), "four"),  # five banks
    (dict(L0=3000, hop0=512, K=(192, 192)), "short"),           # a tile would touch both clip ends
    (dict(L0=40000, hop0=512, K=(320, 192)), "kernel"),         # kernel > 256
])
def test_unsupported_shapes_are_refused(kw, why):
    rc, _ = library_plan(kw["L0"], kw["hop0"], kw["K"], kw["L0"] // kw["hop0"] + 1, 0)
    assert rc == _abi.E_UNSUPPORTED, (rc, why)
    assert why in _abi.load().mispec_last_error().decode()
