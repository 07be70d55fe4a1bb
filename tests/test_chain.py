"""The chain kernel of CQT1992v2's fp32 contraction (csrc/cqt_chain.hip; `basis_chain` of mispec_framed_gemm_args): the
bank as a stream of MFMA fragments, the frames' samples in LDS delay lines.  Its contract is BIT IDENTITY with the library's
sequential reference kernel (one thread per output, fmaf over the taps in ascending order = the reference's conv1d,
cqt.py:749-750): tests/test_reference_order.py holds the shipped module to the reference's own fixtures; this file holds the
kernel to the reference kernel over the shapes that exercise its plan (row sets, batches per ring row, every hop it takes,
the three padding modes with their mirror-image fast paths, partial column tiles, idle waves) and checks what the host side
promises without a GPU (which banks it takes, that the layout does not depend on the hop)."""
import ctypes

import numpy as np
import pytest
import torch

from nnaudio_amd import _abi


def cqt_like_bank(F, K, rng, ratio=12.0, min_len=8):
    """random taps with centred supports that shrink by 2^(1/ratio) per row: the shape of a CQT bank (utils.py:457-469)"""
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sup = np.zeros((F, 2), np.int32)
    for f in range(F):
        ln = max(min_len, int(K * 2.0 ** (-f / ratio)))
        lo = (K - ln) // 2
        sup[f] = (lo, lo + ln)
        wr[f, :lo] = wr[f, lo + ln:] = 0.0
        wi[f, :lo] = wi[f, lo + ln:] = 0.0
    return wr, wi, sup


def chain_bytes(sup, K):
    sup = np.ascontiguousarray(sup, dtype=np.int32)
    return _abi.load().mispec_basis_chain_bytes(sup.ctypes.data, sup.shape[0], K)


def test_plan_takes_nested_banks_and_refuses_the_others():
    rng = np.random.default_rng(0)
    _, _, sup = cqt_like_bank(84, 4096, rng)
    n = chain_bytes(sup, 4096)
    assert n > 0 and n % 1024 == 0
    assert chain_bytes(sup, 4096) == n  # (a function of the supports alone)
    # the bricks hold the taps of every 16-row tile's support, rounded to 16 taps: between the dense bank's size over
    # the useful taps and 1.5 x that
    useful = int((sup[:, 1] - sup[:, 0]).sum()) * 2 * 4
    assert useful < n < 1.5 * useful + 200 * 1024
    # supports that do not nest by length (a later, longer row sticking out): refused, the tile kernels serve such banks
    bad = sup.copy()
    bad[40] = (0, 100)
    assert chain_bytes(bad, 4096) == _abi.E_UNSUPPORTED
    # malformed supports
    worse = sup.copy()
    worse[3] = (50, 40)
    assert chain_bytes(worse, 4096) == _abi.E_INVALID
    assert _abi.load().mispec_basis_chain_bytes(None, 84, 4096) == _abi.E_INVALID
    # more 16-row tiles than the row sets hold (8 sets of 10 tiles = 640 bins)
    _, _, big = cqt_like_bank(700, 2048, rng, ratio=200.0)
    assert chain_bytes(big, 2048) == _abi.E_UNSUPPORTED


def test_empty_rows_and_tiny_banks_plan():
    sup = np.zeros((5, 2), np.int32)  # no non-zero tap at all: one brick of padding behind the header
    assert chain_bytes(sup, 256) > 0
    sup = np.array([[10, 200], [60, 140]], np.int32)
    assert chain_bytes(sup, 256) > 0


def test_pack_refuses_a_wrong_size_without_touching_the_device():
    lib = _abi.load()
    sup = np.array([[10, 200], [60, 140]], np.int32)
    rc = lib.mispec_chain_basis_f32(ctypes.c_void_p(8), ctypes.c_void_p(8), 256, 2, 256, sup.ctypes.data, ctypes.c_void_p(8), 12, None)
    assert rc == _abi.E_INVALID and b"dst_bytes" in lib.mispec_last_error()


CASES = [  # F, K, hop, B, L, pad mode, epilogue
    (84, 4096, 256, 2, 20000, "reflect", "complex"),
    (84, 4096, 512, 3, 30000, "reflect", "complex"),    # 11 tiles: two row sets
    (130, 2048, 512, 1, 30000, "reflect", "magnitude"),  # 17 tiles: three row sets
    (7, 1024, 64, 5, 3000, "zero", "complex"),           # one half-empty tile, the shortest ring rows
    (24, 1000, 128, 2, 9000, "zero", "complex"),         # K % 16 != 0
    (40, 2048, 192, 2, 12345, "reflect", "complex"),     # a hop that is not a power of two: 18 ring rows
    (33, 512, 320, 4, 7001, "none", "complex"),          # center=False
    (84, 4096, 384, 1, 2100, "reflect", "phase"),        # a clip barely longer than the padding: every block is mirrored
    (12, 640, 448, 9, 5000, "zero", "magnitude"),        # 9 clips x 2 column tiles: idle waves in the last group
    # shared delay lines (round 6): runs of column tiles of one clip inside a workgroup, workgroups across clip boundaries
    (48, 2048, 256, 5, 20000, "reflect", "complex"),     # hop 256: 5 tiles per clip -> runs of 4, 1 + 3, 2 + 2, 3 + 1; register-path prologue
    (84, 4096, 512, 5, 20000, "reflect", "magnitude"),   # hop 512: 3 tiles per clip -> runs of 3 + 1, 2 + 2, 1 + 3
    (20, 1024, 512, 3, 70001, "zero", "complex"),        # 9 tiles per clip: runs of 4, 4, 1 + 3 ...; zero padding through the register path
    (7, 256, 256, 3, 200, "reflect", "complex"),         # clips shorter than a 256-block: the DMA prologue
    (16, 512, 256, 2, 512, "none", "complex"),           # one frame per clip, center=False, L = two blocks exactly
]


@pytest.mark.gpu
@pytest.mark.parametrize("F,K,hop,B,L,pad_mode,epi", CASES)
def test_chain_kernel_is_the_sequential_chain(F, K, hop, B, L, pad_mode, epi):
    from nnaudio_amd import engine

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(F * K + hop)
    wr, wi, sup = cqt_like_bank(F, K, rng)
    x = torch.as_tensor(rng.standard_normal((B, L)).astype(np.float32)).to(dev)
    wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
    chain = engine.chain_basis_f32(wr_d, wi_d, sup)
    assert chain is not None
    mode = {"reflect": engine.PAD_REFLECT, "zero": engine.PAD_ZERO, "none": engine.PAD_NONE}[pad_mode]
    e = {"complex": engine.EPI_COMPLEX, "magnitude": engine.EPI_MAGNITUDE, "phase": engine.EPI_PHASE_COSSIN}[epi]
    scale = torch.as_tensor(rng.uniform(0.5, 2.0, F).astype(np.float32)).to(dev)
    kw = dict(hop=hop, pad=0 if pad_mode == "none" else K // 2, pad_mode=mode, epilogue=e, im_sign=-1.0, precision="fp32", row_scale=scale)
    y = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, basis_chain=chain, **kw)
    ref = engine.framed_gemm(x, wr_d, wi_d, reference_kernel=True, **kw)
    assert torch.equal(y, ref), float((y - ref).abs().max())
    # ... and what the chain copy replaces computes the same bits (the tile kernels, round 5)
    y0 = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, **kw)
    assert torch.equal(y, y0)


@pytest.mark.gpu
def test_chain_kernel_writes_its_row_block_in_place():
    """out / out_rows_total / out_row_offset (the all-gather slot of nnaudio_amd.dist, octave assembly): the rows around the
    bank's block stay untouched"""
    from nnaudio_amd import engine

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    F, K, hop = 30, 1024, 256
    wr, wi, sup = cqt_like_bank(F, K, rng)
    x = torch.as_tensor(rng.standard_normal((3, 9000)).astype(np.float32)).to(dev)
    wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
    chain = engine.chain_basis_f32(wr_d, wi_d, sup)
    kw = dict(hop=hop, pad=K // 2, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_MAGNITUDE, precision="fp32",
              row_support=sup_d, row_support_host=sup, basis_chain=chain)
    y = engine.framed_gemm(x, wr_d, wi_d, **kw)
    big = torch.full((3, F + 9, y.shape[2]), -7.0, device=dev)
    engine.framed_gemm(x, wr_d, wi_d, out=big, out_rows_total=F + 9, out_row_offset=4, **kw)
    assert torch.equal(big[:, 4:4 + F], y)
    assert bool((big[:, :4] == -7.0).all()) and bool((big[:, 4 + F:] == -7.0).all())


@pytest.mark.gpu
def test_banks_and_hops_the_chain_kernel_does_not_take_run_on_the_tile_kernels():
    from nnaudio_amd import engine

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    F, K = 20, 512
    wr = rng.standard_normal((F, K)).astype(np.float32)
    wi = rng.standard_normal((F, K)).astype(np.float32)
    sup = np.zeros((F, 2), np.int32)
    for f in range(F):  # the first 8-bin tile lives in [0, 120), the others in [300, 500): the tiles' supports do not nest
        lo = int(rng.integers(0, 60)) if f < 8 else int(rng.integers(300, 400))
        hi = lo + int(rng.integers(20, 60)) if f < 8 else int(rng.integers(420, 500))
        sup[f] = (lo, hi)
        wr[f, :lo] = wr[f, hi:] = 0.0
        wi[f, :lo] = wi[f, hi:] = 0.0
    wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
    assert engine.chain_basis_f32(wr_d, wi_d, sup) is None
    x = torch.as_tensor(rng.standard_normal((2, 6000)).astype(np.float32)).to(dev)
    kw = dict(hop=128, pad=K // 2, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_COMPLEX, precision="fp32")
    y = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, basis_chain=None, **kw)
    assert torch.equal(y, engine.framed_gemm(x, wr_d, wi_d, reference_kernel=True, **kw))
    # ragged rows whose TILES nest (each tile's union of supports inside the longer tile's): taken, the same bits
    for f in range(F):
        lo = int(rng.integers(0, 200)) if f < 8 else int(rng.integers(200, 250))
        hi = int(rng.integers(330, 512)) if f < 8 else int(rng.integers(260, 330))
        if f in (0, 8):
            lo, hi = (0, 512) if f == 0 else (200, 330)
        sup[f] = (lo, hi)
        wr[f] = rng.standard_normal(K).astype(np.float32)
        wi[f] = rng.standard_normal(K).astype(np.float32)
        wr[f, :lo] = wr[f, hi:] = 0.0
        wi[f, :lo] = wi[f, hi:] = 0.0
    wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
    chain = engine.chain_basis_f32(wr_d, wi_d, sup)
    assert chain is not None
    y = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, basis_chain=chain, **kw)
    assert torch.equal(y, engine.framed_gemm(x, wr_d, wi_d, reference_kernel=True, **kw))
    # a nested bank at a hop the kernel does not take (not a multiple of 64): the copy is ignored, the bits are the same
    wr, wi, sup = cqt_like_bank(F, K, rng)
    wr_d, wi_d, sup_d = (torch.as_tensor(t).to(dev) for t in (wr, wi, sup))
    chain = engine.chain_basis_f32(wr_d, wi_d, sup)
    kw = dict(hop=100, pad=K // 2, pad_mode=engine.PAD_REFLECT, epilogue=engine.EPI_COMPLEX, precision="fp32")
    y = engine.framed_gemm(x, wr_d, wi_d, row_support=sup_d, row_support_host=sup, basis_chain=chain, **kw)
    ref = engine.framed_gemm(x, wr_d, wi_d, reference_kernel=True, **kw)
    assert torch.equal(y, ref)


@pytest.mark.gpu
def test_cqt1992v2_module_takes_the_chain_kernel_and_keeps_its_bits():
    """the shipped module (fp32 default) with and without the chain copy: the same bits, for every output format, and the
    copy follows the buffers (load_state_dict / in-place edits rebuild it)"""
    from nnaudio_amd.features import CQT1992v2

    dev = torch.device("cuda:0")
    x = torch.randn(3, 30000, generator=torch.Generator().manual_seed(1)).to(dev)
    for fmt in ("Magnitude", "Complex", "Phase"):
        mod = CQT1992v2(sr=22050, hop_length=256, fmin=55, n_bins=60, bins_per_octave=12, output_format=fmt, verbose=False).to(dev)
        with torch.no_grad():
            y = mod(x)
            mod.chain = False
            y0 = mod(x)
            mod.chain = True
        assert torch.equal(y, y0), fmt
    with torch.no_grad():
        mod.cqt_kernels_real.mul_(0.5)
        mod.cqt_kernels_imag.mul_(0.5)
        y2 = mod(x)
    assert torch.allclose(y2, y, rtol=0, atol=0)  # (phase: cos / sin of the same angle -- scaling both parts by 2^-1 is exact)
