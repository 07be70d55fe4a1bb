"""Shared host logic of the constant-Q family: per-call row scales, kernel supports and
the octave recursion of CQT2010v2 / VQT (reference: cqt.py:1070-1139, vqt.py:143-215)."""
import warnings

import numpy as np

import torch

from .. import engine
from ..basis import decimated_length


def output_epilogue(output_format):
    if output_format == "Magnitude":
        return engine.EPI_MAGNITUDE
    if output_format == "Complex":
        return engine.EPI_COMPLEX
    if output_format == "Phase":
        return engine.EPI_PHASE_COSSIN
    return None


def normalisation_scale(lenghts, normalization_type, extra=1.0):
    """Per-bin multiplier applied to (re, im) before the epilogue: sqrt(lenghts) ('librosa'),
    1 ('convolutional') or 2 ('wrap'), times ``extra`` (the early down-sampling factor)."""
    if normalization_type == "librosa":
        s = torch.sqrt(lenghts)
    elif normalization_type == "convolutional":
        s = torch.ones_like(lenghts)
    elif normalization_type == "wrap":
        s = torch.full_like(lenghts, 2.0)
    else:
        raise ValueError(
            "The normalization_type %r is not part of our current options." % normalization_type
        )
    if extra != 1.0:
        # same association order as the reference: (CQT * factor) * sqrt(lenghts)
        s = s * float(extra)
    return s.to(torch.float32).contiguous()


class SupportCache:
    """[start, stop) of the non-zero taps of every kernel row, recomputed whenever the
    kernel tensors are replaced or modified in place (see ``engine.DerivedCache``)."""

    def __init__(self, by_memory=False):
        self._cache = engine.DerivedCache(by_memory)

    @staticmethod
    def _build(real, imag):
        nz = (real.reshape(real.shape[0], -1) != 0) | (imag.reshape(imag.shape[0], -1) != 0)
        K = nz.shape[1]
        idx = torch.arange(K, device=nz.device)
        big = torch.where(nz, idx, torch.full_like(idx, K))
        small = torch.where(nz, idx + 1, torch.zeros_like(idx))
        start = big.min(dim=1).values
        stop = small.max(dim=1).values
        start = torch.minimum(start, stop)
        sup = torch.stack((start, stop), 1).to(torch.int32).contiguous()
        # host copy for the library's launch planning (mispec.h: row_support_host); one
        # synchronising copy per kernel bank, kept on the tensor that engine.framed_gemm receives
        sup.host_copy = sup.cpu().numpy().copy()
        return sup

    def get(self, real, imag):
        return self._cache.get((real, imag), lambda: self._build(real, imag))

    def host(self, real, imag):
        """The same (n_bins, 2) int32 values in host memory (``row_support_host``)."""
        return self.get(real, imag).host_copy


def zero_margin(re, im):
    """Taps that can be cut off BOTH ends of a bank ``(rows, K)`` without touching a non-zero
    coefficient: a multiple of 16, the kernel stays centred, a multiple of 32 wide and at least 32
    taps (0 for widths that are not a multiple of 32).  One host read per call."""
    K = re.shape[1]
    if K % 32 or K < 64 or re.shape[0] == 0:
        return 0
    idx = torch.nonzero(((re != 0) | (im != 0)).any(0)).flatten()
    if not idx.numel():
        return 0
    m = min(int(idx[0]), K - 1 - int(idx[-1])) // 16 * 16
    return max(0, min(m, (K - 32) // 32 * 16))


class OctaveCache:
    """Per-module derived operands of the octave recursion on the fused kernel (``precision``
    "bf16x3" / "f16x3"): the split planes of every octave's bank rows (see ``engine.DerivedCache``:
    rebuilt when the buffers change)."""

    def __init__(self, by_memory=False):
        self._banks = {}
        self._by_memory = by_memory  # (see engine.DerivedCache: the custom ops' instances)

    def scale(self, lenghts, normalization_type, factor):
        """``normalisation_scale`` of the module, built once (a forward must not launch the
        sqrt / multiply slivers every call) and rebuilt when ``lenghts`` changes."""
        c = self.__dict__.setdefault("_scale", engine.DerivedCache(self._by_memory))
        return c.get((lenghts,), lambda: normalisation_scale(lenghts, normalization_type, factor),
                     extra=(normalization_type, float(factor)))

    def fir_headroom_bits(self, lowpass):
        """ceil(log2(sum |taps|)): what one anti-alias FIR can add to the magnitude of a signal, in
        bits (fp16 operands of the f16x3 arithmetic keep that much room per level).  One host read
        per filter."""
        import math

        c = self.__dict__.setdefault("_gain", engine.DerivedCache(self._by_memory))
        return c.get((lowpass,), lambda: max(0, math.ceil(math.log2(max(float(lowpass.abs().sum()), 1e-30)))))

    def fir_gain_log2(self, lowpass):
        """log2(sum |taps|), unrounded: k cascaded FIRs add at most ceil(k * this) bits (the streaming
        kernel keeps the room of a whole launch, not of k roundings).  One host read per filter."""
        import math

        c = self.__dict__.setdefault("_gain_log2", engine.DerivedCache(self._by_memory))
        return c.get((lowpass,), lambda: max(0.0, math.log2(max(float(lowpass.abs().sum()), 1e-30))))

    def bank(self, i, kr, ki, first, precision="bf16x3"):
        """(split planes, kernel width) of octave ``i``'s rows ``first:``.  The reference's kernels
        sit centred in a power-of-two width and the longest of an octave spans ~0.69 of it: equal
        zero margins (multiples of 16 taps) are cut off both ends -- the same frames, centred as
        before, on a narrower kernel (256 -> 192 taps for the reference's banks: the fused kernel
        then keeps 6 instead of 8 steps of kernel rows in registers)."""
        c = self._banks.setdefault((i, precision), engine.DerivedCache(self._by_memory))

        def build():
            r = kr.reshape(kr.shape[0], -1)[first:]
            im = ki.reshape(ki.shape[0], -1)[first:]
            K, m = r.shape[1], zero_margin(r, im)
            if m:
                r, im = r[:, m:K - m].contiguous(), im[:, m:K - m].contiguous()
            split = engine.split_basis_f16 if precision == "f16x3" else engine.split_basis
            return split(r.contiguous(), im.contiguous()), K - 2 * m

        return c.get((kr, ki), build, extra=first)


def _stream_chain(x, octs, lowpass, epi, im_sign, eps, out, cache, precision):
    """The leading octaves on ``engine.octave_stream``: launches of up to four octaves (the first of a
    follow-up launch is the deepest level of the one before, re-read from HBM without a bank).  Returns
    (number of octaves done, the fp32 signal of the last octave done); (0, x) when the first launch is not
    served."""
    import math

    done, xd = 0, x
    gain = cache.fir_gain_log2(lowpass) if precision == "f16x3" else 0.0
    while done < len(octs):
        first = done == 0
        base = 0 if first else done - 1
        top = min(base + (4 if first else 5), len(octs))
        levels = []
        for i in range(base, top):
            o = octs[i]
            if i == base and not first:
                levels.append(None)
                continue
            split, k_eff = cache.bank(i, o["kr"], o["ki"], o["first"], precision)
            levels.append(dict(split=split, n_bins=o["rows"], kernel=k_eff, row_offset=o["row0"],
                               pad_mode=o["mode"], row_scale=o["scale"]))
        last = octs[top - 1]
        x_last = None
        if top < len(octs):
            # (rows a multiple of 4 floats apart: the next launch streams them in with 16-byte LDS-direct loads)
            x_last = torch.empty((x.shape[0], (last["L"] + 3) // 4 * 4), dtype=torch.float32,
                                 device=x.device)[:, :last["L"]]
        ok = engine.octave_stream(xd, levels, hop=octs[base]["hop"], n_frames=out.shape[2], taps=lowpass,
                                  epilogue=epi, im_sign=im_sign, eps=eps, out=out, x_last=x_last,
                                  precision=precision,
                                  fir_headroom_bits=math.ceil((len(levels) - 1) * gain - 1e-9))
        if not ok:
            break
        done = top
        if x_last is None:
            break
        xd = x_last
    return done, xd


def _fused_chain(x, octs, lowpass, epi, im_sign, eps, out, cache, precision="bf16x3", done=0):
    """Run as many leading octaves as possible through ``engine.octave_pyramid`` (three levels per
    launch, the deepest level of a launch feeding the next).  Returns (number of octaves done, the
    fp32 signal of the last octave done).  "f16x3": the clips' largest |sample| is found once for x
    and gathered by every launch for the level it hands on (two ping-pong buffers)."""
    xd = x
    f16 = precision == "f16x3"
    # (one zeroed buffer per launch boundary: the library's atomic maxima land in zeroed words)
    absmax = torch.zeros((len(octs) // 2 + 2, 32 * x.shape[0]), dtype=torch.int32, device=x.device) if f16 else None
    gain_bits = cache.fir_headroom_bits(lowpass) if f16 else 0
    launch = 0
    while done < len(octs):
        first = done == 0
        # levels of this launch: octave `base` (contracted only in the first launch: later it was the
        # deepest level of the previous one) and up to two more
        base = 0 if first else done - 1
        top = min(base + 3, len(octs))
        levels = []
        for i in range(base, top):
            o = octs[i]
            if i == base and not first:
                levels.append(None)
                continue
            split, k_eff = cache.bank(i, o["kr"], o["ki"], o["first"], precision)
            levels.append(dict(split=split, n_bins=o["rows"], kernel=k_eff, row_offset=o["row0"],
                               pad_mode=o["mode"], row_scale=o["scale"]))
        last = octs[top - 1]
        x_last = None
        if top < len(octs):  # someone will need the deepest level
            x_last = torch.empty((x.shape[0], last["L"]), dtype=torch.float32, device=x.device)
        extra = {}
        if f16:
            extra = dict(precision="f16x3", fir_headroom_bits=gain_bits * (len(levels) - 1),
                         absmax_in=absmax[launch], absmax_in_ready=launch > 0,
                         absmax_out=absmax[launch + 1] if x_last is not None else None)
        ok = engine.octave_pyramid(xd, levels, hop=octs[base]["hop"], n_frames=out.shape[2],
                                   taps=lowpass, epilogue=epi, im_sign=im_sign, eps=eps, out=out,
                                   x_last=x_last, **extra)
        if not ok:
            break
        launch += 1
        done = top
        if x_last is None:
            break
        xd = x_last
    return done, xd


def octave_recursion(x, banks, lenghts, hop, n_bins, lowpass, downsample_factor, pad_mode,
                     output_format, normalization_type, trainable, supports=None, graph=False,
                     scale=None, im_sign=-1.0, precision="fp32", cache=None):
    """Top-down octave loop.  ``banks[i] = (real_i, imag_i)`` (i = 0: top octave); each octave
    halves the signal with the anti-alias FIR kernel and halves the hop, and its framed
    contraction writes straight into its row block of the final ``(B, n_bins, T[, 2])``
    tensor (the reference's growing ``torch.cat`` is gone).

    ``precision="bf16x3"`` (and no autograd graph): the fused kernel keeps the decimated signals
    in LDS and contracts every octave from there (``engine.octave_pyramid``); octaves it does not
    serve (hop below 4 samples at the bottom) and ``precision="fp32"`` run one FIR decimation +
    one (grouped) contraction per octave on the exact fp32 kernels."""
    epi = output_epilogue(output_format)
    # the single-op route of torch.compile carries neither a precomputed scale nor an imaginary sign:
    # it is CQT2010v2 / VQT's call (scale from normalization_type, im_sign -1, a module cache)
    single_op_ok = scale is None and im_sign == -1.0 and cache is not None
    if scale is None:
        # (under torch.compile / export the tensors are fake: no cache look-ups, the ops are traced)
        scale = (cache.scale(lenghts, normalization_type, downsample_factor)
                 if cache is not None and not engine.compiling()
                 else normalisation_scale(lenghts, normalization_type, downsample_factor))
    if epi is None:
        return None
    n_oct = len(banks)
    n_filters = banks[0][0].shape[0]
    total_rows = n_oct * n_filters
    drop = total_rows - n_bins  # rows cut from the bottom octave by CQT[:, -n_bins:]
    nt = lowpass.numel()
    # ---- per-octave geometry (the lengths follow from conv1d's formula: no kernel needed)
    octs = []
    L = x.shape[-1]
    T_ref = None
    for i, (kr, ki) in enumerate(banks):
        if i > 0:
            hop = hop // 2
            L = decimated_length(L, nt, 2)
            if L <= 0:
                raise RuntimeError(
                    "Calculated padded input size per channel: (%d). Kernel size: (%d). "
                    "Kernel size can't be greater than actual input size" % (L + 2 * ((nt - 1) // 2), nt))
        K = kr.shape[-1]
        pad = K // 2
        mode = engine.pad_mode_id(pad_mode)
        if mode == engine.PAD_REFLECT and pad >= L:
            warnings.warn(
                f"\ninput size = {(x.shape[0], 1, L)}\tkernel size = {K}\n"
                "padding with reflection mode might not be the best choice, try using constant padding",
                UserWarning,
            )
            mode = engine.PAD_ZERO
        T = engine.n_frames(L, K, hop, pad)
        if T_ref is None:
            T_ref = T
        elif T != T_ref:
            raise RuntimeError(
                "Sizes of tensors must match except in dimension 1. Expected size %d but got "
                "size %d for octave %d (hop_length must stay divisible while it is halved)"
                % (T_ref, T, i)
            )
        # rows of this octave inside the concatenated bank, after the bottom cut
        row0 = (n_oct - 1 - i) * n_filters - drop
        first = 0
        if row0 < 0:
            first = -row0
            row0 = 0
        rows = n_filters - first
        octs.append(dict(i=i, kr=kr, ki=ki, K=K, L=L, hop=hop, pad=pad, mode=mode, row0=row0,
                         first=first, rows=rows,
                         scale=scale[row0:row0 + rows] if rows > 0 else None))
    while octs and octs[-1]["rows"] <= 0:  # octaves cut away entirely
        octs.pop()
    eps = 1e-8 if trainable else 0.0
    if engine.compiling() and not graph and not trainable and single_op_ok:
        # torch.compile with frozen kernels: the whole recursion as one op, which at run time is the
        # eager path below (fused pyramid kernel / grouped contractions) with cached operands
        from .. import ops

        return ops.octave_recursion(x, [b[0] for b in banks], [b[1] for b in banks], lenghts, lowpass,
                                    int(octs[0]["hop"]) if octs else int(hop), int(n_bins), float(downsample_factor),
                                    str(pad_mode), str(output_format), str(normalization_type), str(precision))
    if graph or engine.compiling():
        # the reference's own structure (cqt.py:1091-1105): one contraction per octave through
        # autograd (or, under torch.compile, through the custom ops), rows concatenated with the
        # lowest octave first
        blocks, xd = [], x
        for o in octs:
            if o["i"] > 0:
                xd = engine.fir_decimate_autograd(xd, lowpass, 2)
            blocks.append((o["row0"], engine.framed_gemm_autograd(
                xd, o["kr"].reshape(n_filters, -1)[o["first"]:], o["ki"].reshape(n_filters, -1)[o["first"]:],
                hop=o["hop"], pad=o["pad"], pad_mode=o["mode"], epilogue=epi, im_sign=im_sign, eps=eps,
                row_scale=o["scale"].contiguous(), precision="fp32")))
        return torch.cat([b for _, b in sorted(blocks, key=lambda rb: rb[0])], 1)
    two = epi in (engine.EPI_COMPLEX, engine.EPI_PHASE_COSSIN)
    shape = (x.shape[0], n_bins, T_ref, 2) if two else (x.shape[0], n_bins, T_ref)
    out = engine.alloc_out(shape, x.device)
    done, xd = 0, x
    if precision in ("bf16x3", "f16x3") and cache is not None and not trainable and x.is_cuda:
        if engine.octave_stream_enabled():
            done, xd = _stream_chain(x, octs, lowpass, epi, im_sign, eps, out, cache, precision)
        if done < len(octs):  # what the streaming kernel does not serve: the pyramid kernel from there
            done, xd = _fused_chain(xd, octs, lowpass, epi, im_sign, eps, out, cache, precision, done=done)
    launches = []  # the remaining per-octave contractions are independent: one grouped launch
    for o in octs[done:]:  # xd: the fp32 signal of the previous octave (x itself before octave 0)
        if o["i"] > 0:
            xd = engine.fir_decimate(xd, lowpass, 2)
        sup = sup_host = None
        if supports is not None and not trainable:
            sup = supports[o["i"]].get(o["kr"], o["ki"])[o["first"]:].contiguous()
            sup_host = np.ascontiguousarray(supports[o["i"]].host(o["kr"], o["ki"])[o["first"]:])
        launches.append((xd, o["kr"].reshape(n_filters, -1)[o["first"]:],
                         o["ki"].reshape(n_filters, -1)[o["first"]:], dict(
            hop=o["hop"], pad=o["pad"], pad_mode=o["mode"], epilogue=epi, im_sign=im_sign, eps=eps,
            row_scale=o["scale"].contiguous(), row_support=sup, row_support_host=sup_host, out=out,
            out_rows_total=n_bins, out_row_offset=o["row0"], precision="fp32")))
    engine.framed_gemm_group(launches)
    return out


def early_decimate(x, taps, factor):
    return engine.fir_decimate_autograd(x, taps, int(factor))


__all__ = ["output_epilogue", "normalisation_scale", "SupportCache", "OctaveCache", "octave_recursion",
           "early_decimate", "decimated_length"]
