"""Index-level model (numpy, float64) of the streaming octave kernel (csrc/octave_stream.inl).

The kernel walks a segment of one clip in steps of `chunk` level-0 samples and keeps, per level
l = 0 .. D-1 of the recursion x_{l+1} = downsampling_by_2(x_l) (utils.py:73-124), a RING of rows of 64
samples in LDS.  In step g

    level 0        chunk g is resident; chunk g+1 (loaded a step earlier) is split and written
    FIR l -> l+1   produces block g - l of level l+1 (32-output columns; column at absolute
                   position P reads the five input rows that start at 2 P - 128)
    contraction    level l contracts the 16-frame tiles that block g - l completes
                   (get_cqt_complex, utils.py:498-521), edge frames from a small patch that holds
                   the mirrored samples (nn.ReflectionPad1d)

with ONE barrier per step: every read of a step touches data written in an earlier step.  This file
restates that schedule sample by sample -- rings with stale NaN garbage, absolute positions, the
patches -- and `check()` compares what it produces with the plain recursion.  It is the executable
specification the C++ plan (mispec_octave_stream_plan) and the kernel follow; tests/test_octave_stream_cpu.py
runs it against the library's plan.  No GPU needed.
"""
import numpy as np

ROW = 64        # samples per ring row
COL = 32        # outputs per FIR column
NTAP_WIN = 320  # input samples a column reads (five rows)


def decimated_length(L, n_taps=256):
    return (L + 2 * ((n_taps - 1) // 2) - n_taps) // 2 + 1


def pow2_at_least(v):
    p = 1
    while p < v:
        p *= 2
    return p


def plan_stream(L0, hop0, K, n_frames, n_taps=256, chunk=4096, n_seg=1, warm=2):
    """K[l] = kernel width of level l's bank (0: level not contracted).  Returns the geometry."""
    D = len(K)
    assert chunk % (COL << (D - 1)) == 0 and chunk % hop0 == 0
    nf = chunk // hop0
    assert nf >= 8 and nf % 8 == 0
    hop = [hop0 >> l for l in range(D)]
    assert all(h >= 4 and (hop0 % (1 << l)) == 0 for l, h in enumerate(hop))
    L = [L0]
    for l in range(1, D):
        L.append(decimated_length(L[-1], n_taps))
    blk = [chunk >> l for l in range(D)]
    # look-ahead of every level's blocks: c[l] = 2 c[l+1] + 128, c[l] >= K[l]/2 - hop[l]
    cD = 0
    while True:
        c = [0] * D
        c[D - 1] = cD
        for l in range(D - 2, -1, -1):
            c[l] = 2 * c[l + 1] + 128
        if all(c[l] >= K[l] // 2 - hop[l] for l in range(D) if K[l]):
            break
        cD += COL
    # frames per contraction tile: 16; a tile is contracted by the block that completes it
    tiles_span = max(1, 16 // nf)  # blocks a tile spans (nf = 8: 2)
    # live range of a ring in step g (samples), see DESIGN: [oldest read, end of the block being written)
    rows = []
    for l in range(D):
        # newest: level 0 is being written chunk g+1; level l >= 1 block g-l+1
        newest_end = blk[l] * 2 + c[l]  # relative to blk*(g-l)
        # oldest: contraction of block g-l reaches back to the tile's first frame - K/2; the FIR to c - 255
        back_c = (tiles_span - 1) * blk[l] + (K[l] // 2 if K[l] else 0)
        back_f = 255 - c[l] if l < D - 1 else 0
        live = newest_end + max(back_c, back_f, 0)
        rows.append(pow2_at_least((live + ROW - 1) // ROW + 1))
    n_blocks = (n_frames + nf - 1) // nf
    n_blocks = (n_blocks + tiles_span - 1) // tiles_span * tiles_span  # whole tiles
    # the deepest level must be covered by the blocks (x_last is written by them)
    while blk[D - 1] * n_blocks + c[D - 1] < L[D - 1]:
        n_blocks += tiles_span
    # segments: consecutive blocks, boundaries on whole tiles
    per = (n_blocks + n_seg - 1) // n_seg
    per = (per + tiles_span - 1) // tiles_span * tiles_span
    segs = []
    b = 0
    while b < n_blocks:
        segs.append((b, min(n_blocks, b + per)))
        b += per
    return dict(D=D, nf=nf, chunk=chunk, hop=hop, L=L, K=list(K), blk=blk, c=c, rows=rows, n_blocks=n_blocks,
                segs=segs, warm=warm, n_frames=n_frames, n_taps=n_taps, tiles_span=tiles_span)


class Ring:
    def __init__(self, rows):
        self.n = rows * ROW
        self.v = np.full(self.n, np.nan)        # stale garbage
        self.pos = np.full(self.n, -(1 << 60))  # absolute position a slot holds

    def write(self, p0, vals):
        idx = (np.arange(p0, p0 + len(vals))) % self.n
        self.v[idx] = vals
        self.pos[idx] = np.arange(p0, p0 + len(vals))

    def read(self, p0, n, strict=True):
        idx = (np.arange(p0, p0 + n)) % self.n
        if strict:
            bad = self.pos[idx] != np.arange(p0, p0 + n)
            assert not bad.any(), "ring read of position %d..: slot holds %s" % (p0, self.pos[idx][bad][:4])
        return self.v[idx]


def toeplitz(taps):
    """T[r, m] = taps[m - 2 r - shift], shift = 128 - dec_pad: 32 outputs x 320 inputs."""
    n = len(taps)
    shift = 128 - (n - 1) // 2
    T = np.zeros((COL, NTAP_WIN))
    for r in range(COL):
        for m in range(NTAP_WIN):
            k = m - 2 * r - shift
            if 0 <= k < n:
                T[r, m] = taps[k]
    return T


def run_segment(plan, seg, x, taps, banks, pad_reflect, out, x_last):
    """banks[l]: complex (n_rows, K[l]) or None.  out[l]: complex (n_rows, n_frames).  x_last: level D-1."""
    D, nf, chunk, hop, L, K, blk, c = (plan[k] for k in ("D", "nf", "chunk", "hop", "L", "K", "blk", "c"))
    T = toeplitz(taps)
    rings = [Ring(r) for r in plan["rows"]]
    b_a, b_e = seg
    warm = plan["warm"]
    n_frames = plan["n_frames"]
    span = plan["tiles_span"]

    def load_chunk(g):
        p0 = chunk * g + c[0]
        idx = np.arange(p0, p0 + chunk)
        v = np.where((idx >= 0) & (idx < L[0]), x[np.clip(idx, 0, L[0] - 1)], 0.0)
        rings[0].write(p0, v)

    g0 = b_a - warm
    load_chunk(g0)  # prologue
    for g in range(g0, b_e + D - 1):
        writes = []  # (level, p0, values): applied at the barrier
        # ---- FIR l -> l+1: block beta = g - l of level l+1
        for l in range(D - 1):
            beta = g - l
            if beta < b_a - warm or beta >= b_e:
                continue
            p_first = blk[l + 1] * beta + c[l + 1]
            for col in range(blk[l + 1] // COL):
                P = p_first + COL * col
                src = rings[l].read(2 * P - 128, NTAP_WIN, strict=False)  # stale slots show as NaN
                y = T @ src
                pos = np.arange(P, P + COL)
                y = np.where((pos >= 0) & (pos < L[l + 1]), y, 0.0)  # zeros outside the level
                writes.append((l + 1, P, y))
                if l + 1 == D - 1 and x_last is not None:
                    own_lo = 0 if b_a == 0 else blk[D - 1] * b_a + c[D - 1]
                    own_hi = blk[D - 1] * b_e + c[D - 1]
                    m = (pos >= own_lo) & (pos < own_hi) & (pos >= 0) & (pos < L[D - 1])
                    assert not np.isnan(y[m]).any(), "x_last from stale data"
                    x_last[pos[m]] = y[m]
        # ---- contraction of the tiles block g - l completes
        for l in range(D):
            if not K[l]:
                continue
            beta = g - l
            if beta < b_a or beta >= b_e:
                continue
            if (beta + 1) % span:
                continue
            f0 = (beta + 1 - span) * nf  # first frame of the tile(s)
            for tile0 in range(f0, (beta + 1) * nf, 16):
                half = K[l] // 2
                frames = np.arange(tile0, tile0 + 16)
                w = frames * hop[l] - half  # window starts
                edge_l = w < 0
                edge_r = (w + K[l] > L[l])
                patch = {}
                if pad_reflect:
                    if edge_l.any():
                        # rows -2 .. 3: positions [-128, K) -- [0, K) is inside every edge frame's reach, so
                        # it is resident; mirrored x(-p) for p in [-128, 0)
                        p0 = -128
                        raw = rings[l].read(0, K[l])
                        assert K[l] >= 129 or not (w < -(K[l] - 1)).any()
                        mir = np.array([raw[-p] if -p < K[l] else np.nan for p in range(-128, 0)])
                        patch["l"] = (p0, np.concatenate((mir, raw)))
                    if edge_r.any():
                        rho0 = int(w[edge_r].min()) >> 6  # the row of the first edge frame's window start
                        p0 = rho0 * ROW
                        assert L[l] + half <= p0 + 8 * ROW, "right patch: 8 rows do not reach L + K/2"
                        n_in = L[l] - p0
                        raw = rings[l].read(p0, n_in)
                        mir = np.array([rings[l].read(2 * (L[l] - 1) - p, 1)[0] for p in range(L[l], p0 + 8 * ROW)])
                        patch["r"] = (p0, np.concatenate((raw, mir)))
                    assert not (edge_l & edge_r).any(), "a frame touches both clip ends"
                for j, t in enumerate(frames):
                    if t >= n_frames:
                        continue
                    if pad_reflect and edge_l[j]:
                        p0, buf = patch["l"]
                        win = buf[w[j] - p0:w[j] - p0 + K[l]]
                    elif pad_reflect and edge_r[j]:
                        p0, buf = patch["r"]
                        assert w[j] >= p0 and w[j] - p0 + K[l] <= len(buf), "right patch too small"
                        win = buf[w[j] - p0:w[j] - p0 + K[l]]
                    else:
                        win = rings[l].read(w[j], K[l])
                    assert not np.isnan(win).any(), "frame %d level %d reads stale data" % (t, l)
                    out[l][:, t] = banks[l] @ win
        # ---- chunk g+1 lands in the level-0 ring (loaded a step ago, split and written in this step)
        for lv, P, y in writes:
            rings[lv].write(P, y)
        if g + 1 <= b_e - 1:
            load_chunk(g + 1)
    return


def reference(x, taps, banks, hop0, n_frames, pad_reflect):
    """The plain recursion (float64): zero-padded FIR decimation, reflect- or zero-padded frames."""
    D = len(banks)
    n = len(taps)
    pad = (n - 1) // 2
    outs, xs = [], [np.asarray(x, np.float64)]
    for l in range(D):
        xl = xs[-1]
        if banks[l] is not None:
            Kl = banks[l].shape[1]
            xp = np.pad(xl, Kl // 2, mode="reflect" if pad_reflect else "constant")
            h = hop0 >> l
            o = np.zeros((banks[l].shape[0], n_frames), complex)
            for t in range(n_frames):
                o[:, t] = banks[l] @ xp[t * h:t * h + Kl]
            outs.append(o)
        else:
            outs.append(None)
        if l + 1 < D:
            xp = np.pad(xl, pad)
            Ln = decimated_length(len(xl), n)
            xs.append(np.array([taps @ xp[2 * i:2 * i + n] for i in range(Ln)]))
    return outs, xs


def check(L0=40000, hop0=512, K=(192, 192, 192, 192, 192), n_seg=2, pad_reflect=True, seed=0, chunk=4096,
          plan=None, verbose=False):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal(L0)
    taps = rng.standard_normal(256) / 16
    n_frames = L0 // hop0 + 1
    banks = [None if not k else (rng.standard_normal((12, k)) + 1j * rng.standard_normal((12, k))) for k in K]
    if plan is None:
        plan = plan_stream(L0, hop0, list(K), n_frames, chunk=chunk, n_seg=n_seg)
    if verbose:
        print({k: v for k, v in plan.items()})
    out = [None if b is None else np.full((12, n_frames), np.nan, complex) for b in banks]
    x_last = np.full(plan["L"][-1], np.nan)
    for seg in plan["segs"]:
        run_segment(plan, seg, x, taps, banks, pad_reflect, out, x_last)
    ref, xs = reference(x, taps, banks, hop0, n_frames, pad_reflect)
    err = 0.0
    for l, (o, r) in enumerate(zip(out, ref)):
        if r is None:
            continue
        assert not np.isnan(o).any(), "level %d: frames not produced: %s" % (l, np.flatnonzero(np.isnan(o).any(0))[:8])
        err = max(err, np.abs(o - r).max() / np.abs(r).max())
    assert not np.isnan(x_last).any(), "x_last holes at %s" % np.flatnonzero(np.isnan(x_last))[:8]
    err = max(err, np.abs(x_last - xs[-1]).max() / np.abs(xs[-1]).max())
    return err, plan


if __name__ == "__main__":
    for kw in (dict(), dict(n_seg=1), dict(n_seg=3, L0=70001), dict(pad_reflect=False),
               dict(hop0=32, K=(0, 192, 192, 192), L0=9000, n_seg=2),
               dict(hop0=256, K=(256, 128, 64, 32), L0=50000)):
        e, p = check(**kw)
        print(kw, "err %.2e" % e, "rows", p["rows"], "c", p["c"], "segs", p["segs"])
