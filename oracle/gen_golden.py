#!/usr/bin/env python3
"""Generate the committed golden fixtures under ``tests/golden`` by running the
reference (read-only, ``/root/reference/Installation``) on this container's CPU.

TEST INFRASTRUCTURE.  The reference cannot travel to the GPU box, so everything the
``-m gpu`` tests and ``smoke()`` need from it is materialised here once:

  tests/golden/ref_ground_truths/*.npy   the ten arrays the reference's own
                                         tests/test_cqt.py:94-262 asserts (byte copies)
  tests/golden/inputs.npz                seeded waveforms shared by the cases
  tests/golden/cases.json                manifest: class, ctor kwargs, forward kwargs,
                                         input key, attributes, buffer digests
  tests/golden/forward.npz               reference forward outputs per case
  tests/golden/buffers.npz               reference state_dict buffers per case
                                         (full when small, strided sample + sha256 when big)

Run:  python oracle/gen_golden.py        (needs /root/reference; CPU only, ~1 min)
"""
import hashlib
import json
import os
import shutil
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference/Installation"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
FULL_LIMIT = 1 << 16  # buffers above 64 KiB are stored as digest + strided sample
SAMPLE_STRIDE = 97


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted at %s" % REF)
    sys.path.insert(0, REF)
    from nnAudio import features as R  # noqa: E402

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ---- (1) the reference's own asserted ground truths -------------------------
    gt_dst = os.path.join(OUT, "ref_ground_truths")
    os.makedirs(gt_dst, exist_ok=True)
    for sweep in ("log", "linear"):
        for name in ("1992-mag", "1992-complex", "1992-phase", "2010-mag", "2010-complex"):
            fn = "%s-sweep-cqt-%s-ground-truth.npy" % (sweep, name)
            shutil.copyfile(os.path.join(REF, "tests", "ground-truths", fn),
                            os.path.join(gt_dst, fn))
            os.chmod(os.path.join(gt_dst, fn), 0o644)

    # ---- (2) shared inputs ---------------------------------------------------
    rng = np.random.default_rng(20240923)
    inputs = {
        "x_1s22k": rng.standard_normal((2, 22050)).astype(np.float32),
        "x_short": rng.standard_normal((3, 4000)).astype(np.float32),
        "x_cfg1": rng.standard_normal((1, 16000)).astype(np.float32),
        "x_2s44k": (0.5 * rng.standard_normal((1, 88200))).astype(np.float32),
        "x_1d": rng.standard_normal((6000,)).astype(np.float32),
        "x_3d": rng.standard_normal((2, 1, 5000)).astype(np.float32),
    }
    # a full-scale sine mix in the 22 kHz slot family
    t = np.arange(22050) / 22050.0
    inputs["x_sines"] = np.stack(
        [np.sin(2 * np.pi * 440 * t) + 0.5 * np.sin(2 * np.pi * 3520 * t + 1.0),
         np.sin(2 * np.pi * 55 * t) * np.cos(2 * np.pi * 3 * t)]
    ).astype(np.float32)
    # spectrograms for the inverse transforms: the reference's own STFT of x_short
    with torch.no_grad():
        xs = torch.from_numpy(inputs["x_short"])
        X512 = R.STFT(n_fft=512, hop_length=128, verbose=False)(xs)
        inputs["X_512_128"] = X512.numpy()
        inputs["X_512_128_full"] = torch.cat(
            (X512, torch.stack((X512[:, 1:-1, :, 0].flip(1), -X512[:, 1:-1, :, 1].flip(1)), -1)), 1).numpy()
        inputs["X_256_64_nocenter"] = R.STFT(n_fft=256, hop_length=64, center=False, verbose=False)(xs).numpy()
        inputs["X_512_100_hamming"] = R.STFT(n_fft=512, hop_length=100, window="hamming",
                                             verbose=False)(xs).numpy()
    np.savez_compressed(os.path.join(OUT, "inputs.npz"), **inputs)

    # ---- (3) cases -----------------------------------------------------------
    C = []

    def case(name, cls, ctor, x, fwd=None, attrs=(), method="forward"):
        C.append(dict(name=name, cls=cls, ctor=ctor, input=x, fwd=fwd or {}, attrs=list(attrs),
                      method=method))

    stft_attrs = ("stride", "n_fft", "freq_bins", "pad_amount", "win_length")
    # STFT: cfg1 + the reference test grid flavours (tests/parameters.py) + odd corners
    case("stft_cfg1_complex", "STFT", dict(n_fft=512, hop_length=128), "x_cfg1", attrs=stft_attrs)
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("stft_512_128_%s" % fmt.lower(), "STFT",
             dict(n_fft=512, hop_length=128, output_format=fmt), "x_short", attrs=stft_attrs)
    case("stft_fwd_override", "STFT", dict(n_fft=512, hop_length=128), "x_short",
         fwd=dict(output_format="Magnitude"))
    case("stft_1024_hamming_h128", "STFT",
         dict(n_fft=1024, hop_length=128, window="hamming", output_format="Magnitude"), "x_short")
    case("stft_1024_ones_h128", "STFT",
         dict(n_fft=1024, hop_length=128, window="ones", output_format="Complex"), "x_short")
    case("stft_2048_hann_h512", "STFT",
         dict(n_fft=2048, hop_length=512, window="hann", output_format="Magnitude"), "x_1s22k")
    # n_fft = 4096 (round 5: the composite instance of the FFT route): default hop with reflect padding, and an even hop that
    # does not divide n_fft with the Complex output
    case("stft_4096_hann_h1024", "STFT",
         dict(n_fft=4096, hop_length=1024, window="hann", output_format="Magnitude"), "x_1s22k")
    case("stft_4096_complex_h600", "STFT",
         dict(n_fft=4096, hop_length=600, window="hamming", output_format="Complex"), "x_1s22k")
    case("stft_256_defaulthop", "STFT", dict(n_fft=256, hop_length=None, output_format="Complex"),
         "x_short")
    case("stft_winlen400", "STFT",
         dict(n_fft=512, win_length=400, hop_length=128, output_format="Complex"), "x_short")
    case("stft_winlen900_h256", "STFT",
         dict(n_fft=1024, win_length=900, hop_length=256, output_format="Magnitude"), "x_short")
    case("stft_nocenter", "STFT",
         dict(n_fft=512, hop_length=160, center=False, output_format="Complex"), "x_short")
    case("stft_constpad", "STFT",
         dict(n_fft=512, hop_length=128, pad_mode="constant", output_format="Complex"), "x_short")
    case("stft_oddhop", "STFT", dict(n_fft=400, hop_length=37, output_format="Magnitude"), "x_short")
    case("stft_freqbins100", "STFT",
         dict(n_fft=512, hop_length=128, freq_bins=100, output_format="Magnitude"), "x_short")
    for fs in ("linear", "log", "log2"):
        case("stft_scale_%s" % fs, "STFT",
             dict(n_fft=512, hop_length=128, freq_bins=120, freq_scale=fs, fmin=60, fmax=7000,
                  sr=16000, output_format="Complex"), "x_short", attrs=("bins2freq", "bin_list"))
    case("stft_trainable_mag", "STFT",
         dict(n_fft=512, hop_length=128, trainable=True, output_format="Magnitude"), "x_short")
    case("stft_istft_buffers", "STFT", dict(n_fft=256, hop_length=64, iSTFT=True), "x_short")
    case("stft_1d_input", "STFT", dict(n_fft=512, hop_length=128, output_format="Magnitude"), "x_1d")
    case("stft_3d_input", "STFT", dict(n_fft=512, hop_length=128, output_format="Magnitude"), "x_3d")
    case("stft_sines", "STFT", dict(n_fft=1024, hop_length=256, output_format="Magnitude"), "x_sines")

    # Mel / Gammatone
    case("mel_cfg3shape", "MelSpectrogram", dict(sr=22050, n_fft=1024, n_mels=128), "x_1s22k",
         attrs=("stride", "n_fft", "power"))
    case("mel_default", "MelSpectrogram", dict(), "x_1s22k")
    case("mel_htk_power1", "MelSpectrogram",
         dict(sr=16000, n_fft=512, n_mels=40, hop_length=160, power=1.0, htk=True, fmin=20.0,
              fmax=7600.0), "x_short")
    case("mel_norm_none_power3", "MelSpectrogram",
         dict(sr=22050, n_fft=512, n_mels=64, hop_length=128, power=3.0, norm=None), "x_short")
    case("mel_winlen_400", "MelSpectrogram",
         dict(sr=16000, n_fft=512, win_length=400, n_mels=80, hop_length=160), "x_short")
    case("mel_nocenter_const", "MelSpectrogram",
         dict(sr=22050, n_fft=512, n_mels=32, hop_length=256, center=False), "x_short")
    case("mel_sines", "MelSpectrogram", dict(sr=22050, n_fft=1024, n_mels=128), "x_sines")
    case("gamma_default", "Gammatonegram", dict(), "x_1s22k", attrs=("stride", "n_fft", "power"))
    case("gamma_small", "Gammatonegram",
         dict(sr=16000, n_fft=512, n_bins=32, hop_length=128, fmin=20.0, power=1.0), "x_short")

    # CQT1992v2
    cqt_attrs = ("kernel_width", "hop_length", "frequencies")
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("cqt1992v2_default_%s" % fmt.lower(), "CQT1992v2", dict(output_format=fmt),
             "x_1s22k", attrs=cqt_attrs)
    for nt in ("convolutional", "wrap"):
        case("cqt1992v2_norm_%s" % nt, "CQT1992v2", dict(output_format="Complex"), "x_1s22k",
             fwd=dict(normalization_type=nt))
    case("cqt1992v2_constpad", "CQT1992v2",
         dict(pad_mode="constant", fmin=110, n_bins=48, output_format="Complex"), "x_1s22k")
    case("cqt1992v2_nocenter", "CQT1992v2",
         dict(center=False, fmin=220, n_bins=36, hop_length=256, output_format="Magnitude"),
         "x_1s22k")
    case("cqt1992v2_bpo24_fs2", "CQT1992v2",
         dict(sr=22050, fmin=110, n_bins=60, bins_per_octave=24, filter_scale=0.5, hop_length=128,
              output_format="Complex"), "x_1s22k")
    case("cqt1992v2_hamming_norm2", "CQT1992v2",
         dict(fmin=220, n_bins=36, window="hamming", norm=2, output_format="Magnitude"),
         "x_1s22k")
    case("cqt1992v2_trainable", "CQT1992v2",
         dict(fmin=220, n_bins=24, trainable=True, output_format="Magnitude"), "x_short")
    case("cqt1992v2_cfg4_2s", "CQT1992v2",
         dict(sr=44100, hop_length=512, fmin=32.70, n_bins=84, bins_per_octave=12), "x_2s44k",
         attrs=cqt_attrs)
    case("cqt1992v2_sines", "CQT1992v2", dict(output_format="Magnitude"), "x_sines")

    # CQT2010v2
    c10_attrs = ("hop_length", "n_fft", "n_octaves", "downsample_factor", "earlydownsample",
                 "fmin_t", "frequencies", "n_bins")
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("cqt2010v2_default_%s" % fmt.lower(), "CQT2010v2", dict(output_format=fmt),
             "x_1s22k", attrs=c10_attrs)
    case("cqt2010v2_sr44k_early", "CQT2010v2", dict(sr=44100, output_format="Complex"), "x_2s44k",
         attrs=c10_attrs)
    case("cqt2010v2_cfg5_2s", "CQT2010v2",
         dict(sr=44100, hop_length=512, n_bins=96, output_format="Magnitude"), "x_2s44k",
         attrs=c10_attrs)
    case("cqt2010v2_noearly", "CQT2010v2",
         dict(sr=44100, earlydownsample=False, output_format="Magnitude"), "x_2s44k",
         attrs=c10_attrs)
    case("cqt2010v2_bins40", "CQT2010v2", dict(n_bins=40, fmin=110, output_format="Complex"),
         "x_1s22k", attrs=c10_attrs)
    case("cqt2010v2_constpad_wrap", "CQT2010v2",
         dict(pad_mode="constant", output_format="Complex"), "x_1s22k",
         fwd=dict(normalization_type="wrap"))
    case("cqt2010v2_convnorm", "CQT2010v2", dict(output_format="Magnitude"), "x_1s22k",
         fwd=dict(normalization_type="convolutional"))
    case("cqt2010v2_bpo24", "CQT2010v2",
         dict(fmin=65.4, n_bins=120, bins_per_octave=24, hop_length=256, output_format="Magnitude"),
         "x_1s22k", attrs=c10_attrs)
    case("cqt2010v2_trainable", "CQT2010v2",
         dict(fmin=220, n_bins=24, trainable=True, output_format="Magnitude"), "x_short")
    case("cqt2010v2_sines", "CQT2010v2", dict(output_format="Magnitude"), "x_sines")

    # VQT
    vqt_attrs = ("hop_length", "n_fft", "n_octaves", "downsample_factor", "earlydownsample",
                 "fmin_t", "frequencies", "n_bins", "n_filters")
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("vqt_gamma0_%s" % fmt.lower(), "VQT", dict(gamma=0, output_format=fmt), "x_1s22k",
             attrs=vqt_attrs)
    case("vqt_gamma10", "VQT", dict(gamma=10, output_format="Complex"), "x_1s22k", attrs=vqt_attrs)
    case("vqt_cfg5_gamma0_2s", "VQT", dict(sr=44100, hop_length=512, n_bins=96, gamma=0),
         "x_2s44k", attrs=vqt_attrs)
    case("vqt_cfg5_gamma10_2s", "VQT", dict(sr=44100, hop_length=512, n_bins=96, gamma=10),
         "x_2s44k", attrs=vqt_attrs)
    case("vqt_sr44k_early_bugcompat", "VQT", dict(sr=44100, gamma=0, output_format="Magnitude"),
         "x_2s44k", attrs=vqt_attrs)
    case("vqt_bins40_gamma5", "VQT", dict(n_bins=40, fmin=110, gamma=5, output_format="Magnitude"),
         "x_1s22k", attrs=vqt_attrs)
    case("vqt_wrap", "VQT", dict(gamma=3, output_format="Complex"), "x_1s22k",
         fwd=dict(normalization_type="wrap"))

    # buffers-only cases at the BASELINE sizes (forward too big to store)
    case("bufonly_stft_cfg2", "STFT", dict(n_fft=2048, hop_length=512, window="hann"), None,
         attrs=stft_attrs)
    # MFCC: mel -> power_to_db -> DCT-II
    case("mfcc_default", "MFCC", dict(), "x_1s22k", attrs=("n_mfcc", "top_db"))
    case("mfcc_13_of_40", "MFCC",
         dict(sr=16000, n_mfcc=13, n_fft=512, n_mels=40, hop_length=160), "x_short")
    case("mfcc_no_topdb_ref2", "MFCC",
         dict(sr=16000, n_mfcc=20, n_fft=512, n_mels=64, hop_length=128, top_db=None, ref=2.0,
              amin=1e-6), "x_short")
    case("mfcc_sines_topdb40", "MFCC",
         dict(sr=22050, n_mfcc=24, n_fft=1024, n_mels=80, hop_length=256, top_db=40.0), "x_sines")
    # inverse STFT: STFT.inverse and the iSTFT class
    case("istft_inverse_default", "STFT", dict(n_fft=512, hop_length=128, iSTFT=True), "X_512_128",
         method="inverse")
    case("istft_inverse_length", "STFT", dict(n_fft=512, hop_length=128, iSTFT=True), "X_512_128",
         fwd=dict(length=8000), method="inverse")
    case("istft_inverse_twosided", "STFT", dict(n_fft=512, hop_length=128, iSTFT=True),
         "X_512_128_full", fwd=dict(onesided=False, length=7000), method="inverse")
    case("istft_inverse_nocenter", "STFT", dict(n_fft=256, hop_length=64, center=False, iSTFT=True),
         "X_256_64_nocenter", method="inverse")
    case("istft_class_onesided", "iSTFT", dict(n_fft=512, hop_length=128), "X_512_128",
         fwd=dict(onesided=True))
    case("istft_class_twosided", "iSTFT", dict(n_fft=512, hop_length=128), "X_512_128_full")
    case("istft_class_hamming_len", "iSTFT", dict(n_fft=512, hop_length=100, window="hamming"),
         "X_512_100_hamming", fwd=dict(onesided=True, length=8000))
    # frequency-domain CQT variants
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("cqt1992_fd_%s" % fmt.lower(), "CQT1992",
             dict(sr=22050, fmin=220, n_bins=48, hop_length=256, output_format=fmt), "x_1s22k")
    case("cqt1992_fd_wrap_constpad", "CQT1992",
         dict(sr=16000, fmin=110, n_bins=36, hop_length=128, pad_mode="constant"), "x_short",
         fwd=dict(normalization_type="wrap"))
    for fmt in ("Magnitude", "Complex", "Phase"):
        case("cqt2010_fd_%s" % fmt.lower(), "CQT2010",
             dict(sr=22050, fmin=55, n_bins=60, output_format=fmt), "x_1s22k")
    case("cqt2010_fd_noearly_conv", "CQT2010",
         dict(sr=16000, fmin=110, n_bins=40, hop_length=256, earlydownsample=False), "x_short",
         fwd=dict(normalization_type="convolutional"))
    case("bufonly_cqt_testgrid", "CQT1992v2",
         dict(sr=44100, fmin=55, n_bins=207, bins_per_octave=24), None, attrs=cqt_attrs)
    case("bufonly_gamma_44k", "Gammatonegram", dict(sr=44100, n_fft=2048, n_bins=64), None)
    case("bufonly_mel_44k_htk", "MelSpectrogram", dict(sr=44100, n_fft=2048, n_mels=229, htk=True,
                                                       fmin=30.0, fmax=8000.0), None)

    fwd_store, buf_store, manifest = {}, {}, []
    for c in C:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import contextlib
            import io

            with contextlib.redirect_stdout(io.StringIO()):  # CQT1992 has no `verbose` switch
                kwv = {} if c["cls"] == "CQT1992" else {"verbose": False}
                mod = getattr(R, c["cls"])(**kwv, **c["ctor"])
        entry = dict(name=c["name"], cls=c["cls"], ctor=c["ctor"], fwd=c["fwd"], input=c["input"])
        if c["method"] != "forward":
            entry["method"] = c["method"]
        # attributes users read
        at = {}
        for a in c["attrs"]:
            v = getattr(mod, a)
            if isinstance(v, np.ndarray):
                v = [float(z) for z in v]
            elif isinstance(v, (list, tuple)):
                v = [float(z) for z in v]
            elif isinstance(v, (np.floating, np.integer)):
                v = v.item()
            at[a] = v
        entry["attrs"] = at
        # state_dict: names, shapes, dtypes, digests (+ data or strided sample)
        st = {}
        for k, v in mod.state_dict().items():
            a = v.detach().cpu().numpy()
            rec = dict(shape=list(a.shape), dtype=str(a.dtype), sha256=sha(a))
            key = "%s/%s" % (c["name"], k)
            if a.nbytes <= FULL_LIMIT:
                buf_store[key] = a
                rec["stored"] = "full"
            else:
                buf_store[key] = a.reshape(-1)[::SAMPLE_STRIDE].copy()
                rec["stored"] = "stride%d" % SAMPLE_STRIDE
            st[k] = rec
        entry["state"] = st
        entry["param_names"] = [k for k, _ in mod.named_parameters()]
        if c["input"] is not None:
            x = torch.from_numpy(inputs[c["input"]])
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                y = getattr(mod, c["method"])(x, **c["fwd"]) if c["method"] != "forward" else mod(x, **c["fwd"])
            y = y.detach().cpu().numpy()
            fwd_store[c["name"]] = y
            entry["out_shape"] = list(y.shape)
            entry["out_absmax"] = float(np.abs(y).max())
        manifest.append(entry)
        print("%-34s %-15s out=%s" % (c["name"], c["cls"], entry.get("out_shape")))

    np.savez_compressed(os.path.join(OUT, "forward.npz"), **fwd_store)
    np.savez_compressed(os.path.join(OUT, "buffers.npz"), **buf_store)
    with open(os.path.join(OUT, "cases.json"), "w") as f:
        json.dump(dict(reference="KinWaiCheuk/nnAudio v0.3.3 (Installation/nnAudio)",
                       torch=torch.__version__, numpy=np.__version__,
                       sample_stride=SAMPLE_STRIDE, cases=manifest), f, indent=1)
    tot = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(OUT) for fn in fns)
    print("wrote %d cases, %.2f MiB under %s" % (len(manifest), tot / 2 ** 20, OUT))


if __name__ == "__main__":
    main()
