"""Build ``csrc/libmispec.so`` in-tree with hipcc for gfx950 (``python -m nnaudio_amd.build``)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "mispec.hip")
OUT = os.path.join(HERE, "csrc", "libmispec.so")
# benchmarking build: the same source with -DMISPEC_ABLATE (ablation / A-B bits behind
# mispec_framed_gemm_args.reserved); only scripts/kbench.py and scripts/profile.sh load it
OUT_ABLATE = os.path.join(HERE, "csrc", "libmispec_ablate.so")
INC = os.path.join(ROOT, "include")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC)")


# translation units of the library: (source, takes -DMISPEC_ABLATE in the benchmarking build)
UNITS = [(SRC, True), (os.path.join(HERE, "csrc", "octave_stream.hip"), True), (os.path.join(HERE, "csrc", "cqt_chain.hip"), False)]


def scratch_instructions(obj):
    """{kernel: number of scratch_* instructions} of the gfx950 code in a compiled object.  hipcc reports a
    non-zero ScratchSize also for a kernel whose only stack object is the bookkeeping slot of SGPR spills
    (they live in VGPR lanes: no memory instruction); what the LDS-direct kernels must not have is scratch
    TRAFFIC, whose loads and stores count on the vmcnt they pace their loads with."""
    import glob
    import re

    llvm = os.path.join(os.path.dirname(os.path.realpath(hipcc())), "..", "lib", "llvm", "bin")
    objdump = os.path.join(llvm, "llvm-objdump")
    if not os.path.exists(objdump):
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    for f in glob.glob(obj + ".*.hipv4-*") + glob.glob(obj + ".*.host-*"):
        os.remove(f)
    subprocess.run([objdump, "--offloading", obj], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
    counts = {}
    for f in glob.glob(obj + ".*.hipv4-*gfx950*"):
        dis = subprocess.run([objdump, "-d", f], stdout=subprocess.PIPE, text=True, check=True).stdout
        name = None
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", m.group(1))
                counts.setdefault(name, 0)
            elif name and re.search(r"\bscratch_(load|store)", line):
                counts[name] += 1
    for f in glob.glob(obj + ".*.hipv4-*") + glob.glob(obj + ".*.host-*"):
        os.remove(f)
    return counts


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-Wno-int-to-pointer-cast"]


def _sidecar(obj):
    return obj + ".remarks.json"


def _flags_key(ablate):
    return " ".join(FLAGS + (["-DMISPEC_ABLATE"] if ablate else []))


def _cached_remarks(obj, ablate):
    """The scratch remarks recorded when `obj` was compiled, or None when the object has no record or was compiled with
    other flags / defines (then it is stale whatever its date)."""
    import json

    try:
        with open(_sidecar(obj)) as f:
            rec = json.load(f)
    except (OSError, ValueError):
        return None
    if rec.get("flags") != _flags_key(ablate) or not os.path.exists(obj):
        return None
    return rec.get("scratch", {})


def _discard(obj):
    for f in (obj, _sidecar(obj)):
        if os.path.exists(f):
            os.remove(f)


def _compile(src, obj, ablate, verbose):
    """src -> obj, with the kernels' scratch sizes recorded NEXT TO the object (obj.remarks.json): the guard in build()
    must see them on every later call that reuses the object, not only in the call that compiled it (a
    refused build left the spilled object on disk, newer than its sources, and the next call linked it silently).
    The object appears under its final name only together with its record."""
    import json

    tmp = obj + ".tmp"
    cmd = [hipcc(), *FLAGS, "-c", "-Rpass-analysis=kernel-resource-usage", "-I", INC, src, "-o", tmp]
    if ablate:
        cmd.insert(1, "-DMISPEC_ABLATE")
    if verbose:
        print(" ".join(cmd[:-1] + [obj]), flush=True)
    _discard(obj)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    remarks, other = resource_usage(res.stderr)
    if other.strip():
        sys.stderr.write(other)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise subprocess.CalledProcessError(res.returncode, cmd)
    if any(v > 0 and k.startswith("octave_stream") for k, v in remarks.items()):
        traffic = scratch_instructions(tmp)
        for k in list(remarks):
            if k.startswith("octave_stream") and remarks[k] > 0 and traffic.get(k, 1) == 0:
                remarks[k] = 0  # (a stack slot nobody loads or stores: SGPR spills in VGPR lanes)
    with open(_sidecar(obj) + ".tmp", "w") as f:
        json.dump({"flags": _flags_key(ablate), "scratch": remarks}, f)
    os.replace(tmp, obj)
    os.replace(_sidecar(obj) + ".tmp", _sidecar(obj))
    return remarks


# Kernels that pace LDS-direct loads with hand-stated s_waitcnt vmcnt(n).
#   framed_* / octave_stream*: scratch traffic inside the K loop sits BETWEEN loads whose count the wait relies on -- a
#     spilling build of the bf16x3 kernel gave run-to-run different results on the MI355X.  Refused outright.
#   stft_fft_* / istft_*: their waits are "at most n operations outstanding, n = the stores issued after the loads": extra
#     scratch operations behind the loads only make that wait stronger (in-order retirement, probed:
#     experiments/ldsdma_hazard), so scratch there is slow, not wrong -- but it must not appear unnoticed: refused unless
#     the instance is on this list with the bytes per lane it is known to have (review them when they change).
SCRATCH_ALLOWED = {
    "stft_fft_kernelILi512ELi2ELb0E": 32,     # n_fft = 1024 Power, two workgroups per CU (128 VGPRs)
    "stft_fft_kernelILi512ELi3ELb0E": 32,     # ... atan2 phase
    "stft_fft_kernelILi512ELi1ELb0E": 32,     # ... Magnitude
    "istft_ola_fft_kernelILi1024EE": 32,       # the fused inverse at n_fft = 2048 (256 VGPRs + a few spilled values; round 6:
                                               # its 32 registers of pre-processing factors in LDS instead = 238 VGPRs and no
                                               # scratch, but 8 KB more LDS where 2 KB are left of the 160: the fused launch is
                                               # then refused and the two-launch inverse takes 0.37 instead of 0.25 ms)
}


def refused_scratch(remarks, ablate):
    """{kernel: bytes} of the kernels in `remarks` whose scratch use the build must refuse."""
    bad = {}
    for k, v in remarks.items():
        if v <= 0:
            continue
        if k.startswith(("framed_", "octave_stream")):
            if ablate and not k.startswith("framed_"):
                continue  # (the benchmarking build's extra switches cost the streaming octave kernel a few registers)
            bad[k] = v
        elif k.startswith(("stft_fft_", "istft_")):
            limit = max([b for pat, b in SCRATCH_ALLOWED.items() if k.startswith(pat)], default=0)
            if v > limit and not ablate:
                bad[k] = v
    return bad


def build(force=False, verbose=True, ablate=False):
    """Compile the product library (or, with ``ablate``, the benchmarking build): every translation unit
    to an object (in parallel; an object is reused while it is newer than the sources), then one link."""
    from concurrent.futures import ThreadPoolExecutor

    out = OUT_ABLATE if ablate else OUT
    csrc = os.path.dirname(SRC)
    deps = [os.path.join(INC, "mispec.h")] + [
        os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".inl", ".h"))
    ]
    newest = max(os.path.getmtime(d) for d in deps)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest:
        return out
    objdir = os.path.join(csrc, "_obj")
    os.makedirs(objdir, exist_ok=True)
    jobs, objs = [], []
    with ThreadPoolExecutor(len(UNITS)) as ex:
        for src, takes_ablate in UNITS:
            tag = "_ablate" if ablate else ""  # (own objects for every unit: the two builds may run side by side)
            obj = os.path.join(objdir, os.path.basename(src).replace(".hip", tag + ".o"))
            objs.append(obj)
            # a unit depends on its own source, the headers and (mispec.hip) the .inl files it includes
            mine = [d for d in deps if not d.endswith(".hip") or d == src]
            if src != SRC:
                mine = [d for d in mine if not d.endswith(".inl") and not d.endswith("fft_core.h")]
            cached = None if force else _cached_remarks(obj, ablate and takes_ablate)
            if cached is None or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in mine):
                jobs.append((obj, ex.submit(_compile, src, obj, ablate and takes_ablate, verbose)))
            else:
                jobs.append((obj, cached))
        remarks, per_obj = {}, {}
        for obj, j in jobs:
            per_obj[obj] = j if isinstance(j, dict) else j.result()
            remarks.update(per_obj[obj])
    # The MFMA kernels feed LDS with global_load_lds and pace it with s_waitcnt vmcnt: a build of
    # the bf16x3 kernel that spilled (scratch loads inside its K loop, which count on the same
    # vmcnt) produced run-to-run different results on the MI355X.  Refuse such a build -- and take the
    # offending objects off the disk, so that the refusal cannot be bypassed by calling build() again.
    spilled = refused_scratch(remarks, ablate)
    if ablate and any(v > 0 and k.startswith("octave_stream") for k, v in remarks.items()):
        sys.stderr.write("warning (benchmarking build only): scratch in %s\n"
                         % {k: v for k, v in remarks.items() if v > 0 and k.startswith("octave_stream")})
    if spilled:
        for obj, r in per_obj.items():
            if any(k in spilled for k in r):
                _discard(obj)
        raise RuntimeError("kernels using LDS-direct loads must not use scratch (beyond build.SCRATCH_ALLOWED): %s" % spilled)
    slow = {k: v for k, v in remarks.items() if v > 0 and k.startswith(("fold", "split_", "clip_", "octave_", "stft_fft"))}
    if slow and verbose:  # (a pre-pass with a stack array runs ~25 % slower: framed_fold2.inl)
        sys.stderr.write("warning: scratch in %s\n" % slow)
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", out + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + ".tmp", out)
    return out


def resource_usage(stderr):
    """Split hipcc's stderr into {kernel: scratch bytes per lane} (from the
    -Rpass-analysis=kernel-resource-usage remarks) and everything else."""
    import re

    usage, other, name = {}, [], None
    for line in stderr.splitlines(True):
        if "[-Rpass-analysis=kernel-resource-usage]" not in line:
            # source echo lines that follow a remark ("  797 | __global__ ..." / "      | ^")
            if name is not None and (re.match(r"\s+(\d+\s+)?\|", line)
                                     or line.startswith("In file included from")):
                continue
            other.append(line)
            continue
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and name:
            short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
            usage[short] = int(m.group(1))
    return usage, "".join(other)


def build_all(force=False, verbose=True):
    """Product library and benchmarking build side by side (two hipcc processes)."""
    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(2) as ex:
        jobs = [ex.submit(build, force=force, verbose=verbose, ablate=a) for a in (False, True)]
        return [j.result() for j in jobs]


if __name__ == "__main__":
    force = "--force" in sys.argv
    if "--ablate" in sys.argv:
        print(build(force=force, ablate=True))
    elif "--product" in sys.argv:
        print(build(force=force))
    else:
        print(*build_all(force=force), sep="\n")
