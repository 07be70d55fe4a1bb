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
UNITS = [(SRC, True), (os.path.join(HERE, "csrc", "octave_stream.hip"), True)]


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


def _compile(src, obj, ablate, verbose):
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-inline-asm", "-c",
           "-Rpass-analysis=kernel-resource-usage", "-I", INC, src, "-o", obj]
    if ablate:
        cmd.insert(1, "-DMISPEC_ABLATE")
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    remarks, other = resource_usage(res.stderr)
    if other.strip():
        sys.stderr.write(other)
    if res.returncode != 0:
        raise subprocess.CalledProcessError(res.returncode, cmd)
    if any(v > 0 and k.startswith("octave_stream") for k, v in remarks.items()):
        traffic = scratch_instructions(obj)
        for k in list(remarks):
            if k.startswith("octave_stream") and remarks[k] > 0 and traffic.get(k, 1) == 0:
                remarks[k] = 0  # (a stack slot nobody loads or stores: SGPR spills in VGPR lanes)
    return remarks


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
            tag = "_ablate" if (ablate and takes_ablate) else ""
            obj = os.path.join(objdir, os.path.basename(src).replace(".hip", tag + ".o"))
            objs.append(obj)
            # a unit depends on its own source, the headers and (mispec.hip) the .inl files it includes
            mine = [d for d in deps if not d.endswith(".hip") or d == src]
            if src != SRC:
                mine = [d for d in mine if not d.endswith(".inl") and not d.endswith("fft_core.h")]
            if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(d) for d in mine):
                jobs.append(ex.submit(_compile, src, obj, ablate and takes_ablate, verbose))
        remarks = {}
        for j in jobs:
            remarks.update(j.result())
    # The MFMA kernels feed LDS with global_load_lds and pace it with s_waitcnt vmcnt: a build of
    # the bf16x3 kernel that spilled (scratch loads inside its K loop, which count on the same
    # vmcnt) produced run-to-run different results on the MI355X.  Refuse such a build.
    spilled = {k: v for k, v in remarks.items() if k.startswith(("framed_", "octave_stream")) and v > 0}
    if spilled and ablate and not any(k.startswith("framed_") for k in spilled):
        # (the benchmarking build's extra switches cost the streaming octave kernel a few registers in the
        # instances the phase-clock scripts do not run: product builds never get here)
        sys.stderr.write("warning (benchmarking build only): scratch in %s\n" % spilled)
    elif spilled:
        raise RuntimeError("kernels using LDS-direct loads must not use scratch: %s" % spilled)
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
