"""List the s_waitcnt vmcnt(...) instructions hipcc placed INSIDE loops of the kernels that pace LDS-direct loads by hand
(a compiler-placed vmcnt(0) in such a loop makes a wave wait for its own freshly issued stores), and optionally dump one
kernel's ISA.   python scripts/isa_waits.py [object] [--dump substring]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def disassemble(obj):
    tmp = tempfile.mkdtemp()
    dst = os.path.join(tmp, os.path.basename(obj))
    subprocess.run(["cp", obj, dst], check=True)
    subprocess.run([OBJDUMP, "--offloading", dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, cwd=tmp)
    f = glob.glob(dst + ".*gfx950*")[0]
    return subprocess.run([OBJDUMP, "-d", f], stdout=subprocess.PIPE, text=True, check=True).stdout.splitlines()


def kernels(lines):
    heads = [(i, m.group(2)) for i, l in enumerate(lines) for m in [re.match(r"^([0-9a-f]+) <(\S+)>:", l)] if m]
    heads.append((len(lines), None))
    for (a, name), (b, _) in zip(heads, heads[1:]):
        yield name, lines[a + 1:b]


def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()


ADDR = re.compile(r"//\s*([0-9A-Fa-f]{12}):")


def loops_of(body):
    addrs = {int(m.group(1), 16): j for j, l in enumerate(body) for m in [ADDR.search(l)] if m}
    out = []
    for j, l in enumerate(body):
        m = re.search(r"s_cbranch_\w+\s+(\d+)|s_branch\s+(\d+)", l)
        if m:
            off = int(m.group(1) or m.group(2))
            if off >= 32768:
                tgt = int(ADDR.search(l).group(1), 16) + 4 + (off - 65536) * 4
                if tgt in addrs:
                    out.append((addrs[tgt], j))
    return out


def main():
    argv = sys.argv[1:]
    dump = None
    if "--dump" in argv:
        i = argv.index("--dump")
        dump = argv[i + 1]
        del argv[i:i + 2]
    obj = argv[0] if argv else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "nnaudio_amd", "csrc", "_obj", "mispec.o")
    for name, body in kernels(disassemble(obj)):
        dn = demangle(name)
        if dump:
            if dump in dn:
                print("\n".join(body))
            continue
        if not any("global_load_lds" in l for l in body):
            continue
        inloop = sorted({j for s, e in loops_of(body) for j in range(s, e + 1) if "s_waitcnt" in body[j] and "vmcnt" in body[j]})
        cnt = {}
        for j in inloop:
            k = int(re.search(r"vmcnt\((\d+)\)", body[j]).group(1))
            cnt[k] = cnt.get(k, 0) + 1
        print("%-100s %s" % (dn[:100], dict(sorted(cnt.items()))))


if __name__ == "__main__":
    main()
