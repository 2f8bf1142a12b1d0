"""Per-kernel register / LDS / spill / code-size table of a built libmrmt3_hip*.so (gfx950 code object), read from the
AMDGPU metadata note and the symbol table — no GPU needed.

    python profiles/tools/kernel_resources.py LIB [LIB2]  [--match REGEX]

With two libraries: the kernels whose figures differ, side by side (round 5: the product build against the round-4
build and against the -DMRMT3_DIAG build, profiles/r05_diag_out_of_product.txt).
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    """The gfx950 code objects of `lib` (one per translation unit: the .hip_fatbin section of the host library is a
    sequence of offload bundles), unbundled into temp files; returns their paths."""
    tmp = tempfile.mkdtemp(prefix="kres_")
    fat = os.path.join(tmp, "fatbin")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    blob = open(fat, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)] + [len(blob)]
    outs = []
    for i in range(len(starts) - 1):
        part = os.path.join(tmp, "bundle%d" % i)
        with open(part, "wb") as fh:
            fh.write(blob[starts[i]:starts[i + 1]])
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + part],
                           capture_output=True, text=True)
        for t in r.stdout.split():
            if "gfx950" in t:
                out = os.path.join(tmp, "co%d" % i)
                subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                                "--targets=" + t, "--output=" + out], check=True)
                outs.append(out)
    return outs


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def table(lib):
    out = {}
    for co in code_objects(lib):
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co], capture_output=True, text=True).stdout
        size = {}
        for l in syms.splitlines():
            f = l.split()
            if len(f) >= 8 and f[3] == "FUNC":
                size[f[7]] = int(f[2])
        for blk in notes.split("  - .agpr_count:")[1:]:
            def g(key, d=0):
                m = re.search(r"\.%s:\s+(\S+)" % key, blk)
                return m.group(1) if m else d
            name = g("name")
            out[name] = dict(vgpr=int(g("vgpr_count")), sgpr=int(g("sgpr_count")), agpr=int(blk.split()[0]),
                             vspill=int(g("vgpr_spill_count")), sspill=int(g("sgpr_spill_count")),
                             lds=int(g("group_segment_fixed_size")), scratch=int(g("private_segment_fixed_size")),
                             code=size.get(name, 0))
    dm = demangle(list(out))
    return {dm[k]: v for k, v in out.items()}


def fmt(v):
    return "%4d VGPR %3d SGPR  spill %d/%d  LDS %6d  scratch %3d  code %6d B" % (
        v["vgpr"], v["sgpr"], v["vspill"], v["sspill"], v["lds"], v["scratch"], v["code"])


def short(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*\)$", "", name)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--match=")), ".")
    a = table(args[0])
    if len(args) == 1:
        for k in sorted(a):
            if re.search(match, k):
                print("%-70s %s" % (short(k), fmt(a[k])))
        print("total code bytes:", sum(v["code"] for v in a.values()))
        return
    b = table(args[1])
    print("A =", args[0], "\nB =", args[1])
    for k in sorted(set(a) | set(b)):
        if not re.search(match, k):
            continue
        if k not in a or k not in b:
            print("%-70s only in %s" % (short(k), "A" if k in a else "B"))
        elif a[k] != b[k]:
            print("%-70s\n    A %s\n    B %s" % (short(k), fmt(a[k]), fmt(b[k])))
    print("total code bytes: A %d  B %d" % (sum(v["code"] for v in a.values()), sum(v["code"] for v in b.values())))


if __name__ == "__main__":
    main()
