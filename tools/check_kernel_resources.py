#!/usr/bin/env python3
"""Compile every HIP source of the library to gfx950 assembly (no GPU needed) and list per kernel: VGPRs, AGPRs, SGPR / VGPR spills,
scratch bytes, LDS bytes — and flag anything that spills to scratch.  `--diff <git-rev>` prints the kernels whose register
counts changed against that revision (an epilogue edit that raises a small tile's VGPR high-water mark over an occupancy step —
128 / 170 / 256 at 8 / 6 / 4 waves per SIMD — slows kernels that never run the edited path: round 3).
    python tools/check_kernel_resources.py [--diff REV] [file.hip ...]"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from videonavqa_amd import build as B
HIPCC = B.HIPCC


def kernels_of(path, inc):
    out = tempfile.mktemp(suffix=".s")
    # the library's own flags (videonavqa_amd/build.py), per-file ones included: the hand-scheduled kernels depend on them
    flags = [f for f in B.FLAGS if f != "-fPIC"] + B.PER_FILE_FLAGS.get(os.path.basename(path), [])
    cmd = [HIPCC] + flags + ["--cuda-device-only", "-S", "-I", inc, "-I", os.path.join(inc, "..", "..", "include"), "-x", "hip", path, "-o", out]
    if subprocess.run(cmd, stderr=subprocess.DEVNULL).returncode != 0:
        return None
    s = open(out).read()
    os.unlink(out)
    res = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
        body = m.group(2)
        g = lambda k: int(re.search(k + r":\s+(\d+)", body).group(1)) if re.search(k + r":\s+(\d+)", body) else 0
        res[m.group(1)] = dict(vgpr=g(r"\.vgpr_count"), agpr=g(r"\.agpr_count"), sspill=g(r"\.sgpr_spill_count"),
                               vspill=g(r"\.vgpr_spill_count"), scratch=g(r"\.private_segment_fixed_size"),
                               lds=g(r"\.group_segment_fixed_size"))
    return res


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    args = sys.argv[1:]
    rev = None
    if args[:1] == ["--diff"]:
        rev, args = args[1], args[2:]
    src = os.path.join(ROOT, "videonavqa_amd", "csrc")
    files = [os.path.abspath(a) for a in args] or sorted(glob.glob(os.path.join(src, "*.hip")))
    old_dir = None
    if rev:
        old_dir = tempfile.mkdtemp()
        subprocess.check_call("git -C %s archive %s videonavqa_amd/csrc include | tar -x -C %s" % (ROOT, rev, old_dir), shell=True)
    bad = 0
    for f in files:
        new = kernels_of(f, src)
        if new is None:
            print("%s: does not compile standalone" % os.path.basename(f))
            continue
        old = None
        if old_dir:
            of = os.path.join(old_dir, "videonavqa_amd", "csrc", os.path.basename(f))
            old = kernels_of(of, os.path.dirname(of)) if os.path.exists(of) else {}
        dm = demangle(list(new))
        for k, r in sorted(new.items()):
            spill = r["scratch"] > 0 or r["vspill"] > 0
            changed = old is not None and (k not in (old or {}) or any(old[k][x] != r[x] for x in ("vgpr", "agpr", "scratch", "vspill")))
            if spill or (old is not None and changed) or (old is None):
                tag = "SPILL " if spill else ""
                was = "" if not (old and k in old) else "   (was vgpr %d agpr %d scratch %d)" % (old[k]["vgpr"], old[k]["agpr"], old[k]["scratch"])
                print("%s%-22s vgpr %3d agpr %3d sspill %2d vspill %2d scratch %4d lds %6d  %s%s"
                      % (tag, os.path.basename(f), r["vgpr"], r["agpr"], r["sspill"], r["vspill"], r["scratch"], r["lds"],
                         dm[k].replace("(anonymous namespace)::", "")[:100], was))
            bad += spill
    print("%d kernel(s) spill to scratch" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
