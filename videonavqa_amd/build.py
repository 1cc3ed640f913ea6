"""Build recipe for libvnqa_hip.so: hipcc, gfx950 only, in-tree output (videonavqa_amd/lib/).

The library is a plain C-ABI shared object (include/vnqa_hip.h); it links only the HIP runtime.
Cross-compiles without a GPU.  Usage: python -m videonavqa_amd.build [--force]
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libvnqa_hip.so")
# The same sources with -DVNQA_H16_IS_F16: the library's 16-bit storage format is IEEE fp16 instead of bf16
# (csrc/vnqa_common.h); selected by precision='fp16' / VNQA_HALF=f16 on the Python side.
LIB_F16 = os.path.join(LIBDIR, "libvnqa_hip_f16.so")
# Host-side AddressSanitizer build (sanitizer row): the same sources with -fsanitize=address on the HOST code only
# (-fno-gpu-sanitize: device code is compiled as usual — GPU ASan / xnack+ code objects are not available on this pool);
# tests/test_capi_asan.py drives every entry point's argument checks through it on the CPU box.  Never loaded by the product.
LIB_ASAN = os.path.join(LIBDIR, "libvnqa_hip_asan.so")
VARIANTS = {"bf16": (LIB, [], "libvnqa_hip"), "f16": (LIB_F16, ["-DVNQA_H16_IS_F16"], "libvnqa_hip_f16"),
            "asan": (LIB_ASAN, ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-g", "-fno-omit-frame-pointer"],
                     "libvnqa_hip_asan")}


def asan_runtime():
    """Path of the shared ASan runtime the 'asan' variant links against (LD_PRELOAD it into the test interpreter)."""
    out = subprocess.check_output([os.path.join(os.path.dirname(HIPCC), "..", "lib", "llvm", "bin", "clang"),
                                   "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    if not os.path.isabs(out):
        import glob
        cand = glob.glob("/opt/rocm*/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
        out = cand[0] if cand else out
    return out
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["-DVNQA_DIAG_SKIP_DMA"] if os.environ.get("VNQA_DIAG") else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast",
         "-Wno-unused-result", "-I", os.path.join(HERE, "..", "include")]


# per-source extra flags.  conv_wreg.hip / conv_ps.hip: their main loops are ONE fully unrolled instruction stream (hand-placed MFMA / LDS read /
# DMA / epilogue interleave); before constant folding the body exceeds LLVM's default pragma-unroll budget, and a loop
# left rolled would index register arrays at run time (scratch): make that a build error, never a slow kernel.
_UNROLL_ALL = ["-mllvm", "-pragma-unroll-threshold=4000000", "-Werror=pass-failed"]
PER_FILE_FLAGS = {"conv_wreg.hip": _UNROLL_ALL, "conv_ps.hip": _UNROLL_ALL}


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(extra=()):
    h = hashlib.sha256()
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    for f in sources() + headers + [os.path.join(HERE, "..", "include", "vnqa_hip.h")]:
        with open(f, "rb") as fh:
            h.update(fh.read())
    # (flags hashed with the checkout path stripped: the same tree at another location — a gpurun box — must not rebuild)
    h.update(repr(sorted(PER_FILE_FLAGS.items())).encode())
    h.update(" ".join(FLAGS + list(extra)).replace(os.path.join(HERE, ".."), "<root>").replace(HERE, "<pkg>").encode())
    return h.hexdigest()


def build(force=False, verbose=True, variant="bf16"):
    """Compile every source and link the library (variant 'bf16': libvnqa_hip.so, 'f16': libvnqa_hip_f16.so, 'all': both)
    unless the source digest matches the stamp.  Safe to call from several
    processes at once (torchrun ranks): an exclusive file lock serialises them, the winners of later turns find the stamp
    up to date, and objects / the library are written to temporary names and renamed into place."""
    import fcntl
    import tempfile
    if variant == "all":
        return [build(force, verbose, v) for v in ("bf16", "f16")]
    LIB, extra, stem_name = VARIANTS[variant]
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, stem_name + ".stamp")
    dig = _digest(extra)

    def fresh():
        return os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig

    if not force and fresh():
        return LIB
    with open(os.path.join(LIBDIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and fresh():          # another process built it while this one waited
                return LIB
            tmpdir = tempfile.mkdtemp(prefix=".build.", dir=LIBDIR)
            try:
                objs, procs = [], []
                for src in sources():
                    obj = os.path.join(tmpdir, os.path.basename(src) + ".o")
                    cmd = [HIPCC] + FLAGS + extra + PER_FILE_FLAGS.get(os.path.basename(src), []) + \
                        (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", src, "-o", obj]
                    procs.append((src, subprocess.Popen(cmd)))
                    objs.append(obj)
                failed = [src for src, p in procs if p.wait() != 0]
                if failed:
                    raise RuntimeError("hipcc failed on %s" % ", ".join(failed))
                tmp_lib = os.path.join(tmpdir, stem_name + ".so")
                link_extra = ["-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan"] if variant == "asan" else []
                subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + link_extra + objs)
                if variant == "bf16":
                    for obj in objs:           # keep the objects next to the library (inspection: llvm-objdump)
                        os.replace(obj, os.path.join(LIBDIR, os.path.basename(obj)))
                os.replace(tmp_lib, LIB)
                with open(stamp + ".tmp", "w") as fh:
                    fh.write(dig)
                os.replace(stamp + ".tmp", stamp)
            finally:
                import shutil
                shutil.rmtree(tmpdir, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, variant="all")
