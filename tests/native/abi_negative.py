#!/usr/bin/env python3
"""Negative-argument sweep over EVERY entry point of the C ABI, run by tests/test_capi_asan.py in a child interpreter with the
host-side AddressSanitizer build of the library (videonavqa_amd/lib/libvnqa_hip_asan.so, LD_PRELOAD = the ASan runtime).

No GPU is needed or touched: every call hands the entry point NULL pointers / zero sizes / a zeroed descriptor, which the
VNQA_CHECK_ARG prologue must reject with VNQA_ERR_INVALID_ARG (or VNQA_ERR_UNSUPPORTED for a geometry query) BEFORE any HIP
call — without dereferencing anything (ASan / a crash would end this process).  Prints one JSON line: per entry point the
return codes of its calls, and the list of offenders."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

OK, INVALID, HIP, UNSUPPORTED = 0, -1, -2, -3


def main():
    from videonavqa_amd import _lib as L
    lib = L.lib()            # VNQA_LIB points at the ASan build (set by the test)
    results, offenders = {}, []
    # entry points that are not "status = f(arguments)": version / error string, 0-or-1 geometry predicates, byte-count queries
    # (called all the same: they must survive zero arguments), and the stream constructors (bad arguments first, then the
    # device query fails without a GPU: VNQA_ERR_HIP is the correct answer there)
    for name in L.exported_symbols():
        res, args = L._SIGNATURES[name]
        fn = getattr(lib, name)

        def zero(a, null_structs):
            if a in (ctypes.c_void_p, ctypes.c_char_p):
                return None
            if isinstance(a, type) and issubclass(a, ctypes._Pointer):
                if null_structs:
                    return None
                return ctypes.pointer(a._type_())          # a zeroed descriptor
            if a in (ctypes.c_float, ctypes.c_double):
                return 0.0
            return 0
        codes = []
        has_struct = any(isinstance(a, type) and issubclass(a, ctypes._Pointer) for a in args)
        for null_structs in ((True, False) if has_struct else (True,)):
            # every call in its own forked child: a crash / ASan abort of ONE entry point is recorded and the sweep goes on
            r, w = os.pipe()
            pid = os.fork()
            if pid == 0:
                os.close(r)
                rc = fn(*[zero(a, null_structs) for a in args])
                msg = lib.vnqa_last_error() or b""
                os.write(w, json.dumps([rc if isinstance(rc, int) else 0, bool(msg)]).encode())
                os._exit(0)
            os.close(w)
            data = os.read(r, 4096)
            os.close(r)
            _, status = os.waitpid(pid, 0)
            if status != 0 or not data:
                codes.append("CRASH(status %d)" % status)
            else:
                rc, has_msg = json.loads(data.decode())
                codes.append(rc)
                if rc != 0 and not has_msg and res is ctypes.c_int and not name.endswith("_supported"):
                    codes.append("no error message")
        results[name] = codes
        if any(isinstance(c, str) and c.startswith("CRASH") for c in codes):
            offenders.append((name, codes, "crashed on zero arguments"))
            continue
        if name in ("vnqa_version", "vnqa_last_error"):
            continue
        if res is not ctypes.c_int or name.endswith("_blocks"):      # byte / block count queries: any value, no crash
            continue
        if name.endswith("_supported"):
            if any(c != 0 for c in codes):
                offenders.append((name, codes, "a zeroed descriptor must not be 'supported'"))
            continue
        ok = (INVALID, UNSUPPORTED, HIP) if name.startswith("vnqa_stream_create") else (INVALID, UNSUPPORTED)
        if any(c not in ok for c in codes):
            offenders.append((name, codes, "expected VNQA_ERR_INVALID_ARG"))
    print(json.dumps({"entry_points": len(results), "offenders": offenders, "codes": results}))
    return 1 if offenders else 0


if __name__ == "__main__":
    sys.exit(main())
