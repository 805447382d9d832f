"""ctypes binding of libbagel_hip.so.  The prototypes are parsed from include/bagel_hip.h, so the header is the
single source of truth for the C ABI.  There is NO fallback: if the library is missing the product fails loudly.
"""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
# BAGEL_HIP_LIB: an alternative build of the same sources (tools/ab_build.sh: A/B measurements of compile-time variants on one GPU box)
LIB_PATH = os.environ.get("BAGEL_HIP_LIB") or os.path.join(_HERE, "libbagel_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bagel_hip.h")


class BagelHipError(RuntimeError):
    pass


def parse_header(path=HEADER_PATH):
    """-> {name: (restype, [argtypes])} for every function declared in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    protos = {}
    for m in re.finditer(r"(const\s+char\s*\*|int)\s+(bagel_\w+)\s*\(([^;{]*)\)\s*;", src):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        restype = ctypes.c_char_p if "char" in ret else ctypes.c_int
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a or "bagel_stream_t" in a:
                    argtypes.append(ctypes.c_void_p)
                elif "int64_t" in a:
                    argtypes.append(ctypes.c_int64)
                elif "int32_t" in a or re.match(r"^(const\s+)?int\b", a):
                    argtypes.append(ctypes.c_int32)
                elif "float" in a:
                    argtypes.append(ctypes.c_float)
                else:
                    raise BagelHipError(f"cannot map C parameter '{a}' of {name}")
        protos[name] = (restype, argtypes)
    return protos


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise BagelHipError(
                f"{LIB_PATH} not found: build it with `python -m bagel_amd.build` (hipcc, gfx950). "
                "bagel_amd has no CPU or eager fallback by design.")
        # ONE HIP runtime per process: torch bundles its own libamdhip64 and the kernels have to launch into the context, the streams
        # and the allocations torch owns.  Loading this library first would bind it to the system ROCm's runtime instead (a second,
        # separate runtime: every launch then fails with "no ROCm-capable device is detected").  Importing torch first makes the
        # dynamic loader resolve libbagel_hip.so's libamdhip64 dependency to the copy that is already mapped.
        import torch  # noqa: F401
        L = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in parse_header().items():
            fn = getattr(L, name)   # AttributeError here == header/library drift
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = L
    return _lib


_DEBUG_SYNC = bool(os.environ.get("BAGEL_DEBUG_SYNC"))


def check(code, what=""):
    if code != 0:
        msg = lib().bagel_hip_last_error().decode()
        raise BagelHipError(f"{what} failed ({code}): {msg}")
    if _DEBUG_SYNC:   # debugging aid: localise an asynchronous kernel fault to the launch that caused it
        import sys
        import torch
        sys.stderr.write(f"[bagel_amd] {what} ...")
        sys.stderr.flush()
        torch.cuda.synchronize()
        sys.stderr.write(" ok\n")
        sys.stderr.flush()
