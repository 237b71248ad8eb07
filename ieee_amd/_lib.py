"""ctypes binding of the C ABI declared in include/ieee_amd.h.

This is exactly the binding a maintainer of the reference would add to call the
library from Python.  The library is built in-tree by
`__graft_entry__.build()` / `make -C ieee_amd/csrc`; it is never pip-installed.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IEEE_AMD_LIB", os.path.join(_HERE, "libieee_amd.so"))   # override: kernel A/B experiments
_lib = None

c_void_p, c_int, c_int64, c_float, c_double = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_float, ctypes.c_double)

IEEE_F32, IEEE_BF16 = 0, 1

HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "ieee_amd.h")

_CTYPE = {"int": c_int, "int64_t": c_int64, "float": c_float, "double": c_double}


def _parse_header(path=HEADER_PATH):
    """include/ieee_amd.h is the single source of truth: every prototype there becomes a ctypes
    signature here (pointers -> void*, scalars by name), so the binding cannot drift from the ABI."""
    import re
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"^\s*#.*$", "", txt, flags=re.M)
    sigs, res = {}, {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(ieee_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argt.append(c_void_p)
                else:
                    base = a.replace("const", "").split()[0]
                    argt.append(_CTYPE[base])
        sigs[name] = argt
        if "char" in ret and "*" in ret:
            res[name] = ctypes.c_char_p
        elif "*" in ret:
            res[name] = c_void_p
        else:
            res[name] = _CTYPE[ret.replace("const", "").split()[0]]
    return sigs, res


_SIGNATURES, _RESTYPE = _parse_header()


class WgradReduceDesc(ctypes.Structure):
    """struct ieee_wgrad_reduce_desc of include/ieee_amd.h (ieee_conv2d_wgrad_deferred / ieee_wgrad_reduce_batch)."""
    _fields_ = [("slab", c_void_p), ("dw", c_void_p), ("slab_gs", c_int64), ("dw_gs", c_int64),
                ("nsplit", ctypes.c_int32), ("Co", ctypes.c_int32), ("Ci", ctypes.c_int32), ("RS", ctypes.c_int32),
                ("kind", ctypes.c_int32), ("sl_log2", ctypes.c_int32), ("blocks", ctypes.c_int32),
                ("accumulate", ctypes.c_int32), ("block_begin", ctypes.c_int32), ("reserved_", ctypes.c_int32)]


class ConvExtras(ctypes.Structure):
    """struct ieee_conv_extras of include/ieee_amd.h (ieee_conv2d_fwd_ex / ieee_conv2d_dgrad_ex): the options of ONE conv call"""
    _fields_ = [("totals", c_void_p), ("group_stride", c_int64), ("replicas", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("overflow", c_void_p), ("start", c_void_p), ("stop", c_void_p)]


class SgemmSet(ctypes.Structure):
    """struct ieee_sgemm_set of include/ieee_amd.h (ieee_sgemm_grouped_pair_ws): one uniform problem set of a pair"""
    _fields_ = [("groups", c_int64), ("A", c_void_p), ("B", c_void_p), ("C", c_void_p), ("bias", c_void_p),
                ("M", c_int64), ("N", c_int64), ("K", c_int64), ("sam", c_int64), ("sak", c_int64), ("sbn", c_int64),
                ("sbk", c_int64), ("ldc", c_int64), ("alpha", c_float), ("relu", ctypes.c_int32), ("accumulate", ctypes.c_int32)]


class IeeeAmdError(RuntimeError):
    pass


def exported_symbols():
    return sorted(_SIGNATURES)


def load():
    """Loads libieee_amd.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise IeeeAmdError(
            "ieee_amd HIP library not built: %s is missing. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C ieee_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch first: its wheel carries its own copy of the HIP runtime (libamdhip64.so under torch/lib).  Loaded before torch, this
    # library binds the system copy instead, the process then holds TWO runtimes, and the one this library talks to does not
    # see the device torch initialised (ieee_device_is_gfx950() = 0 -- met by `python __graft_entry__.py smoke`, whose build()
    # loaded the library before smoke() imported torch).  Importing torch does not start the runtime.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().ieee_last_error()
        raise IeeeAmdError((msg or b"").decode("utf-8", "replace") or "ieee_amd error %d" % status)


def require_gpu():
    """The product path needs a gfx950 GPU; fail loudly otherwise."""
    import torch
    if not torch.cuda.is_available():
        raise IeeeAmdError("ieee_amd needs an MI355X (gfx950) GPU: torch.cuda.is_available() is False and there "
                           "is no CPU fallback.")
    lib = load()
    if not lib.ieee_device_is_gfx950():
        raise IeeeAmdError("ieee_amd kernels are built for gfx950 only; current device is not gfx950.")
    return lib


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
