"""ctypes loader for liblamp_hip.so.

The signatures are derived from include/lamp_hip.h itself, so the Python binding can never drift
from the C ABI: every `int lamp_*(...)` declaration in the header becomes a checked callable.
This is the same mechanical mapping a JNI adapter would use.

There is deliberately NO fallback: if the HIP library is missing or a call fails, an exception
is raised (the product path must fail loudly - it never routes through oracle/ or a CPU path).
"""
from __future__ import annotations

import ctypes as C
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HEADERS = [os.path.join(ROOT, "include", "lamp_hip.h"), os.path.join(ROOT, "include", "lamp_host.h")]
# LAMP_LIB_PATH: a diagnostic build of the same library (in-kernel stamps: `make EXTRA=-DGEMM_STAMP BUILD=build_stamp LIBDIR=../lib_stamp`)
LIB_PATH = os.environ.get("LAMP_LIB_PATH") or os.path.join(_HERE, "lib", "liblamp_hip.so")


class LampError(RuntimeError):
    pass


_PTR = C.c_void_p
_TYPE_MAP = [
    # (regex on the normalised parameter type, ctypes type)
    (r"^const char\*$", C.c_char_p),
    (r"^char\*$", C.c_char_p),
    (r"^(const )?(lamp_\w+)\*\*$", C.POINTER(_PTR)),          # out handle
    (r"^(const )?(lamp_\w+)\* ?const\*$", C.POINTER(_PTR)),   # array of handles
    (r"^(const )?(lamp_\w+)\*\[\d*\]$", C.POINTER(_PTR)),      # out3[3]
    (r"^(const )?(lamp_\w+)\*$", _PTR),                        # handle
    (r"^const uint8_t\[\d*\]$", C.POINTER(C.c_uint8)),
    (r"^(const )?uint8_t\*$", C.POINTER(C.c_uint8)),
    (r"^(const )?int64_t\*$", C.POINTER(C.c_int64)),
    (r"^(const )?double\*$", C.POINTER(C.c_double)),
    (r"^(const )?int\*$", C.POINTER(C.c_int)),
    (r"^(const )?uint64_t\*$", C.POINTER(C.c_uint64)),
    (r"^(const )?void\*\*$", C.POINTER(_PTR)),
    (r"^(const )?void\*$", _PTR),
    (r"^double$", C.c_double),
    (r"^int$", C.c_int),
    (r"^int64_t$", C.c_int64),
    (r"^uint64_t$", C.c_uint64),
    (r"^size_t$", C.c_size_t),
]


def _ctype_of(param: str):
    p = re.sub(r"/\*.*?\*/", "", param).strip()
    # split off the parameter name (last identifier), keep array suffix with the type
    m = re.match(r"^(.*?)(\b\w+)?(\[\d*\])?$", p)
    arr = ""
    if "[" in p:
        arr = p[p.index("["):]
        p = p[: p.index("[")].strip()
    # drop the name
    toks = p.replace("*", " * ").split()
    if toks and re.match(r"^\w+$", toks[-1]) and toks[-1] not in (
        "int", "double", "int64_t", "uint64_t", "size_t", "char", "void", "uint8_t", "const") and len(toks) > 1:
        toks = toks[:-1]
    t = " ".join(toks).replace(" *", "*").replace("* ", "*")
    t = re.sub(r"\s+", " ", t).strip() + arr
    t = t.replace("* const*", "* const*")
    for rx, ct in _TYPE_MAP:
        if re.match(rx, t):
            return ct
    raise LampError(f"cannot map C parameter type {param!r} (normalised {t!r})")


def parse_header(path: str):
    """Return {name: (restype, [ctypes...], [raw param strings])} for every lamp_* declaration."""
    if not os.path.exists(path):
        return {}
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    decls = {}
    for m in re.finditer(r"\b(int|const char\*)\s+(lamp_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, params = m.group(1), m.group(2), " ".join(m.group(3).split())
        plist = [] if params in ("", "void") else [p.strip() for p in params.split(",")]
        decls[name] = (C.c_char_p if "char" in ret else C.c_int, [_ctype_of(p) for p in plist], plist)
    return decls


class _Lib:
    def __init__(self):
        self._dll = None
        self.decls = {}
        for h in HEADERS:
            self.decls.update(parse_header(h))

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise LampError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # kernel arguments in device memory: the step is ~100 short launches, and fetching each launch's arguments from host memory
        # costs 1 % of the graph-replayed ResNet step and 10 % of the eager one (measured).  Must be in the environment before the HIP
        # runtime initialises; a value the user has set wins.  (The library's own constructor does the same for non-Python hosts.)
        os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
        self._dll = C.CDLL(LIB_PATH, mode=C.RTLD_LOCAL)
        self._dll.lamp_last_error.restype = C.c_char_p
        self.missing = []
        for name, (res, args, _) in self.decls.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:
                self.missing.append(name)  # tests/test_abi.py asserts this list is empty
                continue
            fn.restype = res
            fn.argtypes = args
        return self._dll

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        dll = self.load()
        fn = getattr(dll, name)
        if self.decls.get(name, (C.c_int,))[0] is not C.c_int:
            return fn

        def checked(*a):
            rc = fn(*a)
            if rc != 0:
                raise LampError(dll.lamp_last_error().decode("utf-8", "replace"))
            return rc

        checked.__name__ = name
        setattr(self, name, checked)
        return checked


lib = _Lib()


def i64_array(xs):
    xs = list(xs)
    return (C.c_int64 * max(len(xs), 1))(*xs)


def f64_array(xs):
    xs = list(xs)
    return (C.c_double * max(len(xs), 1))(*xs)


def handle_array(hs):
    hs = list(hs)
    return (_PTR * max(len(hs), 1))(*hs)
