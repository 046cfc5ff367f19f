"""In-kernel phase clocks of ig_wgrad8h_kernel (diagnostic build: -DLAMP_WG8H_STAMPS, LAMP_LIB_PATH=lamp_amd/lib_dbg/liblamp_hip.so)."""
import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N, Cc = 2048, 128
rng = np.random.default_rng(0)
x = S.STen.from_numpy(rng.standard_normal((N, Cc, 8, 8)).astype(np.float32), 0, S.BF16)
w = S.STen.from_numpy((rng.standard_normal((Cc, Cc, 3, 3)) * 0.05).astype(np.float32), 0, S.BF16)
g = S.STen.from_numpy(rng.standard_normal((N, Cc, 8, 8)).astype(np.float32), 0, S.BF16)
one, z = i64_array([1, 1]), i64_array([0, 0])
mask = (C.c_uint8 * 3)(0, 1, 0)
def run():
    out3 = (C.c_void_p * 3)()
    lib.lamp_convolution_backward(out3, g, x, w, one, one, one, 2, 0, z, 1, mask)
    return S.STen(out3[1])
for _ in range(3): run()
lib.lamp_device_synchronize()
dll = lib._dll
n = 256 * 8 * 8
buf = (C.c_uint * n)()
rc = dll.lamp_debug_wg8h_stamps(buf, n)
a = np.frombuffer(buf, dtype=np.uint32).reshape(256, 8, 8).astype(np.float64)
names = ["loop", "late C0", "store", "load issue", "early C0", "C1", "barrier", "kernel"]
pairs = 16
print("rc", rc, "cycles per PAIR of images (mean over 256 workgroups); prio", os.environ.get("LAMP_WGRAD_PRIO", "1"), "stagger", os.environ.get("LAMP_WGRAD_STAGGER", "1"))
for grp, sl in (("waves 0-3 (early)", slice(0, 4)), ("waves 4-7 (late) ", slice(4, 8))):
    m = a[:, sl, :].mean(axis=(0, 1))
    print(grp, " ".join(f"{names[k]} {m[k] / (pairs if k not in (0, 7) else 1):.0f}" for k in range(8)), f"| loop per pair {m[0] / pairs:.0f}")
