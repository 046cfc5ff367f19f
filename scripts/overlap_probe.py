"""Do a batch-norm backward (HBM-bound, no LDS) and a wide weight-gradient kernel (MFMA / LDS-bound) overlap when they are issued on two streams?
Serial on one stream vs concurrent on two, wall time between device synchronisations.  Also the input-gradient kernel against the weight gradient."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib, i64_array
lib.load()
from lamp_amd import sten as S
import numpy as np
N, Cc = 2048, 128
rng = np.random.default_rng(0)
mk = lambda *shape: S.STen.from_numpy(rng.standard_normal(shape).astype(np.float32), 0, S.BF16)
x, dy, w = mk(N, Cc, 8, 8), mk(N, Cc, 8, 8), mk(Cc, Cc, 3, 3)
g, b, rm, rv = mk(Cc), mk(Cc), mk(Cc), S.STen.ones([Cc], S.BF16, 0)
one, p1, z = i64_array([1, 1]), i64_array([1, 1]), i64_array([0, 0])
out3 = (C.c_void_p * 3)()
lib.lamp_native_batch_norm_relu(out3, x, g, b, rm, rv, 1, 0.1, 1e-5)
y, sm, si = [S.STen(h) for h in out3]
m3 = lambda a, b_, c: (C.c_uint8 * 3)(a, b_, c)
def bn_bwd():
    o = (C.c_void_p * 3)()
    lib.lamp_native_batch_norm_relu_backward(o, dy, x, g, b, rm, rv, sm, si, 1, 1e-5, m3(1, 1, 1))
    return [S.STen(h) for h in o]
def wgrad():
    o = (C.c_void_p * 3)()
    lib.lamp_convolution_backward(o, dy, x, w, one, p1, one, 2, 0, z, 1, m3(0, 1, 0))
    return S.STen(o[1])
def dgrad():
    o = (C.c_void_p * 3)()
    lib.lamp_convolution_backward(o, dy, x, w, one, p1, one, 2, 0, z, 1, m3(1, 0, 0))
    return S.STen(o[0])
main = C.c_void_p(); lib.lamp_stream_get_current(0, C.byref(main))
side = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(side))
def timed(fn, R=40):
    for _ in range(3): fn()
    lib.lamp_device_synchronize()
    t0 = time.perf_counter()
    keep = [fn() for _ in range(R)]
    lib.lamp_device_synchronize()
    return (time.perf_counter() - t0) / R * 1e6
def both(f_main, f_side):
    def run():
        lib.lamp_stream_wait_stream(side, main)
        a = f_main()
        lib.lamp_stream_set_current(side)
        b_ = f_side()
        lib.lamp_stream_set_current(main)
        lib.lamp_stream_wait_stream(main, side)
        return a, b_
    return run
for name, f in (("bn_bwd", bn_bwd), ("wgrad", wgrad), ("dgrad", dgrad)):
    print(f"{name} alone: {timed(f):.1f} us")
for (na, fa), (nb, fb) in ((("bn_bwd", bn_bwd), ("wgrad", wgrad)), (("dgrad", dgrad), ("wgrad", wgrad)), (("bn_bwd", bn_bwd), ("dgrad", dgrad))):
    ser = timed(lambda: (fa(), fb()))
    con = timed(both(fa, fb))
    print(f"{na} + {nb}: serial {ser:.1f} us, two streams {con:.1f} us")
lib.lamp_flush_deferred()
