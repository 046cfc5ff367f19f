"""does a host -> device transfer run beside the training step?  K eager steps on the main stream with, on a side stream and with NO dependency
between the two, (a) nothing, (b) the PCIe row gather kernel (f32 / u8 records), (c) a plain pinned -> device copy of the same bytes (the DMA
engine).  Perfect overlap: the time of (a)."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S, nn
B, K = 2048, 40
m = nn.resnet(100, 0.0, S.BF16, 0)
model = nn.SupervisedModel(m, nn.SupervisedModel.NLL, S.STen.ones([100], S.BF16, 0))
opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=True)([p.value for p in m.parameters])
x = S.STen.from_numpy(np.random.default_rng(0).standard_normal((B, 3, 32, 32)).astype(np.float32), 0, S.BF16)
t = S.STen.from_numpy((np.arange(B) * 7 % 100).astype(np.int64), 0)
acc = S.STen.zeros([1], S.F64, 0)
N = 20000
pix = (np.arange(N * 3072, dtype=np.int64) % 251).astype(np.uint8).reshape(N, 3, 32, 32)
def pinned(a):
    o = C.c_void_p(); h = S.STen.from_numpy(a, S.CPU); lib.lamp_pin_memory(C.byref(o), h); return S.STen(o)
p32, p8 = pinned(pix.astype(np.float32)), pinned(pix)
idx = S.STen.from_numpy(np.random.default_rng(1).permutation(N)[:B].astype(np.int64), 0)
flat32 = pinned(pix[:B].astype(np.float32))
dst32 = S.STen.zeros([B, 3, 32, 32], S.F32, 0)
main, side = C.c_void_p(), C.c_void_p()
lib.lamp_stream_get_current(0, C.byref(main)); lib.lamp_stream_get_from_pool(0, 0, C.byref(side))
def run(kind):
    for rep in range(2):
        lib.lamp_device_synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            if kind != "none":
                lib.lamp_stream_set_current(side)
                if kind == "gather_f32": o = C.c_void_p(); lib.lamp_index_select_pinned(C.byref(o), p32, idx, S.BF16); S.STen(o)
                elif kind == "gather_u8": o = C.c_void_p(); lib.lamp_index_select_pinned(C.byref(o), p8, idx, S.BF16); S.STen(o)
                elif kind == "copy_f32": lib.lamp_copy_(dst32, flat32, 1)
                lib.lamp_stream_set_current(main)
            model.train_step(opt, x, t, acc, None, 1.0)
        lib.lamp_device_synchronize()
        dt = time.perf_counter() - t0
    print(f"{kind:12s}: {dt / K * 1e3:.3f} ms per step")
for kind in ("none", "gather_f32", "gather_u8", "copy_f32", "none"):
    run(kind)
