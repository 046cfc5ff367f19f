"""a few batches of the host-resident epoch loop (for rocprofv3 --kernel-trace: does the PCIe gather of batch i + 1 run beside step i?)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S, nn
from lamp_amd.data import BatchStream
N, B = 2048 * 24, 2048
kind = sys.argv[1] if len(sys.argv) > 1 else "host"
pix = (np.arange(N * 3072, dtype=np.int64) % 251).astype(np.uint8).reshape(N, 3, 32, 32).astype(np.float32)
lab = S.STen.from_numpy((np.arange(N, dtype=np.int64) * 7) % 100, S.CPU)
order = np.random.default_rng(1).permutation(N)
m = nn.resnet(100, 0.0, S.BF16, 0)
model = nn.SupervisedModel(m, nn.SupervisedModel.NLL, S.STen.ones([100], S.BF16, 0))
opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=True)([p.value for p in m.parameters])
if kind == "host":
    st = BatchStream.minibatchesFromFull(B, False, S.STen.from_numpy(pix, S.CPU), lab, order=order, hostResident=True, outDtype=S.BF16)
elif kind == "host_u8":
    st = BatchStream.minibatchesFromFull(B, False, S.STen.from_numpy(pix.astype(np.uint8), S.CPU), lab, order=order, hostResident=True, outDtype=S.BF16)
else:
    st = BatchStream.minibatchesFromFull(B, False, S.STen.from_numpy(pix, 0, S.BF16), lab, order=order)
acc = S.STen.zeros([1], S.F64, 0)
for ep in range(3):
    st.reset(); lib.lamp_device_synchronize()
    t = time.perf_counter(); host_next = 0.0; host_step = 0.0
    while True:
        t0 = time.perf_counter()
        b = st.nextBatch()
        t1 = time.perf_counter()
        if b is None: break
        model.train_step(opt, b[0], b[1], acc, None, 1.0)
        t2 = time.perf_counter()
        host_next += t1 - t0; host_step += t2 - t1
    lib.lamp_device_synchronize()
    dt = time.perf_counter() - t
    print(f"{kind} epoch {ep}: {N / dt:9.0f} records/s; per batch {dt / 24 * 1e3:.3f} ms, host time in nextBatch {host_next / 24 * 1e3:.3f} ms, in train_step {host_step / 24 * 1e3:.3f} ms")
