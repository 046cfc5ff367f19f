"""VERDICT r3 item 2(b), second stage: the toy graph survives `rocprofv3 --kernel-trace`, the training step's graph does not.  Which captured
work makes the profiler fault inside hipGraphLaunch?  One case per process:  python3 scripts/graph_bisect.py CASE [replays]
(scripts/graph_bisect.sh runs every case bare and under the profiler)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lamp_amd._capi import lib; lib.load()
from lamp_amd import sten as S, nn, autograd as A

case = sys.argv[1]
replays = int(sys.argv[2]) if len(sys.argv) > 2 else 50
st = C.c_void_p(); lib.lamp_stream_get_from_pool(0, 0, C.byref(st)); lib.lamp_stream_set_current(st)
rng = np.random.default_rng(1)
def t(shape, dt=S.F32, scale=1.0): return S.STen.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32), 0, dtype=dt)
def i64(v): return (C.c_int64 * len(v))(*v)

keep = []
def conv(x, w, b, pad):
    o = C.c_void_p()
    lib.lamp_convolution(C.byref(o), x, w, b, i64([1, 1]), i64([pad, pad]), i64([1, 1]), 2, 0, i64([0, 0]), 1)
    return S.STen(o)

def body():
    if case == "ew3":
        keep.append(((A_ + B_) * B_).relu())
    elif case == "ew200":
        c = A_
        for _ in range(200): c = c + B_
        keep.append(c)
    elif case == "zeros":                      # allocation + fill inside the capture
        z = S.STen.zeros([1 << 18], S.F32, 0)
        keep.append(z + A_)
    elif case == "sum":                        # a reduction (two-stage / atomics)
        keep.append((A_ * B_).sum())
    elif case in ("conv_bf16", "conv_f32", "conv_small_bf16"):
        keep.append(conv(X_, W_, Bi_, 1))
    elif case in ("conv_bwd_bf16", "conv_bwd_f32"):
        out = (C.c_void_p * 3)()
        lib.lamp_convolution_backward(out, GY_, X_, W_, i64([1, 1]), i64([1, 1]), i64([1, 1]), 2, 0, i64([0, 0]), 1, (C.c_uint8 * 3)(1, 1, 1))
        keep.extend(S.STen(out[i]) for i in range(3) if out[i])
    elif case in ("bn_bf16", "bn_f32"):
        out = (C.c_void_p * 3)()
        lib.lamp_native_batch_norm(out, X_, G_, Be_, RM_, RV_, 1, 0.1, 1e-5)
        keep.extend(S.STen(out[i]) for i in range(3) if out[i])
    elif case.startswith("mlp") or case.startswith("resnet"):
        n, grads = MODEL.addTotalLossAndReturnGradientsAndNumExamples(X_, T_, ACC)
        keep.append(grads)
    else:
        raise SystemExit("unknown case " + case)

n = 1 << 18
A_ = S.STen.from_numpy(np.linspace(-1, 1, n, dtype=np.float32), 0)
B_ = S.STen.from_numpy(np.full(n, 0.5, dtype=np.float32), 0)
if case.startswith("conv") or case.startswith("bn"):
    dt = S.BF16 if case.endswith("bf16") else S.F32
    cin = 16 if "small" in case else 128
    hw = 32 if "small" in case else 8
    X_ = t((64, cin, hw, hw), dt); W_ = t((cin, cin, 3, 3), dt, 0.05); Bi_ = t((cin,), dt); GY_ = t((64, cin, hw, hw), dt)
    G_ = t((cin,), dt); Be_ = t((cin,), dt); RM_ = S.STen.zeros([cin], dt, 0); RV_ = S.STen.ones([cin], dt, 0)
if case.startswith("mlp"):
    net = nn.Sequential(nn.MLP(784, 10, [256], S.F32, 0), nn.Fun("logsoftmax", 1))
    MODEL = nn.SupervisedModel(net, nn.SupervisedModel.NLL, S.STen.ones([10], S.F32, 0))
    X_ = t((1024, 784)); T_ = S.STen.from_numpy((np.arange(1024) % 10).astype(np.int64), 0); ACC = S.STen.zeros([1], S.F32, 0)
if case.startswith("resnet"):
    dt = S.BF16 if "bf16" in case else S.F32
    lib.lamp_manual_seed(1234)
    net = nn.resnet(100, 0.0, dt, 0)
    MODEL = nn.SupervisedModel(net, nn.SupervisedModel.NLL, S.STen.ones([100], dt, 0))
    Bn = 2048 if "b2048" in case else 256
    OPT = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(dt == S.BF16))([p.value for p in net.parameters]) if "opt" in case else None
    X_ = t((Bn, 3, 32, 32), dt); T_ = S.STen.from_numpy((np.arange(Bn) % 100).astype(np.int64), 0); ACC = S.STen.zeros([1], S.F32, 0)

lib.lamp_device_synchronize()
body(); lib.lamp_device_synchronize(); keep.clear()          # one eager pass: attributes set, caches filled
lib.lamp_graph_begin_capture()
body()
g = C.c_void_p(); lib.lamp_graph_end_capture(C.byref(g))
for _ in range(replays):
    lib.lamp_graph_launch(g)
    if case.startswith("resnet") and OPT is not None:
        OPT.step(keep[-1], 1.0)                 # the eager optimiser + re-pack launches between replays, as bench.py's step
lib.lamp_device_synchronize()
print(f"graph_bisect {case}: ok, {replays} replays")
