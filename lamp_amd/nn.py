"""lamp.nn mirror over the host C ABI: modules, optimisers, SupervisedModel, training steps.

Reference: lamp-core/src/main/scala/lamp/nn/{Module,Linear,Conv2D,BatchNorm,BatchNorm2D,LayerNorm,
MLP,AdamW,SGD,SupervisedModel,LossFunctions}.scala and example-cifar100/.../cnn.scala.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

from ._capi import lib, i64_array, handle_array
from .autograd import Variable
from .sten import STen, F32


class Module:
    def __init__(self, handle):
        self.h = handle.value if isinstance(handle, C.c_void_p) else handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                lib.lamp_module_release(h)
            except Exception:
                pass

    @property
    def _as_parameter_(self):
        return C.c_void_p(self.h)

    def forward(self, x: Variable) -> Variable:
        o = C.c_void_p(); lib.lamp_module_forward(self.h, x.h, C.byref(o)); return Variable(o)

    @property
    def state(self) -> List[Variable]:
        n = C.c_int64(); lib.lamp_module_num_state(self.h, C.byref(n))
        out = []
        for i in range(n.value):
            o = C.c_void_p(); lib.lamp_module_state(self.h, i, C.byref(o)); out.append(Variable(o))
        return out

    @property
    def parameters(self) -> List[Variable]:
        return [v for v in self.state if v.needsGrad]

    def zeroGrad(self): lib.lamp_module_zero_grad(self.h)
    def asEval(self): lib.lamp_module_set_training(self.h, 0); return self
    def asTraining(self): lib.lamp_module_set_training(self.h, 1); return self

    def gradients(self, loss: Variable, zeroGrad=True):
        """Module.gradients (Module.scala:300-314)."""
        if zeroGrad:
            self.zeroGrad()
        loss.backprop()
        return [p.partialDerivative for p in self.parameters]

    def load(self, tensors: Sequence[STen]):
        """Load.make: copy into the state tensors in order."""
        st = self.state
        assert len(st) == len(tensors), f"state has {len(st)} tensors, got {len(tensors)}"
        for v, t in zip(st, tensors):
            v.value.copyFrom(t)


def _mk(fn, *args) -> Module:
    o = C.c_void_p(); getattr(lib, fn)(C.byref(o), *args); return Module(o)


def Linear(in_, out, dtype=F32, device=0, bias=True): return _mk("lamp_module_linear", in_, out, dtype, device, int(bias))
def Conv2D(inChannels, outChannels, kernelSize, dtype=F32, device=0, bias=False, stride=1, padding=0, dilation=1, groups=1):
    return _mk("lamp_module_conv2d", inChannels, outChannels, kernelSize, dtype, device, int(bias), stride, padding, dilation, groups)
def BatchNorm(features, dtype=F32, device=0): return _mk("lamp_module_batch_norm", features, dtype, device, 0)
def BatchNorm2D(features, dtype=F32, device=0): return _mk("lamp_module_batch_norm", features, dtype, device, 1)
def LayerNorm(normalizedShape, dtype=F32, device=0, scale=True, bias=True):
    return _mk("lamp_module_layer_norm", i64_array(normalizedShape), len(normalizedShape), dtype, device, int(scale), int(bias))
def Dropout(p): return _mk("lamp_module_dropout", float(p))
def Fun(name, a=0.0, b=0.0): return _mk("lamp_module_fun", name.encode(), float(a), float(b))
def Sequential(*mods): return _mk("lamp_module_sequential", handle_array([m.h for m in mods]), len(mods))
def Residual(right, left=None): return _mk("lamp_module_residual", right.h, left.h if left is not None else None)
def MLP(in_, out, hidden, dtype=F32, device=0, dropout=0.0, lastNonLinearity=False, activationFunction="relu", norm="BatchNorm", bias=True):
    code = {"NoNorm": 0, "BatchNorm": 1, "LayerNorm": 2}[norm]
    return _mk("lamp_module_mlp", in_, out, i64_array(hidden), len(hidden), dtype, device, float(dropout), int(lastNonLinearity),
               activationFunction.encode(), code, int(bias))
def resnet(numClasses, dropout=0.0, dtype=F32, device=0):
    """Cnn.resnet (example-cifar100/.../cnn.scala:89-137)."""
    return _mk("lamp_module_resnet", numClasses, float(dropout), dtype, device)


class Optimizer:
    def __init__(self, handle):
        self.h = handle.value if isinstance(handle, C.c_void_p) else handle

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                lib.lamp_optimizer_release(h)
            except Exception:
                pass

    def step(self, gradients: Sequence[Optional[STen]], scheduleFactor=1.0):
        hs = handle_array([g.h if g is not None else None for g in gradients])
        lib.lamp_optimizer_step(self.h, hs, len(gradients), float(scheduleFactor))

    def load(self, tensors: Sequence[STen]):
        """Optimizer.load (AdamW.scala:87-93): copy into the state tensors in order, restore the step count."""
        lib.lamp_optimizer_load(self.h, handle_array([t.h for t in tensors]), len(tensors))

    @property
    def state(self) -> List[STen]:
        n = C.c_int64(); lib.lamp_optimizer_num_state(self.h, C.byref(n))
        out = []
        for i in range(n.value):
            o = C.c_void_p(); lib.lamp_optimizer_state(self.h, i, C.byref(o)); out.append(STen(o))
        return out


def AdamW(parameters: Sequence[STen], weightDecay, learningRate=0.001, beta1=0.9, beta2=0.999, eps=1e-8, clip=None, debias=True,
          mixedPrecision=False) -> Optimizer:
    o = C.c_void_p()
    lib.lamp_optimizer_adamw(C.byref(o), handle_array([p.h for p in parameters]), len(parameters), float(weightDecay), float(learningRate),
                             float(beta1), float(beta2), float(eps), -1.0 if clip is None else float(clip), int(debias), int(mixedPrecision))
    opt = Optimizer(o)
    opt._keep = list(parameters)
    return opt


def AdamW_tagged(parameters: Sequence[STen], weightDecay: Sequence[float], learningRate, beta1=0.9, beta2=0.999, eps=1e-8, clip=None, debias=True,
                 mixedPrecision=False) -> Optimizer:
    """AdamW whose hyperparameters are functions of the parameter tag (AdamW.scala:29-47): one value per parameter (scalars are repeated)."""
    from ._capi import f64_array
    n = len(parameters)
    per = lambda v: f64_array([float(x) for x in v] if hasattr(v, "__len__") else [float(v)] * n)
    o = C.c_void_p()
    lib.lamp_optimizer_adamw_tagged(C.byref(o), handle_array([p.h for p in parameters]), n, per(weightDecay), per(learningRate), per(beta1), per(beta2),
                                    float(eps), -1.0 if clip is None else float(clip), int(debias), int(mixedPrecision))
    opt = Optimizer(o)
    opt._keep = list(parameters)
    return opt


def AdamW_factory(weightDecay, learningRate=0.001, beta1=0.9, beta2=0.95, eps=1e-8, clip=None, debias=True, mixedPrecision=False):
    """AdamW.factory: note beta2 = 0.95 (AdamW.scala:12)."""
    return lambda params: AdamW(params, weightDecay, learningRate, beta1, beta2, eps, clip, debias, mixedPrecision)


def SGDW(parameters: Sequence[STen], learningRate, weightDecay, momentum=None, clip=None) -> Optimizer:
    o = C.c_void_p()
    lib.lamp_optimizer_sgdw(C.byref(o), handle_array([p.h for p in parameters]), len(parameters), float(learningRate), float(weightDecay),
                            -1.0 if momentum is None else float(momentum), -1.0 if clip is None else float(clip))
    opt = Optimizer(o)
    opt._keep = list(parameters)
    return opt


def gradientClippingInPlace(gradients: Sequence[Optional[STen]], theta: float):
    lib.lamp_gradient_clipping_in_place(handle_array([g.h if g is not None else None for g in gradients]), len(gradients), float(theta))


class SupervisedModel:
    """SupervisedModel(module, LossFunctions.NLL(numClasses, classWeights)) - SupervisedModel.scala:151-211."""

    NLL, MSE, IDENTITY = 0, 1, 2

    def __init__(self, module: Module, loss_kind=0, classWeights: Optional[STen] = None, reduction=1, ignore=-100):
        o = C.c_void_p()
        lib.lamp_model_create(C.byref(o), module.h, loss_kind, classWeights.h if classWeights is not None else None, reduction, ignore)
        self.h = o.value
        self.module = module
        self._cw = classWeights

    def __del__(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            try:
                lib.lamp_model_release(h)
            except Exception:
                pass

    def addTotalLossAndReturnGradientsAndNumExamples(self, samples: STen, target: STen, acc: Optional[STen], zeroGrad=True):
        n = C.c_int64()
        lib.lamp_model_gradients(self.h, samples.h, target.h, acc.h if acc is not None else None, int(zeroGrad), C.byref(n))
        return n.value, [p.partialDerivative for p in self.module.parameters]

    def addTotalLossAndReturnNumExamples(self, samples: STen, target: STen, acc: Optional[STen]):
        n = C.c_int64()
        lib.lamp_model_forward_loss(self.h, samples.h, target.h, acc.h if acc is not None else None, C.byref(n))
        return n.value

    def train_step(self, optimizer: Optimizer, samples: STen, target: STen, acc: Optional[STen] = None, comm=None,
                   scheduleFactor: float = 1.0) -> int:
        """one batch of IOLoops.oneEpoch / distributed oneBatch: gradients (+ all-reduce) + optimizer.step."""
        n = C.c_int64()
        if scheduleFactor == 1.0:
            lib.lamp_model_train_step(self.h, optimizer.h, comm, samples.h, target.h, acc.h if acc is not None else None, C.byref(n))
        else:
            lib.lamp_model_train_step_scheduled(self.h, optimizer.h, comm, samples.h, target.h, acc.h if acc is not None else None,
                                                float(scheduleFactor), C.byref(n))
        return n.value

    def exchange_and_step(self, optimizer: Optimizer, grads: Sequence[STen], numExamples: int, comm, scheduleFactor: float = 1.0) -> None:
        """averageGradients + optimizer.step on gradients computed elsewhere (a replayed HIP graph): distributed/package.scala:690-759."""
        arr = (C.c_void_p * len(grads))(*[g.h for g in grads])
        lib.lamp_model_exchange_and_step(self.h, optimizer.h, comm, arr, len(grads), int(numExamples), float(scheduleFactor))

    def sync_state(self, optimizer: Optimizer, comm, root: int = 0) -> None:
        """distributed `broadcast` (distributed/package.scala:683-688): rank `root`'s module + optimiser state on every rank."""
        lib.lamp_model_sync_state(self.h, optimizer.h, comm, int(root))


def dataParallelSynchronousStep(mainModel: SupervisedModel, optimizer: Optimizer, models: Sequence[SupervisedModel], batches, accs=None,
                                zeroGrad=True, step=True, scheduleFactor=1.0) -> int:
    """DataParallel.driveSynchronousLoop's synchronousStep (lamp-data DataParallel.scala:195-311): one process, the main model + optimiser
    on one GPU and one replica per other GPU.  `batches` = [(samples, target)] and `accs` = [loss accumulator] per model, main first."""
    n = len(models) + 1
    assert len(batches) == n, "assertion failed: batch.size == models.size + 1"
    arr = lambda hs: (C.c_void_p * n)(*hs)
    num = C.c_int64()
    lib.lamp_data_parallel_step(mainModel.h, optimizer.h, handle_array([m.h for m in models]) if models else None, len(models),
                                arr([b[0].h for b in batches]), arr([b[1].h for b in batches]),
                                arr([(a.h if a is not None else None) for a in accs]) if accs is not None else None,
                                int(zeroGrad), int(step), float(scheduleFactor), C.byref(num))
    return num.value
