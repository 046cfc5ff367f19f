"""error of the HIP bf16 ResNet gradients against the f32 oracle, next to the ATen-CPU bf16 oracle's (tests/test_resnet_bf16_gpu.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lamp_amd._capi import lib; lib.load()
import torch
from lamp_amd import nn, sten as S
from oracle import lamp_oracle as O
from tests.util import to_sten, to_torch
from tests.test_resnet_bf16_gpu import _three_models, _l2
for B in (256, 2048):
    ob, of, hm = _three_models()
    x = O.closed_form(B * 3 * 32 * 32, 5, 1.0, torch.bfloat16).reshape(B, 3, 32, 32)
    target = (torch.arange(B) * 7) % 100
    model = nn.SupervisedModel(hm, nn.SupervisedModel.NLL, S.STen.ones([100], S.BF16))
    lb, gb = O.training_step(ob, O.nll_loss(100, torch.ones(100, dtype=torch.bfloat16)), x, target, None)
    lf, gf = O.training_step(of, O.nll_loss(100, torch.ones(100)), x.float(), target, None)
    acc = S.STen.zeros([1], S.F64)
    n, hg = model.addTotalLossAndReturnGradientsAndNumExamples(to_sten(x), to_sten(target), acc)
    print("B", B, "loss hip", float(to_torch(acc)[0]) / B, "cpu bf16", float(lb), "f32", float(lf))
    th = tb = tf = 0.0
    for i, (h, b, f) in enumerate(zip(hg, gb, gf)):
        eh, eb, nf = _l2(to_torch(h), f), _l2(b.float(), f), float(f.double().norm())
        ehb = _l2(to_torch(h), b.float())
        th += eh * eh; tb += eb * eb; tf += nf * nf
        print(f"{i:2d} {str(list(f.shape)):18s} hip-f32 {eh / nf:.3e}  cpu16-f32 {eb / nf:.3e}  ratio {eh / eb:.2f}  hip-cpu16 {ehb / nf:.3e}")
    print("all tensors: hip", (th / tf) ** 0.5, "cpu16", (tb / tf) ** 0.5, "ratio", (th / tb) ** 0.5)
