"""lamp-data's epoch loops over the host C ABI.

Mirror of lamp.data.IOLoops.oneEpoch / validationOneEpoch (lamp-data/src/main/scala/lamp/data/IOLoops.scala:607-750, :751-830) and of
the per-batch body of distributed.oneEpoch (distributed/package.scala:733-780).  The loops are host glue in the reference as well (Scala
on the JVM); every batch is one or two calls into liblamp_hip.so:

  accumulateGradientOverNBatches <= 1   lamp_model_train_step  (gradients + optional RCCL exchange + optimizer.step)
  accumulateGradientOverNBatches  > 1   lamp_model_gradients(zero_grad = 0) per batch, optimizer.step + zeroGrad every N-th batch
  validation                            lamp_model_forward_loss on the module in eval mode

The reference's `prefetch` / `overlapModelWithLoad` switches hide the host gather + PCIe copy of the next minibatch; with the data set
resident in HBM (lamp_amd.data.BatchStream) a minibatch is one gather kernel on the same stream and there is nothing to hide.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

from . import sten as S
from .data import BatchStream
from .nn import Optimizer, SupervisedModel


def oneEpoch(epochCount: int, model: SupervisedModel, optimizer: Optimizer, trainBatches: BatchStream, learningRateScheduleFactor: float = 1.0,
             accumulateGradientOverNBatches: int = 1, trainingCallback: Optional[Callable] = None, logger: Optional[Callable[[str], None]] = None,
             comm=None) -> float:
    """One pass over `trainBatches`; returns the average training loss (sum of loss * numInstances over the batches / instances)."""
    first = model.module.state[0].value
    lossAcc = S.STen.zeros([1], S.F64, first.device)        # STen.scalarDouble(0, options): f64 whatever the model type (IOLoops.scala:715)
    numInstances, batchCount = 0, 0
    t1 = time.perf_counter()
    trainBatches.reset()
    if accumulateGradientOverNBatches > 1:
        model.module.zeroGrad()
    for sample, target in trainBatches:
        if accumulateGradientOverNBatches <= 1:
            n = model.train_step(optimizer, sample, target, lossAcc, comm, learningRateScheduleFactor)
        else:
            assert comm is None, "gradient accumulation is a single-process option in the reference (IOLoops.oneEpoch)"
            n, grads = model.addTotalLossAndReturnGradientsAndNumExamples(sample, target, lossAcc, False)
            if batchCount % accumulateGradientOverNBatches == accumulateGradientOverNBatches - 1:
                optimizer.step(grads, learningRateScheduleFactor)
                model.module.zeroGrad()
        numInstances += n
        batchCount += 1
    totalLoss = float(lossAcc.to_numpy().reshape(-1)[0])
    trainingLoss = totalLoss / max(numInstances, 1)
    if logger is not None:
        seconds = time.perf_counter() - t1
        logger(f"Avg training loss in epoch {epochCount} over {numInstances} examples: {trainingLoss} ({numInstances / seconds:.2f} instances/sec)")
    if trainingCallback is not None:
        trainingCallback(epochCount, trainingLoss, model.module)
    return trainingLoss


def validationOneEpoch(model: SupervisedModel, validationBatches: BatchStream, epochCount: int = 0, validationCallback: Optional[Callable] = None,
                       logger: Optional[Callable[[str], None]] = None) -> float:
    """Average validation loss with the module in eval mode (restored to training mode afterwards, as `model.asEval` is a copy there)."""
    first = model.module.state[0].value
    totalLoss = S.STen.zeros([1], S.F64, first.device)      # f64 accumulator (IOLoops.scala:809)
    totalExamples = 0
    model.module.asEval()
    try:
        validationBatches.reset()
        for sample, target in validationBatches:
            totalExamples += model.addTotalLossAndReturnNumExamples(sample, target, totalLoss)
    finally:
        model.module.asTraining()
    validationLoss = float(totalLoss.to_numpy().reshape(-1)[0]) / max(totalExamples, 1)
    if logger is not None:
        logger(f"Avg validation loss in epoch {epochCount} over {totalExamples} examples: {validationLoss}")
    if validationCallback is not None:
        validationCallback(epochCount, validationLoss)
    return validationLoss
