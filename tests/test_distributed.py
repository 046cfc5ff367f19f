"""N > 1 logic on CPU: two gloo processes run the exchange protocol of the data-parallel step.

What runs here is the host logic (rendezvous, bucket layout, example-weighted averaging contract, sharding); the device
kernels of the same step (lamp_flatten_into_ / lamp_comm_all_reduce / lamp_unflatten_from_) are covered on the GPU in
test_flat_bucket_and_single_rank_collectives, and the complete overlapped two-bucket step (second stream, events, AdamW) in
test_overlapped_data_parallel_step_single_rank (world size 1 communicator).
"""
import ctypes as C
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from lamp_amd import distributed as D
from oracle import lamp_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist = D.init_control_plane()
    # 1. unique id rendezvous: every rank ends up with root's 128 bytes
    uid = D.exchange_unique_id(dist, lambda: bytes(range(128)))
    assert uid == bytes(range(128))
    # 2. one data-parallel exchange: local gradients of a small model on this rank's shard, flat bucket, all-reduce(sum), / sum n
    torch.manual_seed(0)
    m = O.Sequential(O.mlp(12, 3, [8], torch.float64), O.Fun(lambda v: v.logSoftMax(1)))
    n_r = [5, 3][rank]                                        # uneven shards: the weighting matters
    x = O.closed_form(8 * 12, 0, 1.0, torch.float64).reshape(8, 12)[sum([5, 3][:rank]):][:n_r]
    t = (torch.arange(8) % 3)[sum([5, 3][:rank]):][:n_r]
    _, grads = O.training_step(m, O.nll_loss(3, torch.ones(3, dtype=torch.float64)), x, t, None)
    offs, total = D.bucket_layout([g.numel() for g in grads])
    bucket = torch.zeros(total, dtype=torch.float32)
    for g, o in zip(grads, offs):                              # lamp_flatten_into_(bucket, grads, scale = n_r)
        bucket[o:o + g.numel()] = (g.reshape(-1) * n_r).float()
    bucket[-1] = n_r
    dist.all_reduce(bucket)                                    # lamp_comm_all_reduce
    avg = [(bucket[o:o + g.numel()] / bucket[-1]).reshape(g.shape) for g, o in zip(grads, offs)]   # lamp_unflatten_from_
    torch.save({"grads": grads, "avg": avg, "n": n_r}, os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_exchange(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    ref = O.average_gradients([r0["grads"], r1["grads"]], [r0["n"], r1["n"]])      # distributed/package.scala:690-719
    for a0, a1, r in zip(r0["avg"], r1["avg"], ref):
        assert torch.equal(a0, a1), "every rank must hold the same averaged gradient"
        assert torch.allclose(a0.double(), r, rtol=1e-6, atol=1e-7)


def test_every_nth_sharding():
    assert D.every_nth(10, 4, 1) == [1, 5]            # two full rounds; the tail (8, 9) is dropped on every rank
    shards = [D.every_nth(12, 3, r) for r in range(3)]
    assert sorted(sum(shards, [])) == list(range(12)) and len({len(s) for s in shards}) == 1
    assert D.bucket_layout([4, 6, 2]) == ([0, 4, 10], 13)


@pytest.mark.gpu
def test_flat_bucket_and_single_rank_collectives(gpu):
    from lamp_amd import sten as S
    from lamp_amd._capi import lib, handle_array
    g = [S.STen.from_numpy(np.arange(6, dtype=np.float32).reshape(2, 3)), S.STen.from_numpy(np.ones(4, dtype=np.float32) * 2)]
    offs, total = D.bucket_layout([t.numel for t in g])
    bucket = S.STen.zeros([total], S.F32)
    lib.lamp_flatten_into_(bucket, handle_array([t.h for t in g]), 2, 3.0)
    bucket.slice(0, total - 1, total).fill_(3.0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)
    lib.lamp_comm_all_reduce(handle_array([bucket.h]), handle_array([comm]), 1, 0)
    lib.lamp_comm_broadcast(handle_array([bucket.h]), handle_array([comm]), 1, 0)
    lib.lamp_comm_reduce(handle_array([bucket.h]), bucket, 0, 0, handle_array([comm]), 1)
    assert np.allclose(bucket.to_numpy(), np.concatenate([np.arange(6) * 3.0, np.ones(4) * 6.0, [3.0]]))
    out = [S.STen.zeros([2, 3]), S.STen.zeros([4])]
    lib.lamp_unflatten_from_(handle_array([t.h for t in out]), 2, bucket, 1)
    assert np.allclose(out[0].to_numpy(), np.arange(6).reshape(2, 3)) and np.allclose(out[1].to_numpy(), 2.0)
    lib.lamp_comm_destroy(comm)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["f32", "bf16"])
def test_overlapped_data_parallel_step_single_rank(gpu, dtype_name):
    """lamp_model_train_step with a world-size-1 RCCL communicator runs the whole two-bucket / second-stream exchange
    (pack * n, all-reduce, / sum n, AdamW).  With one rank the averaged gradient is the local one, so three steps must
    land on the parameters of the plain step (n = 64 is a power of two: *n and /n are exact)."""
    from lamp_amd import sten as S, nn
    from lamp_amd._capi import lib
    dt = S.F32 if dtype_name == "f32" else S.BF16
    B = 64
    x = S.STen.from_numpy((np.arange(B * 3 * 32 * 32) * 7919 % 1009 / 1009.0 - 0.5).reshape(B, 3, 32, 32).astype(np.float32), 0, dt)
    target = S.STen.from_numpy((np.arange(B) * 7 % 100).astype(np.int64), 0)
    uid = (C.c_uint8 * 128)(); lib.lamp_comm_get_unique_id(uid)
    comm = C.c_void_p(); lib.lamp_comm_init_rank(C.byref(comm), 1, uid, 0)

    def run(use_comm):
        lib.lamp_manual_seed(99)
        mod = nn.resnet(100, 0.0, dt, 0)
        model = nn.SupervisedModel(mod, nn.SupervisedModel.NLL, S.STen.ones([100], dt, 0))
        opt = nn.AdamW_factory(weightDecay=0.0, learningRate=1e-3, mixedPrecision=(dt == S.BF16))([p.value for p in mod.parameters])
        acc = S.STen.zeros([1], dt, 0)
        for _ in range(3):
            assert model.train_step(opt, x, target, acc, comm if use_comm else None) == B
        lib.lamp_device_synchronize()
        return [s.value.castToDouble().to_numpy() for s in mod.state], acc.castToDouble().to_numpy()

    plain, acc0 = run(False)
    dp, acc1 = run(True)
    lib.lamp_comm_destroy(comm)
    assert np.array_equal(acc0, acc1)
    for a, b in zip(plain, dp):
        assert np.array_equal(a, b), "single-rank data-parallel step must reproduce the plain step bit for bit"


def test_row_shards_cover_every_row_once():
    """host logic of the row-sharded kNN graph: the per-rank blocks partition the rows, and the fixed-size contributions
    of the all-gather (short / empty last blocks re-query the tail rows) can always be cut back to exactly those blocks."""
    for n in (1, 7, 64, 1000, 1001):
        for world in (1, 2, 3, 8):
            per = (n + world - 1) // world
            seen = []
            for r in range(world):
                lo, hi = D.row_shard(n, world, r)
                assert 0 <= lo <= hi <= n and hi - lo <= per
                seen += list(range(lo, hi))
                if hi > lo and per <= n:
                    q = max(min(lo, n - per), 0)                 # first row this rank really queries
                    assert q <= lo and q + per >= hi, "the shifted block still contains the rank's rows"
            assert seen == list(range(n))


@pytest.mark.gpu
def test_sharded_knn_single_rank_and_all_gather(gpu):
    """world size 1: the sharded driver returns the plain graph; lamp_comm_all_gather on a 1-rank RCCL communicator copies."""
    from lamp_amd import sten as S, umap as U
    from lamp_amd._capi import lib
    rng = np.random.default_rng(3)
    X = S.STen.from_numpy(rng.random((500, 64), dtype=np.float32), 0)
    ref = U.knn_search(X, X, 5).to_numpy()
    assert np.array_equal(D.knn_search_sharded(X, 5, None, 1, 0).to_numpy(), ref)
    buf = (C.c_uint8 * 128)()
    lib.lamp_comm_get_unique_id(buf)
    h = C.c_void_p()
    lib.lamp_comm_init_rank(C.byref(h), 1, buf, 0)
    try:
        got = D.knn_search_sharded(X, 5, h, 1, 0).to_numpy()
        assert np.array_equal(got, ref)
        # the sharded UMAP layout driver through a real (1-rank) communicator: its all-reduces run and change nothing
        dist = np.sort(rng.random((500, 5)), 1); dist[:, 0] = 0.0
        ew = U.edge_weights(S.STen.from_numpy(dist, 0, S.F64), S.STen.from_numpy(ref.astype(np.int64), 0))
        l1, v1 = U.optimize(ew, 500, 0.1, 4, 0.0, 5, 42, True, 1.0, 0, 2)
        l2, v2 = U.optimize_sharded(ew, 500, 0.1, 4, 0.0, 5, 42, True, 1.0, 0, 2, h, 1, 0)
        assert abs(v1 - v2) <= 1e-9 * abs(v1) and np.abs(l1.to_numpy() - l2.to_numpy()).max() <= 1e-9
        src = S.STen.from_numpy(np.arange(12, dtype=np.int64).reshape(3, 4), 0)
        dst = S.STen.zeros([3, 4], S.I64, 0)
        lib.lamp_comm_all_gather(dst, src, h)
        assert np.array_equal(dst.to_numpy(), src.to_numpy())
        # the reassembly for several ranks, driven with the single-process pieces a 3-rank job would produce
        n, world, k = 500, 3, 5
        per = (n + world - 1) // world
        blocks = []
        for r in range(world):
            lo, hi = D.row_shard(n, world, r)
            q = max(min(lo, n - per), 0)
            blocks.append(U.knn_search(X, X.slice(0, q, q + per), k).to_numpy())
        gathered = np.concatenate(blocks, 0)
        parts = []
        for r in range(world):
            lo, hi = D.row_shard(n, world, r)
            q = max(min(lo, n - per), 0)
            parts.append(gathered[r * per + (lo - q): r * per + (lo - q) + (hi - lo)])
        assert np.array_equal(np.concatenate(parts, 0), ref)
    finally:
        lib.lamp_comm_destroy(h)


@pytest.mark.gpu
@pytest.mark.parametrize("accumulate", [1, 2])
def test_single_process_data_parallel_step(gpu, accumulate):
    """DataParallel.synchronousStep (lamp-data DataParallel.scala:195-311) with the main model and two replicas (all on the one GPU of
    the test box: same host threads, copies, weighting and reduction as with one GPU each): replicas receive the main state, the
    example-weighted mean gradient and the AdamW step equal the oracle's, the loss accumulators are per model."""
    from lamp_amd import nn, sten as S
    from tests.util import to_sten, to_torch, assert_close, closed_form
    dt = torch.float64
    def oracle_model(): return O.Sequential(O.mlp(12, 3, [8], dt), O.Fun(lambda v: v.logSoftMax(1)))
    def hip_model(): return nn.Sequential(nn.MLP(12, 3, [8], S.F64), nn.Fun("logsoftmax", 1))
    om = oracle_model()
    hmods = [hip_model() for _ in range(3)]
    hmods[0].load([to_sten(v.value) for v in om.state()])            # replicas start with different (random) weights
    cw = torch.ones(3, dtype=dt)
    models = [nn.SupervisedModel(m, nn.SupervisedModel.NLL, to_sten(cw)) for m in hmods]
    hopt = nn.AdamW([p.value for p in hmods[0].parameters], weightDecay=0.01, learningRate=1e-2)
    oopt = O.AdamW([p.value for p in om.parameters()], weightDecay=0.01, learningRate=1e-2)
    sizes = [5, 3, 7]
    accs = [S.STen.zeros([1], S.F64) for _ in range(3)]
    expected_acc = [0.0, 0.0, 0.0]
    for it in range(2 * accumulate):
        xs = [closed_form((n, 12), 100 * it + 10 * i, 2.0, dt) for i, n in enumerate(sizes)]
        ts = [(torch.arange(n) + i + it) % 3 for i, n in enumerate(sizes)]
        zero, step = it % accumulate == 0, it % accumulate == accumulate - 1
        total = nn.dataParallelSynchronousStep(models[0], hopt, models[1:], [(to_sten(x), to_sten(t)) for x, t in zip(xs, ts)], accs,
                                               zeroGrad=zero, step=step)
        assert total == sum(sizes)
        # oracle: every model holds the main state; gradients accumulate over `accumulate` batches per model, then the reference's
        # in-place `grad *= n` of the last batch, sum over models, / total of the last batch
        if zero:
            per_model = [None] * 3
        for i in range(3):
            loss = om.forward(O.const(xs[i])).nllLoss(ts[i], cw, 1, -100)
            expected_acc[i] += loss.value.item() * sizes[i]
            g = [x.clone() for x in om.gradients(loss)]
            per_model[i] = g if per_model[i] is None else [a + b for a, b in zip(per_model[i], g)]
        # the replicas now hold the state the main model had at the start of this step
        for r in hmods[1:]:
            for hv, ov in zip(r.parameters, om.parameters()):        # (the batch-norm running statistics move with every forward)
                assert_close(to_torch(hv.value), ov.value, 1e-12, "replica parameters = main parameters")
        if step:
            avg = [sum(per_model[i][k] * sizes[i] for i in range(3)) / sum(sizes) for k in range(len(per_model[0]))]
            oopt.step(avg, 1.0)
            for hv, ov in zip(hmods[0].parameters, om.parameters()):
                assert_close(to_torch(hv.value), ov.value, 1e-10, "main parameters after the step")
    for a, e in zip(accs, expected_acc):
        assert abs(float(a.to_numpy()[0]) - e) <= 1e-9 * abs(e)
