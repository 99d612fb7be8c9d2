"""CPU, world_size 2, gloo: the data-parallel gradient exchange (FlatParams + DataParallel.all_reduce_grads)
— rank-0 broadcast at start, bucketed sum all-reduce, 1/world folded into the optimiser step.  The model here is a
plain torch module (the HIP path needs a GPU); the exchange code is the one bench.py runs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "models-for-relational-multimodal-data_amd"))
    from tabgnn_amd.train import DataParallel, FlatParams
    torch.manual_seed(100 + rank)                      # ranks start from DIFFERENT weights
    model = torch.nn.Sequential(torch.nn.Linear(13, 7), torch.nn.ReLU(), torch.nn.BatchNorm1d(7), torch.nn.Linear(7, 3))
    flat = FlatParams(model)
    ddp = DataParallel(model, flat, bucket_mb=0.0001)   # tiny buckets -> several in-flight all-reduces
    start = flat.flat.clone()
    torch.manual_seed(7 + rank)                         # ...and see different mini-batches
    x = torch.randn(16, 13)
    flat.zero_grad()
    model(x).pow(2).mean().backward()
    local = flat.grad.clone()
    scale = ddp.all_reduce_grads()
    out[rank] = dict(start=start, local=local, summed=flat.grad.clone(), scale=scale,
                     views_ok=all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o
                                  for p, o in zip(flat.params, flat.offsets)),
                     bn_mean=model[2].running_mean.clone())
    dist.destroy_process_group()


def test_gradient_all_reduce_two_ranks():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    a, b = out[0], out[1]
    assert torch.equal(a["start"], b["start"])                        # broadcast from rank 0
    assert not torch.equal(a["local"], b["local"])
    want = a["local"] + b["local"]
    assert torch.allclose(a["summed"], want) and torch.equal(a["summed"], b["summed"])
    assert a["scale"] == 0.5 and a["views_ok"] and b["views_ok"]
    assert not torch.equal(a["bn_mean"], b["bn_mean"])                # BatchNorm statistics stay per rank


def _overlap_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "models-for-relational-multimodal-data_amd"))
    from tabgnn_amd.train import DataParallel, FlatParams
    torch.manual_seed(3)
    blk = lambda i, o: torch.nn.Sequential(torch.nn.Linear(i, o), torch.nn.ReLU())
    stages = [blk(13, 16), blk(16, 16), blk(16, 16), torch.nn.Linear(16, 3)]
    model = torch.nn.Sequential(*stages)
    flat = FlatParams(model)
    ddp = DataParallel(model, flat, bucket_mb=0.0001)
    torch.manual_seed(7 + rank)
    x = torch.randn(16, 13)

    def step(overlap):
        flat.zero_grad()
        model(x).pow(2).mean().backward()
        local = None if overlap else flat.grad.clone()
        scale = ddp.all_reduce_grads()
        return flat.grad.clone(), local, scale

    plain, local, _ = step(False)
    # overlapped: record what each range held when its all-reduce was issued, and whether the backward was still running
    ddp.enable_overlap(stages)
    issued = []
    inner = ddp._issue

    def spy(lo, hi):
        issued.append((lo, hi, flat.grad[lo:hi].clone(), bool((stages[0][0].weight.grad == 0).all())))
        inner(lo, hi)
    ddp._issue = spy
    calls0 = ddp.calls
    over, _, scale = step(True)
    early = [r for r in issued if r[3]]                      # issued while stage 0's weight gradient was still zero
    res = dict(equal=torch.equal(plain, over), scale=scale, n_early=len(early), n_issued=len(issued),
                     final_when_issued=all(torch.equal(v, local[lo:hi]) for lo, hi, v, _ in issued),
                     covered=sum(hi - lo for lo, hi, _, _ in issued) == flat.grad.numel(),
                     overlapped_bytes=ddp.overlapped_bytes, calls=ddp.calls - calls0)
    # a second overlapped step gives the same result (the per-step state re-arms)
    again, _, _ = step(True)
    res["again"] = torch.equal(again, plain)
    ddp.disable_overlap()
    off, _, _ = step(False)
    res["off"] = torch.equal(off, plain)
    out[rank] = res
    dist.destroy_process_group()


def test_bucket_ready_all_reduce_equals_the_post_backward_exchange_bit_for_bit():
    """SURVEY 8e "all-reduce overlapped with backward": the gradient ranges of the later stages are exchanged from
    autograd pre-hooks while the earlier stages' backward still runs (DataParallel.enable_overlap).  Each range holds its
    FINAL local gradient when its all-reduce is issued, every element is exchanged exactly once, and the summed
    gradients equal the plain post-backward exchange bit for bit."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        o = out[r]
        assert o["equal"] and o["again"] and o["off"], o
        assert o["final_when_issued"] and o["covered"], o
        assert o["n_early"] >= 3 and o["n_early"] < o["n_issued"], o        # stages 1..3 early, stage 0 after the backward
        assert o["overlapped_bytes"] > 0 and o["scale"] == 0.5 and o["calls"] >= o["n_issued"]
