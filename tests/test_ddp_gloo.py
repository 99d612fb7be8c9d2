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
