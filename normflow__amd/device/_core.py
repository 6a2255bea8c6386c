"""One process per GPU, data-parallel over the batch (reference: src/device/_core.py).

MI355X-first differences from the reference's DDP wrapper:
  * no DistributedDataParallel: its bucketed reducer and its per-forward broadcast of
    every buffer (the deterministic uint8 masks included) are replaced by ONE flat
    gradient buffer and ONE `all_reduce` per step (<= 2.3 MB for the BASELINE configs:
    latency-bound on xGMI, so a single collective is the right shape);
  * parameters are broadcast from rank 0 once, when the group is formed;
  * `net_` is not wrapped, so snapshots carry no `module.` key prefix;
  * the backend string is "nccl" (= RCCL on ROCm) on GPUs and "gloo" on CPU hosts,
    the rendezvous address is 127.0.0.1.
Sampling / forward needs no collective at all: samples are independent.
"""
import os
import warnings
from functools import partial

import torch
import torch.distributed as dist
from torch.multiprocessing.spawn import ProcessException
from torch._utils import _flatten_dense_tensors, _unflatten_dense_tensors


class ModelDeviceHandler:
    """Device placement and data-parallel plumbing of a Model."""

    def __init__(self, model):
        self._model = model
        self.nranks = 1
        self.rank = 0

    def to(self, *args, **kwargs):
        self._model.net_.to(*args, **kwargs)
        self._model.prior.to(*args, **kwargs)

    # ---- data parallelism
    def ddp_wrapper(self, rank, nranks, device=None):
        """Attach this process to the group: place prior/net_ on its GPU, make every rank
        start from rank 0's parameters.  (Name kept from the reference, _core.py:39-49.)"""
        if device is None:
            device = torch.device('cuda', rank) if torch.cuda.is_available() else torch.device('cpu')
        if device.type == 'cuda':
            torch.cuda.set_device(device)
        self._model.prior.to(device=device)
        self._model.net_.to(device=device)
        if device.type == 'cpu':
            self._unshare_cpu_storage()
        self.nranks, self.rank = nranks, rank
        self.broadcast_parameters()

    def _unshare_cpu_storage(self):
        """torch.multiprocessing hands CPU tensors to spawned children through SHARED memory: without this every rank of a
        CPU (gloo) run would step the same parameter storage concurrently.  On a GPU `.to(device)` already made the
        rank's own copy.  Each rank takes a private copy of the parameters and buffers here."""
        with torch.no_grad():
            for t in list(self._model.net_.parameters()) + list(self._model.net_.buffers()):
                if t.is_shared():
                    t.data = t.detach().clone()

    def broadcast_parameters(self, src=0):
        """Every rank takes rank `src`'s parameters.  The copy goes through the parameter itself (under no_grad), not
        through `.data`: it bumps the tensors' version counters, which the split-fp16 range check of the conv kernels
        keys on (`_hip._weights_fit_fp16`); that cache is dropped as well."""
        if self.nranks == 1:
            return
        params = list(self._model.net_.parameters())
        with torch.no_grad():
            for dtype in {p.dtype for p in params}:
                group = [p for p in params if p.dtype == dtype]
                flat = _flatten_dense_tensors([p.detach() for p in group])
                dist.broadcast(flat, src=src)
                for p, new in zip(group, _unflatten_dense_tensors(flat, group)):
                    p.copy_(new)
        from .. import _hip
        _hip.invalidate_weight_checks()

    def all_reduce_gradients(self):
        """Mean of the gradients over ranks: one flat buffer, one collective."""
        if self.nranks == 1:
            return
        grads = [p.grad for p in self._model.net_.parameters() if p.grad is not None]
        for dtype in {g.dtype for g in grads}:
            group = [g for g in grads if g.dtype == dtype]
            flat = _flatten_dense_tensors(group)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(self.nranks)
            for g, new in zip(group, _unflatten_dense_tensors(flat, group)):
                g.copy_(new)

    def all_gather_into_tensor(self, x):
        if self.nranks == 1:
            return x
        out = torch.zeros((x.shape[0] * self.nranks, *x.shape[1:]), dtype=x.dtype, device=x.device)
        dist.all_gather_into_tensor(out, x.contiguous())
        return out

    def spawnprocesses(self, fn, nranks, master_port=12354, seeds_torch=None, *args, **kwargs):
        """Run fn(model, *args, **kwargs) in `nranks` processes, one per GPU
        (_core.py:51-85).  The children train copies of the model; snapshots are the
        channel back to the parent."""
        seeds_torch = prepare_seeds(nranks, seeds_torch)
        worker = DistributedFunc(fn)
        try:
            torch.multiprocessing.spawn(partial(worker, **kwargs),
                                        args=(nranks, master_port, seeds_torch, self._model) + tuple(args),
                                        nprocs=nranks, join=True)
        except ProcessException:
            warnings.warn("Distributed run could not be spawned; if the master port is in use, "
                          "pass another one via master_port.")
            raise

    def attach_from_env(self):
        """Join the group described by RANK / WORLD_SIZE / LOCAL_RANK (torchrun)."""
        rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
        local = int(os.environ.get("LOCAL_RANK", rank))
        if world > 1 and not dist.is_initialized():
            dist.init_process_group(backend=default_backend(), rank=rank, world_size=world)
        device = torch.device('cuda', local) if torch.cuda.is_available() else torch.device('cpu')
        self.ddp_wrapper(rank, world, device=device)


class DistributedFunc:
    """Per-process entry point used by spawnprocesses (_core.py:98-117)."""

    def __init__(self, fn):
        self.fn = fn

    def __call__(self, rank, nranks, master_port, seeds_torch, model, *args, **kwargs):
        setup_process_group(rank, nranks, master_port=master_port)
        try:
            model.device_handler.ddp_wrapper(rank, nranks)
            torch.manual_seed(seeds_torch[rank])
            return self.fn(model, *args, **kwargs)
        finally:
            dist.destroy_process_group()


def default_backend():
    return "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm


def setup_process_group(rank, world_size, master_addr='127.0.0.1', master_port=12354, backend=None):
    os.environ['MASTER_ADDR'] = master_addr
    os.environ['MASTER_PORT'] = str(master_port)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group(backend=backend or default_backend(), rank=rank, world_size=world_size)


def prepare_seeds(nranks, seeds_torch):
    if seeds_torch is None:
        return gen_seed(size=(nranks,))
    assert len(seeds_torch) == nranks, "Numbers of seeds != nranks"
    return seeds_torch


def gen_seed(size=None):
    hi = 2 ** 32 - 1
    draw = torch.randint(hi, size=[1] if size is None else size, device='cpu').tolist()
    return draw[0] if size is None else draw
