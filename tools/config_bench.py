"""Throughput of the BASELINE.json configurations other than the bench's (parity-test cases; one GPU, fp32,
forward + log|det J| under no_grad, synthetic inputs).  python tools/config_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
from normflow__amd.nn import ConvAct, RQSplineCoupling_, AffineCoupling_, ModuleList_
from normflow__amd.mask import EvenOddMask

DEV = torch.device("cuda:0")


def build(shape, kinds, m=16, dtype=torch.float32):
    d = len(shape)
    mask = EvenOddMask(shape=shape)
    lim = dict(xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
    blocks = []
    for kind in kinds:
        C = 3 * m - 2 if kind == 'rqs' else 2
        net = ConvAct(1, C, 3, conv_dim=d, hidden_sizes=[8, 8], acts=['tanh', 'tanh', None])
        with torch.no_grad():
            for p in list(net.parameters())[-2:]:
                p.mul_(0.3)
        blocks.append(RQSplineCoupling_([net], mask=mask, **lim) if kind == 'rqs' else AffineCoupling_([net], mask=mask))
    net = ModuleList_(blocks)
    net.to(device=DEV, dtype=dtype)
    return net


def run(name, shape, kinds, B, reps=3, dtype=torch.float32, m=16, hidden=8):
    """dtype = torch.float16: BASELINE config 5's storage (half parameters and field, fp32 arithmetic and log-det)."""
    torch.manual_seed(0)
    net = build(shape, kinds, m=m, dtype=dtype)
    x = torch.randn((B,) + shape, device=DEV, dtype=torch.float32).to(dtype)
    with torch.no_grad():
        net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net(x)
        torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    sites = B
    for n in shape:
        sites *= n
    print(f"{name:34s} B={B:5d}  {dt * 1e3:9.2f} ms/step  {B / dt:10.1f} configs/s  {sites * len(kinds) / dt / 1e9:7.2f} Gsite-layers/s")


def run_graphed(name, shape, kinds, B, reps=50):
    """The same pass replayed from a HIP graph (normflow__amd.GraphedFlow): what is left when the host's launch path is gone."""
    from normflow__amd import GraphedFlow
    torch.manual_seed(0)
    net = build(shape, kinds)
    x = torch.randn((B,) + shape, device=DEV, dtype=torch.float32)
    g = GraphedFlow(net, x)
    g(x, clone=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        g(x, clone=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    sites = B
    for n in shape:
        sites *= n
    print(f"{name:34s} B={B:5d}  {dt * 1e3:9.2f} ms/step  {B / dt:10.1f} configs/s  {sites * len(kinds) / dt / 1e9:7.2f} Gsite-layers/s")


if __name__ == "__main__":
    run("c2 16x16, 4 affine", (16, 16), ['affine'] * 4, 512, reps=20)
    run_graphed("c2 16x16, 4 affine (HIP graph)", (16, 16), ['affine'] * 4, 512)
    run("c3 16^3, 8 rqs m=16", (16, 16, 16), ['rqs'] * 8, 1024)
    run("c4 32^4, 8 rqs (per-GPU share)", (32,) * 4, ['rqs'] * 8, 128)
    run("c5 48^4, 8 affine + 8 rqs", (48,) * 4, ['affine', 'rqs'] * 8, 8)
    run("c5 48^4, 8+8, fp16 storage", (48,) * 4, ['affine', 'rqs'] * 8, 8, dtype=torch.float16)
    run("c5 48^4, 8+8, fp16 storage", (48,) * 4, ['affine', 'rqs'] * 8, 32, dtype=torch.float16)
    run("   32^4, 8 affine + 8 rqs", (32,) * 4, ['affine', 'rqs'] * 8, 40)
    run("   48^4, 8 rqs", (48,) * 4, ['rqs'] * 8, 16)
    run("   32^4, 8 rqs (same sites: B=81)", (32,) * 4, ['rqs'] * 8, 81)
    run("   64^4, 8 rqs", (64,) * 4, ['rqs'] * 8, 8)
