"""One training step (forward + KL loss + backward + Adam) of a lattice flow, eager and replayed from a HIP graph
(normflow__amd.GraphedTrainStep), beside an inference pass of the same network."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import normflow__amd as nf
from normflow__amd.prior import NormalPrior
from normflow__amd.action import ScalarPhi4Action
from normflow__amd.fitter import kl_mean
from tools.config_bench import build, DEV

CASES = (("16^3 8 rqs", (16, 16, 16), ['rqs'] * 8, 256), ("32^4 2 rqs", (32,) * 4, ['rqs'] * 2, 4),
         ("16^2 4 affine", (16, 16), ['affine'] * 4, 512), ("16^2 8 rqs", (16, 16), ['rqs'] * 8, 512))
for name, shape, kinds, B in CASES:
    torch.manual_seed(0)
    net = build(shape, kinds)
    prior = NormalPrior(loc=torch.zeros(shape, device=DEV), scale=torch.ones(shape, device=DEV))
    model = nf.Model(net_=net, prior=prior, action=ScalarPhi4Action(kappa=0.67, m_sq=-4 * 0.67, lambd=0.5))
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)

    def step():
        x, logr = prior.sample_(B)
        y, logj = net(x)
        loss = kl_mean(logr - logj, -model.action(y))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        return loss

    def timed(fn, n):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n

    dt = timed(step, 5)
    gstep = nf.GraphedTrainStep(model, kl_mean, B)

    def graphed():
        x, logr = prior.sample_(B)
        loss, _ = gstep(x, logr)
        opt.step()
        return loss

    dg = timed(graphed, 20)
    with torch.no_grad():
        x = prior.sample(B)
        di = timed(lambda: net(x), 5)
    print(f"{name:16s} B={B:4d}  train step {dt * 1e3:8.2f} ms  graphed {dg * 1e3:8.2f} ms   inference {di * 1e3:7.2f} ms   "
          f"ratio {dt / di:5.1f} / {dg / di:5.1f}", flush=True)
