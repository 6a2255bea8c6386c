"""One KL training step of a 2-layer 32^4 spline network at batch 16 (forward + loss + backward + Adam), timed, with an
inference pass of the same network beside it -- the workload behind DESIGN section 6's training numbers.  Under
`rocprofv3 --kernel-trace --stats` it gives the per-kernel breakdown kept in profiles/r02_train_kernel_stats.csv."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import normflow__amd as nf
from normflow__amd.prior import NormalPrior
from normflow__amd.action import ScalarPhi4Action
from tools.config_bench import build, DEV

shape, B = (32,) * 4, 16
net = build(shape, ['rqs'] * 2)
prior = NormalPrior(loc=torch.zeros(shape, device=DEV), scale=torch.ones(shape, device=DEV))
model = nf.Model(net_=net, prior=prior, action=ScalarPhi4Action(kappa=0.67, m_sq=-4 * 0.67, lambd=0.5))
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def step():
    x = prior.sample(B)
    logr = prior.log_prob(x)
    y, logj = net(x)
    loss = (logr - logj + model.action(y)).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(2):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 2
with torch.no_grad():
    x = prior.sample(B)
    net(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    net(x)
    torch.cuda.synchronize()
di = time.perf_counter() - t0
print(f"32^4, 2 spline layers, B={B}: train step {dt * 1e3:.1f} ms, inference {di * 1e3:.1f} ms, ratio {dt / di:.1f}")
