"""Three eager training steps of BASELINE config 3's network (16^3, 8 RQ-spline layers, batch 256) -- run under rocprofv3 --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")
import torch
import normflow__amd as nf
from normflow__amd.prior import NormalPrior
from normflow__amd.action import ScalarPhi4Action
from normflow__amd.fitter import kl_mean
from tools.config_bench import build, DEV
shape, B = (16, 16, 16), 256
net = build(shape, ['rqs'] * 8)
prior = NormalPrior(loc=torch.zeros(shape, device=DEV), scale=torch.ones(shape, device=DEV))
model = nf.Model(net_=net, prior=prior, action=ScalarPhi4Action(kappa=0.67, m_sq=-4 * 0.67, lambd=0.5))
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
for _ in range(4):
    x, logr = prior.sample_(B)
    y, logj = net(x)
    loss = kl_mean(logr - logj, -model.action(y))
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("done", float(loss))
