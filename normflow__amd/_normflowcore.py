"""Model / Posterior / Fitter (reference: src/_normflowcore.py).

Host-side orchestration: it calls `net_(x)` / `net_.backward(y)` (the HIP path) and the
prior / action end points, and is API-compatible with the reference: `Model(prior=,
net_=, action=)`, `model.fit(...)`, `model.posterior.sample(n)`, `model.mcmc.sample(n)`,
`model.device_handler.spawnprocesses(fn, nranks)`.
"""
import os
import time

import numpy as np
import torch

from .device import ModelDeviceHandler
from .lib.stats import estimate_logz, fmt_val_err
from .mcmc import MCMCSampler


class Model:
    """prior -> net_ -> action; see README of the reference for the user story."""

    def __init__(self, *, prior, net_, action, name=None):
        self.name = name
        self.net_ = net_
        self.prior = prior
        self.action = action
        self.fit = Fitter(self)
        self.posterior = Posterior(self)
        self.raw_dist = self.posterior
        self.mcmc = MCMCSampler(self)
        self.device_handler = ModelDeviceHandler(self)

    def transform(self, x):
        return self.net_(x)[0]


class Posterior:
    """Raw samples of the trained flow, no accept/reject (_normflowcore.py:70-119)."""

    def __init__(self, model):
        self._model = model

    @torch.no_grad()
    def sample(self, batch_size=1, **kwargs):
        return self.sample_(batch_size=batch_size, **kwargs)[0]

    @torch.no_grad()
    def sample_(self, batch_size=1, preprocess_func=None):
        """-> (y, log q(y)):  x, log r ~ prior;  y, log J = net_(x);  log q = log r - log J."""
        x, logr = self._model.prior.sample_(batch_size)
        if preprocess_func is not None:
            x, logr = preprocess_func(x, logr)
        y, logJ = self._model.net_(x)
        return y, logr - logJ

    @torch.no_grad()
    def sample__(self, batch_size=1, **kwargs):
        """-> (y, log q, log p) with log p = -S(y) (unnormalised)."""
        y, logq = self.sample_(batch_size=batch_size, **kwargs)
        return y, logq, -self._model.action(y)

    @torch.no_grad()
    def log_prob(self, y):
        x, minus_logJ = self._model.net_.backward(y)
        return self._model.prior.log_prob(x) + minus_logJ


class Fitter:
    """Reverse-KL training of `net_` (_normflowcore.py:123-428)."""

    def __init__(self, model):
        self._model = model
        self.train_batch_size = 1
        self.train_history = dict(loss=[], logqp=[], logz=[], ess=[], rho=[], accept_rate=[])
        self.hyperparam = dict(lr=0.001, weight_decay=0.01)
        self.checkpoint_dict = dict(display=False, print_stride=100, print_batch_size=1024,
                                    print_extra_func=None, snapshot_path=None, epochs_run=0)

    def __call__(self, n_epochs=1000, save_every=None, batch_size=64, optimizer_class=torch.optim.AdamW,
                 scheduler=None, loss_fn=None, hyperparam={}, checkpoint_dict={}):
        """Train for `n_epochs` steps of `batch_size` fresh prior samples each."""
        self.hyperparam.update(hyperparam)
        self.checkpoint_dict.update(checkpoint_dict)
        path = self.checkpoint_dict['snapshot_path']
        save_every = n_epochs if save_every is None else save_every
        if path is None:
            print("Not saving model snapshots")
        elif os.path.exists(path):
            print(f"Trying to load snapshot from {path}")
            self._load_snapshot()
        else:
            print("Starting training from scratch")
        self.loss_fn = Fitter.calc_kl_mean if loss_fn is None else loss_fn
        net_ = self._model.net_
        grouped = getattr(net_, '_groups', None) is not None and hasattr(net_, 'grouped_parameters')
        params = net_.grouped_parameters() if grouped else net_.parameters()
        self.optimizer = optimizer_class(params, **self.hyperparam)
        self.scheduler = None if scheduler is None else scheduler(self.optimizer)
        return self.train(n_epochs, batch_size, save_every)

    # ---- snapshots: {"MODEL_STATE": state_dict, "EPOCHS_RUN": n}, rank 0 only
    def _load_snapshot(self):
        path = self.checkpoint_dict['snapshot_path']
        rank = self._model.device_handler.rank
        loc = f"cuda:{rank}" if torch.cuda.is_available() else None
        snap = torch.load(path, map_location=loc, weights_only=True)
        self._model.net_.load_state_dict(snap["MODEL_STATE"])
        self.checkpoint_dict['epochs_run'] = snap['EPOCHS_RUN']
        print(f"Snapshot found: {path}\nResuming training at epoch {snap['EPOCHS_RUN']}")

    def _save_snapshot(self, epoch):
        path = self.checkpoint_dict['snapshot_path']
        done = epoch + self.checkpoint_dict['epochs_run']
        out = path.rsplit('.', 2)[0] + f".E{done}.tar"
        torch.save({"MODEL_STATE": self._model.net_.state_dict(), "EPOCHS_RUN": done}, out)
        print(f"Epoch {done} | Model Snapshot saved at {out}")

    def train(self, n_epochs, batch_size, save_every):
        self.train_batch_size = batch_size
        t0 = time.time()
        loss = None
        for epoch in range(1, n_epochs + 1):
            loss, _ = self.step()
            self.checkpoint(epoch, loss, save_every)
            if self.scheduler is not None:
                self.scheduler.step()
        if n_epochs > 0 and self._model.device_handler.rank == 0:
            print(f"({loss.device}) Time = {time.time() - t0:.3g} sec.")

    def step(self):
        """One optimisation step on a fresh batch (_normflowcore.py:275-294)."""
        m = self._model
        x, logr = m.prior.sample_(self.train_batch_size)
        y, logJ = m.net_(x)
        logq = logr - logJ
        logp = -m.action(y)
        loss = self.loss_fn(logq, logp)
        self.optimizer.zero_grad()
        loss.backward()
        m.device_handler.all_reduce_gradients()     # one flat RCCL all-reduce when nranks > 1
        if torch.isnan(loss):
            print("OOPS: loss is divergent -> no *step* is taken.")
        else:
            self.optimizer.step()
        return loss, logq - logp

    def checkpoint(self, epoch, loss, save_every):
        dh = self._model.device_handler
        cd = self.checkpoint_dict
        if dh.rank == 0:
            self.train_history['loss'].append(loss.item())
            if cd['snapshot_path'] is not None and epoch % save_every == 0:
                self._save_snapshot(epoch)
        if epoch in (1, 10) or epoch % cd['print_stride'] == 0:
            _, logq, logp = self._model.posterior.sample__(cd['print_batch_size'] // dh.nranks)
            logq, logp = dh.all_gather_into_tensor(logq), dh.all_gather_into_tensor(logp)
            if dh.rank == 0:
                self._append_to_train_history(logq, logp)
                self.print_fit_status(epoch, loss=self.loss_fn(logq, logp))

    # ---- losses / diagnostics (all on (B,) tensors)
    @staticmethod
    def calc_kl_mean(logq, logp):
        return (logq - logp).mean()

    @staticmethod
    def calc_kl_var(logq, logp):
        return (logq - logp).var()

    @staticmethod
    def calc_corrcoef(logq, logp):
        return torch.corrcoef(torch.stack([logq, logp]))[0, 1]

    @staticmethod
    def _logz(logq, logp):
        return torch.logsumexp(logp - logq, dim=0) - np.log(logp.shape[0])

    @staticmethod
    def calc_direct_kl_mean(logq, logp):
        w = logp - logq - Fitter._logz(logq, logp)
        return (torch.exp(w) * w).mean()

    @staticmethod
    def calc_kl_mean_includelogz(logq, logp):
        return (logq - logp).mean() + Fitter._logz(logq, logp)

    @staticmethod
    def calc_least_squares(logq, logp):
        return torch.mean((logq - logp + Fitter._logz(logq, logp)) ** 2)

    @staticmethod
    def calc_minus_logz(logq, logp):
        return -Fitter._logz(logq, logp)

    @staticmethod
    def calc_ess(logq, logp):
        """Normalised effective sample size (sum w)^2 / (n sum w^2), w = p/q."""
        d = logq - logp
        return torch.exp(2 * torch.logsumexp(-d, dim=0) - torch.logsumexp(-2 * d, dim=0)) / len(d)

    def calc_minus_ess(self, logq, logp):
        return -self.calc_ess(logq, logp)

    @torch.no_grad()
    def _append_to_train_history(self, logq, logp):
        d = logq - logp
        h = self.train_history
        h['logz'].append(estimate_logz(d, method='jackknife'))
        h['accept_rate'].append(self._model.mcmc.estimate_accept_rate(d))
        h['ess'].append(self.calc_ess(d, 0))
        h['rho'].append(self.calc_corrcoef(logq, logp))
        h['logqp'].append((d.mean().item(), d.std().item()))

    def print_fit_status(self, epoch, loss=None):
        h = self.train_history
        loss = h['loss'][-1] if loss is None else loss
        qp_mean, qp_std = h['logqp'][-1]
        z_mean, z_std = h['logz'][-1]
        ar_mean, ar_std = h['accept_rate'][-1]
        ess, rho = h['ess'][-1], h['rho'][-1]
        if epoch == 1:
            print(f"\n>>> Training progress ({ess.device}) <<<\n")
            print("Note: log(q/p) is estimated with normalized p; "
                  "mean & error are obtained from samples in a batch\n")
        epoch += self.checkpoint_dict['epochs_run']
        line = (f"Epoch: {epoch} | loss: {loss:g} | ess: {ess:g} | rho: {rho:g}"
                f" | log(z): {fmt_val_err(z_mean, z_std, err_digits=2)}"
                f" | log(q/p): {fmt_val_err(qp_mean + z_mean, qp_std, err_digits=2)}"
                f" | accept_rate: {fmt_val_err(ar_mean, ar_std, err_digits=1)}")
        extra = self.checkpoint_dict['print_extra_func']
        if extra is not None:
            line += extra(epoch)
        print(line)


@torch.no_grad()
def backward_sanitychecker(model, n_samples=5, net_=None, return_details=False):
    """forward then backward must give back the input and cancel the log-Jacobian."""
    net_ = model.net_ if net_ is None else net_
    x = model.prior.sample(n_samples)
    y, logJ = net_(x)
    x_hat, log0_hat = net_.backward(y, log0=logJ)
    print("Sanity check is OK if following numbers are zero up to round off:")
    print(f"{torch.sum(torch.abs(x - x_hat)).item():g}", f"{torch.sum(torch.abs(log0_hat)).item():g}")
    if return_details:
        return (x, y, x_hat), (logJ, log0_hat)
