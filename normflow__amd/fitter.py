"""`Fitter`: reverse-KL training of the flow (`model.fit(...)`).

API and defaults of the reference's Fitter (src/_normflowcore.py:123-428): AdamW, lr 1e-3, weight
decay 1e-2, fresh prior batch per step, progress line at epochs 1, 10 and every `print_stride`,
snapshots `{"MODEL_STATE", "EPOCHS_RUN"}`.  Under data parallelism the gradients are averaged with
one flat all-reduce (device/_core.py) between backward() and the optimizer step.
"""
import os
import time

import numpy as np
import torch

from .lib.stats import estimate_logz, fmt_val_err


# ---- objective functions and diagnostics on (log q, log p) sample vectors -------------------
def _log_mean_weight(logq, logp):
    """log of the mean importance weight p/q, i.e. the log Z estimate."""
    return torch.logsumexp(logp - logq, dim=0) - np.log(logp.shape[0])


def kl_mean(logq, logp):
    return (logq - logp).mean()


def kl_var(logq, logp):
    return (logq - logp).var()


def corrcoef(logq, logp):
    return torch.corrcoef(torch.stack([logq, logp]))[0, 1]


def direct_kl_mean(logq, logp):
    w = logp - logq - _log_mean_weight(logq, logp)
    return (w.exp() * w).mean()


def kl_mean_includelogz(logq, logp):
    return kl_mean(logq, logp) + _log_mean_weight(logq, logp)


def least_squares(logq, logp):
    return ((logq - logp + _log_mean_weight(logq, logp)) ** 2).mean()


def minus_logz(logq, logp):
    return -_log_mean_weight(logq, logp)


def ess(logq, logp):
    """(sum w)^2 / (n sum w^2) for w = p/q."""
    d = logq - logp
    return torch.exp(2 * torch.logsumexp(-d, dim=0) - torch.logsumexp(-2 * d, dim=0)) / len(d)


class Fitter:

    # the reference exposes these as Fitter.calc_*
    calc_kl_mean = staticmethod(kl_mean)
    calc_kl_var = staticmethod(kl_var)
    calc_corrcoef = staticmethod(corrcoef)
    calc_direct_kl_mean = staticmethod(direct_kl_mean)
    calc_kl_mean_includelogz = staticmethod(kl_mean_includelogz)
    calc_least_squares = staticmethod(least_squares)
    calc_minus_logz = staticmethod(minus_logz)
    calc_ess = staticmethod(ess)

    def calc_minus_ess(self, logq, logp):
        return -ess(logq, logp)

    def __init__(self, model):
        self._model = model
        self.train_batch_size = 1
        self.train_history = {k: [] for k in ('loss', 'logqp', 'logz', 'ess', 'rho', 'accept_rate')}
        self.hyperparam = dict(lr=0.001, weight_decay=0.01)
        self.checkpoint_dict = dict(display=False, print_stride=100, print_batch_size=1024,
                                    print_extra_func=None, snapshot_path=None, epochs_run=0)
        # MI355X-side option (no counterpart in the reference): forward + loss + backward of a step replayed from ONE HIP graph
        # (graphs.GraphedTrainStep) -- for the launch-bound small lattices; same numbers as the eager step
        self.graphed = False
        self._graph_step = None

    def __call__(self, n_epochs=1000, save_every=None, batch_size=64, optimizer_class=torch.optim.AdamW,
                 scheduler=None, loss_fn=None, hyperparam={}, checkpoint_dict={}, graphed=None):
        if graphed is not None:
            self.graphed = bool(graphed)
        self.hyperparam.update(hyperparam)
        self.checkpoint_dict.update(checkpoint_dict)
        self._maybe_resume()
        self.loss_fn = loss_fn or kl_mean
        net_ = self._model.net_
        use_groups = getattr(net_, '_groups', None) is not None and hasattr(net_, 'grouped_parameters')
        self.optimizer = optimizer_class(net_.grouped_parameters() if use_groups else net_.parameters(),
                                         **self.hyperparam)
        self.scheduler = scheduler(self.optimizer) if scheduler is not None else None
        return self.train(n_epochs, batch_size, n_epochs if save_every is None else save_every)

    # ---- snapshots
    def _maybe_resume(self):
        path = self.checkpoint_dict['snapshot_path']
        if path is None:
            print("Not saving model snapshots")
        elif not os.path.exists(path):
            print("Starting training from scratch")
        else:
            print(f"Trying to load snapshot from {path}")
            self._load_snapshot()

    def _load_snapshot(self):
        path = self.checkpoint_dict['snapshot_path']
        where = f"cuda:{self._model.device_handler.rank}" if torch.cuda.is_available() else None
        snap = torch.load(path, map_location=where, weights_only=True)
        self._model.net_.load_state_dict(snap["MODEL_STATE"])
        self.checkpoint_dict['epochs_run'] = snap['EPOCHS_RUN']
        print(f"Snapshot found: {path}\nResuming training at epoch {snap['EPOCHS_RUN']}")

    def _save_snapshot(self, epoch):
        total = epoch + self.checkpoint_dict['epochs_run']
        stem = self.checkpoint_dict['snapshot_path'].rsplit('.', 2)[0]
        out = f"{stem}.E{total}.tar"
        torch.save({"MODEL_STATE": self._model.net_.state_dict(), "EPOCHS_RUN": total}, out)
        print(f"Epoch {total} | Model Snapshot saved at {out}")

    # ---- the loop
    def train(self, n_epochs, batch_size, save_every):
        self.train_batch_size = batch_size
        started, loss = time.time(), None
        for epoch in range(1, n_epochs + 1):
            loss, _ = self.step()
            self.checkpoint(epoch, loss, save_every)
            if self.scheduler is not None:
                self.scheduler.step()
        if loss is not None and self._model.device_handler.rank == 0:
            print(f"({loss.device}) Time = {time.time() - started:.3g} sec.")

    def step(self):
        """Draw, push through the flow, evaluate the loss, backpropagate, update."""
        model = self._model
        x, logr = model.prior.sample_(self.train_batch_size)
        if self.graphed and x.is_cuda:
            loss, logqp = self._graphed_step(x, logr)
            logq, logp = logqp, 0
        else:
            y, logJ = model.net_(x)
            logq, logp = logr - logJ, -model.action(y)
            loss = self.loss_fn(logq, logp)
            self.optimizer.zero_grad()
            loss.backward()
        model.device_handler.all_reduce_gradients()
        if bool(torch.isnan(loss)):
            print("OOPS: loss is divergent -> no *step* is taken.")
        else:
            self.optimizer.step()
        return loss, logq - logp

    def _graphed_step(self, x, logr):
        key = (self.train_batch_size, self.loss_fn, tuple(x.shape), x.dtype)
        if self._graph_step is None or self._graph_step[0] != key:
            from .graphs import GraphedTrainStep
            self._graph_step = (key, GraphedTrainStep(self._model, self.loss_fn, self.train_batch_size))
        return self._graph_step[1](x, logr)

    def checkpoint(self, epoch, loss, save_every):
        handler, opts = self._model.device_handler, self.checkpoint_dict
        if handler.rank == 0:
            self.train_history['loss'].append(loss.item())
            if opts['snapshot_path'] is not None and epoch % save_every == 0:
                self._save_snapshot(epoch)
        if epoch not in (1, 10) and epoch % opts['print_stride']:
            return
        _, logq, logp = self._model.posterior.sample__(opts['print_batch_size'] // handler.nranks)
        logq, logp = handler.all_gather_into_tensor(logq), handler.all_gather_into_tensor(logp)
        if handler.rank == 0:
            self._append_to_train_history(logq, logp)
            self.print_fit_status(epoch, loss=self.loss_fn(logq, logp))

    @torch.no_grad()
    def _append_to_train_history(self, logq, logp):
        d = logq - logp
        record = dict(logz=estimate_logz(d, method='jackknife'),
                      accept_rate=self._model.mcmc.estimate_accept_rate(d),
                      ess=ess(d, 0), rho=corrcoef(logq, logp), logqp=(d.mean().item(), d.std().item()))
        for key, val in record.items():
            self.train_history[key].append(val)

    def print_fit_status(self, epoch, loss=None):
        last = {k: v[-1] for k, v in self.train_history.items() if v}
        loss = last['loss'] if loss is None else loss
        (z, dz), (qp, dqp), (ar, dar) = last['logz'], last['logqp'], last['accept_rate']
        if epoch == 1:
            print(f"\n>>> Training progress ({last['ess'].device}) <<<\n")
            print("Note: log(q/p) is estimated with normalized p; "
                  "mean & error are obtained from samples in a batch\n")
        epoch += self.checkpoint_dict['epochs_run']
        fields = [f"Epoch: {epoch}", f"loss: {loss:g}", f"ess: {last['ess']:g}", f"rho: {last['rho']:g}",
                  f"log(z): {fmt_val_err(z, dz, err_digits=2)}",
                  f"log(q/p): {fmt_val_err(qp + z, dqp, err_digits=2)}",
                  f"accept_rate: {fmt_val_err(ar, dar, err_digits=1)}"]
        line = " | ".join(fields)
        extra = self.checkpoint_dict['print_extra_func']
        print(line + (extra(epoch) if extra is not None else ""))
