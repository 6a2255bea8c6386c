"""Independence-Metropolis sampling on top of the flow (reference: src/mcmc/mcmc.py).
Host-side and serial by nature (accept/reject is a chain); the proposals come from
`posterior.sample__`, i.e. from the HIP path."""
import copy

import numpy as np
import torch

from ..lib.stats import Resampler, estimate_logz, fmt_val_err

seize = lambda t: t.detach().cpu().numpy()


class Metropolis:
    @staticmethod
    @torch.no_grad()
    def calc_accept_status(logqp, logqp_ref=None):
        """accept[i] = log u_i < logqp_ref - logqp[i], ref updated on accept (mcmc.py:304-317)."""
        logqp = np.asarray(logqp)
        ref = logqp[0] if logqp_ref is None else logqp_ref
        logu = np.log(np.random.rand(logqp.shape[0]))
        status = np.empty(len(logqp), dtype=bool)
        for i, cur in enumerate(logqp):
            status[i] = logu[i] < ref - cur
            if status[i]:
                ref = cur
        return status

    @staticmethod
    def calc_accept_indices(accept_seq):
        """Index of the configuration kept at each chain position (mcmc.py:319-328)."""
        idx = np.arange(len(accept_seq))
        last = 0
        for i, ok in enumerate(accept_seq):
            if ok:
                last = i
            else:
                idx[i] = last
        return idx

    @staticmethod
    def calc_accept_count(accept_seq):
        pos = np.where(accept_seq)[0]
        return pos[1:] - pos[:-1]


class MCMCHistory:
    _KEYS = ('logq', 'logp', 'raw_logq', 'raw_logp', 'accept_seq', 'accept_ind', 'accept_rate')

    def __init__(self):
        self.reset_history()

    def reset_history(self):
        for k in self._KEYS:
            setattr(self, k, [])

    def bookkeeping(self, **items):
        for k, v in items.items():
            if v is None:
                continue
            if k in ('logq', 'logp'):
                v = seize(v)
            elif k in ('raw_logq', 'raw_logp'):
                v = copy.copy(seize(v))
            getattr(self, k).append(v)

    @property
    def logqp(self):
        return [q - p for q, p in zip(self.logq, self.logp)]

    @property
    def raw_logqp(self):
        return [q - p for q, p in zip(self.raw_logq, self.raw_logp)]

    def report_summary(self, since=0, asstr=False):
        fmt = (lambda m, s: fmt_val_err(m, s, err_digits=2)) if asstr else (lambda m, s: (m, s))
        logqp = torch.tensor(self.logq[-1] - self.logp[-1])
        rate = torch.tensor(self.accept_rate)
        ms = lambda t: (t.mean().item(), t.std().item())
        return {'logqp': fmt(*ms(logqp)), 'logz': fmt(*estimate_logz(logqp)), 'accept_rate': fmt(*ms(rate))}


class MCMCSampler:
    """Draw proposals from the flow, keep/repeat them by Metropolis (mcmc.py:15-128)."""

    def __init__(self, model):
        self._model = model
        self.history = MCMCHistory()
        self._ref = dict(sample=None, logq=None, logp=None, logqp=None)

    @torch.no_grad()
    def sample(self, batch_size=1, **kwargs):
        return self.sample__(batch_size=batch_size, **kwargs)[0]

    @torch.no_grad()
    def sample_(self, batch_size=1, **kwargs):
        return self.sample__(batch_size=batch_size, **kwargs)[:2]

    @torch.no_grad()
    def sample__(self, batch_size=1, bookkeeping=False):
        y, logq, logp = self._model.posterior.sample__(batch_size=batch_size)
        if bookkeeping:
            self.history.bookkeeping(raw_logq=logq, raw_logp=logp)
        y, logq, logp = self._accept_reject_step(y, logq, logp, bookkeeping=bookkeeping)
        if bookkeeping:
            self.history.bookkeeping(logq=logq, logp=logp)
        return y, logq, logp

    @torch.no_grad()
    def _accept_reject_step(self, y, logq, logp, bookkeeping=False):
        ref = self._ref
        accept = Metropolis.calc_accept_status(seize(logq - logp), ref['logqp'])
        if not accept[0]:
            y[0], logq[0], logp[0] = ref['sample'], ref['logq'], ref['logp']
        keep = Metropolis.calc_accept_indices(accept)
        keep_t = torch.as_tensor(keep, dtype=torch.long, device=y.device)
        y, logq, logp = (t.index_select(0, keep_t) for t in (y, logq, logp))
        ref.update(sample=y[-1], logq=logq[-1].item(), logp=logp[-1].item())
        ref['logqp'] = ref['logq'] - ref['logp']
        self.history.bookkeeping(accept_rate=np.mean(accept))
        if bookkeeping:
            self.history.bookkeeping(accept_seq=accept, accept_ind=keep)
        return y, logq, logp

    @torch.no_grad()
    def serial_sample_generator(self, n_samples, batch_size=16):
        for i in range(n_samples):
            j = i % batch_size
            if j == 0:
                y, logq, logp = self.sample__(batch_size)
            yield y[j].unsqueeze(0), logq[j].unsqueeze(0), logp[j].unsqueeze(0)

    @torch.no_grad()
    def calc_accept_rate(self, n_samples=1024, batch_size=None, n_resamples=10, method='shuffling'):
        if batch_size is None or batch_size > n_samples:
            batch_size = n_samples
        chunks = []
        for _ in range(int(np.ceil(n_samples / batch_size))):
            _, logq, logp = self._model.posterior.sample__(batch_size=batch_size)
            chunks.append(seize(logq - logp))
        return self.estimate_accept_rate(np.concatenate(chunks))

    @staticmethod
    @torch.no_grad()
    def estimate_accept_rate(logqp, n_resamples=10, method='shuffling'):
        if torch.is_tensor(logqp):
            logqp = seize(logqp)
        rate = lambda q: np.mean(Metropolis.calc_accept_status(q))
        return Resampler(method).eval(logqp, fn=rate, n_resamples=n_resamples)

    def log_prob(self, y, action_logz=0):
        return -self._model.action(y) - action_logz
