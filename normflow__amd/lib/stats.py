"""Host-side diagnostics: resampling, log Z estimate, value(error) formatting
(reference: src/lib/stats/resampler.py, src/lib/combo/combo.py).  O(batch) work."""
import math

import numpy as np
import torch


class Resampler:
    """Generator of bootstrap / jackknife / shuffling resamples of axis 0."""

    def __init__(self, method='bootstrap'):
        if method not in ('bootstrap', 'jackknife', 'shuffling'):
            raise ValueError(f"unknown resampling method {method!r}")
        self.method = method

    def __call__(self, samples, n_resamples=100, binsize=1, batch_size=None):
        nbins = samples.shape[0] // binsize
        binned = samples[:nbins * binsize].reshape(nbins, binsize, -1)
        is_t = torch.is_tensor(samples)
        dev = dict(device=samples.device) if is_t else {}
        arange = (lambda n: torch.arange(n, **dev)) if is_t else np.arange
        if self.method == 'jackknife':
            picks = (arange(nbins)[arange(nbins) != i] for i in range(nbins))
            out_len = (nbins - 1) * binsize
        elif self.method == 'bootstrap':
            size = nbins if batch_size is None else batch_size
            draw = (lambda: torch.randint(nbins, (size,), **dev)) if is_t else (lambda: np.random.randint(nbins, size=(size,)))
            picks = (draw() for _ in range(n_resamples))
            out_len = nbins * binsize
        else:
            perm = (lambda: torch.randperm(nbins, **dev)) if is_t else (lambda: np.random.permutation(nbins))
            picks = (perm() for _ in range(n_resamples))
            out_len = nbins * binsize
        for idx in picks:
            yield binned[idx].reshape(out_len, *samples.shape[1:])

    def eval(self, samples, fn=lambda x: np.mean(x), **kwargs):
        vals = [fn(r) for r in self(samples, **kwargs)]
        return np.mean(vals), np.std(vals)


def estimate_logz(logqp, n_resamples=10, method='bootstrap'):
    """log Z ~ log mean exp(-logqp), with a resampling error (combo.py:11-23)."""
    n = logqp.shape[0]
    logz = lambda t: torch.logsumexp(t, dim=0).item() - math.log(t.shape[0])
    mean = torch.logsumexp(-logqp, dim=0).item() - math.log(n)
    std = np.std([logz(r) for r in Resampler(method)(-logqp, n_resamples)])
    return mean, std


def fmt_val_err(value, error, err_digits=1):
    """1.1124(2)-style formatting (combo.py:26-34)."""
    try:
        digits = max(0, -int(math.floor(math.log10(error))) + err_digits - 1)
        return "{0:.{2}f}({1:.0f})".format(value, error * 10 ** digits, digits)
    except (ValueError, OverflowError, TypeError):
        return f"{value}+-{error}"
