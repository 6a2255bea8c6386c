from .stats import Resampler, estimate_logz, fmt_val_err
