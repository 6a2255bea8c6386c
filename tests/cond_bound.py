"""Per-site conditioned float32 error bound of an RQ-spline coupling (test infrastructure).

north_star asks for 1e-5 relative agreement of y and log|J| "on identical inputs".  A float32 evaluation cannot meet a flat
1e-5 on every input: the map's value and log-derivative at a site are functions of the site's knots, and float32 knows the
input and the knots only to eps * |x| and eps * range.  A narrow, steep bin amplifies that by range / bin-width times the
bin's d(log g)/d(theta) (the reference's own float32 run shows it: tests/golden/ref_fp32.npz).  This module turns that
into a bound *per site*, computed in float64 from the oracle's knots, which both the HIP kernel and the reference's float32
outputs are held to with the SAME constant:

    |f32(f) - f|_site  <=  C * bound_site,      bound = sum_q |df/dq| * delta_q  +  roundoff(f)

  q        : x (the site's input), x0, x1, y0, y1 (the bin's knots), d0, d1 (its knot derivatives)
  delta_q  : what float32 cannot know about q:  eps |x| ; eps * max|xlim| for a knot abscissa (a cumsum of softmax weights
             scaled to the range: couplings_.py:227-235) ; eps * max|ylim| for an ordinate ; 2 eps (1 + d) for a softplus
  df/dq    : by autograd through the segment formula (spline.py:185-220) in float64
  roundoff : eps * (|y0| + |y| + (4 + 2A)|y - y0|) for the value and eps * (4 + 4A) for log g, A = the cancellation factor
             of the denominator s + (d0 + d1 - 2s) theta (1 - theta)
  ties     : an input within 4 eps * range of a knot may be evaluated by either neighbouring bin (the rounded knot decides;
             value and derivative are continuous there, d(log g)/dx is not): the bound is the largest over the bins that
             x - D, x, x + D select.

eps = 2^-23.  log|J| of a sample is a sum over its sites: its bound is the sum of the sites' bounds.
"""
import numpy as np
import torch

from oracle import nf_oracle as O

EPS32 = 2.0 ** -23


def _segment_eval(x, x0, x1, y0, y1, d0, d1):
    s = (y1 - y0) / (x1 - x0)
    curv = d1 + d0 - 2 * s
    th = (x - x0) / (x1 - x0)
    t1 = th * (1 - th)
    den = s + curv * t1
    val = y0 + (y1 - y0) * th * (s * th + d0 * (1 - th)) / den
    g = s * s * (d0 + 2 * (s - d0) * th + curv * th * th) / (den * den)
    return val, torch.log(g), (s, th, t1, den)


def rqs_forward_bound(x, out, **kw):
    """x: (N,) float64 inputs of N active sites, out: (N, C) float64 raw logits.  Returns float64 numpy arrays
    (y, logg, bound_y, bound_logg) of the float64 oracle's per-site values and the per-site float32 bounds."""
    with torch.device("cpu"):          # the package makes the GPU torch's default device; this is CPU arithmetic
        return _forward_bound(x, out, **kw)


def _forward_bound(x, out, *, xlim, ylim, extrap=None, knots_x=None, knots_y=None, eps=EPS32):
    x = torch.as_tensor(np.asarray(x), dtype=torch.float64, device="cpu")
    out = torch.as_tensor(np.asarray(out), dtype=torch.float64, device="cpu")
    knots_x = knots_x.cpu() if torch.is_tensor(knots_x) else knots_x
    knots_y = knots_y.cpu() if torch.is_tensor(knots_y) else knots_y
    N = x.shape[0]
    o = out.t().unsqueeze(0)                                   # (1, C, N): channel axis 1, as the oracle expects
    kx, ky, kd = O.knots_from_logits(o, xlim, ylim, knots_x, knots_y)
    kx, ky, kd = (O._bcast_like(k, o) for k in (kx, ky, kd))
    full = (1, kd.shape[1], N)
    kx, ky, kd = (k.expand(full) for k in (kx, ky, kd))
    kx, ky, kd = O.augment_knots(kx, ky, kd, axis=1, **(extrap or {}))
    kx, ky, kd = (k[0].t().contiguous() for k in (kx, ky, kd))  # (N, K)
    K = kx.shape[1]
    rx = max(abs(float(xlim[0])), abs(float(xlim[1])))
    ry = max(abs(float(ylim[0])), abs(float(ylim[1])))
    tie = 4 * eps * max(rx, 1.0)
    best = None
    for shift in (0.0, -tie, tie):
        seg = (kx[:, 1:K - 1] < (x + shift).unsqueeze(1)).sum(dim=1, keepdim=True)
        take = lambda t, off: torch.gather(t, 1, seg + off).squeeze(1)
        q = [x, take(kx, 0), take(kx, 1), take(ky, 0), take(ky, 1), take(kd, 0), take(kd, 1)]
        q = [t.detach().clone().requires_grad_(True) for t in q]
        val, lg, (s, th, t1, den) = _segment_eval(*q)
        gy = torch.autograd.grad(val.sum(), q, retain_graph=True)
        gl = torch.autograd.grad(lg.sum(), q)
        xq, x0, x1, y0, y1, d0, d1 = [t.detach() for t in q]
        dq = [eps * xq.abs(), eps * rx * torch.ones_like(xq), eps * rx * torch.ones_like(xq),
              eps * ry * torch.ones_like(xq), eps * ry * torch.ones_like(xq),
              2 * eps * (1 + d0), 2 * eps * (1 + d1)]
        s, t1, den = s.detach(), t1.detach().abs(), den.detach()
        A = (s.abs() + (d0 + d1 + 2 * s.abs()) * t1) / den.abs()
        vd, ld = val.detach(), lg.detach()
        by = sum(g.abs() * d for g, d in zip(gy, dq)) + eps * (y0.abs() + vd.abs() + (4 + 2 * A) * (vd - y0).abs())
        bl = sum(g.abs() * d for g, d in zip(gl, dq)) + eps * (4 + 4 * A)
        if best is None:
            best = [vd, ld, by, bl]
        else:
            best[2], best[3] = torch.maximum(best[2], by), torch.maximum(best[3], bl)
    return tuple(t.numpy() for t in best)


def active_sites(z, tag):
    """(x (B, n), out (B, n, C), index of the active sites) of an atoms.npz case, float64."""
    shape, parity = tuple(int(v) for v in z[f"{tag}/shape"]), int(z[f"{tag}/parity"])
    am = O.channel_mask(shape, parity).numpy().reshape(-1).astype(bool)
    x, out = np.asarray(z[f"{tag}/x_active"]), np.asarray(z[f"{tag}/out"])
    B, C = x.shape[0], out.shape[1]
    return x.reshape(B, -1)[:, am], np.moveaxis(out.reshape(B, C, -1)[:, :, am], 1, 2), am


def _segment_invert(w, x0, x1, y0, y1, d0, d1, r):
    """The stable root of the oracle (nf_oracle.rqs_invert) with theta scaled by (1 + r): r carries the roundoff of the
    root itself into the autograd sensitivities."""
    s = (y1 - y0) / (x1 - x0)
    curv = d1 + d0 - 2 * s
    eta = (w - y0) / (y1 - y0)
    a2 = -curv * eta + d0 - s
    bb = a2 + s
    a0 = s * eta
    disc2 = bb * bb - 4 * a0 * a2
    disc = torch.sqrt(torch.clamp(disc2, min=0))
    safe = lambda t: torch.where(t == 0, torch.ones_like(t), t)
    th = torch.where(bb >= 0, 2 * a0 / safe(bb + disc), (bb - disc) / safe(2 * a2)) * (1 + r)
    t1 = th * (1 - th)
    den = s + curv * t1
    g = s * s * (d0 + 2 * (s - d0) * th + curv * th * th) / (den * den)
    amp = (bb * bb + 4 * (a0 * a2).abs()) / torch.clamp(disc2.abs(), min=1e-300)
    return x0 + (x1 - x0) * th, -torch.log(g), (s, t1, den, amp)


def rqs_inverse_bound(w, out, **kw):
    """The same statement for the inverse map x = f^-1(w), -log g: w (N,) float64, out (N, C).  Returns
    (x, -logg, bound_x, bound_logg)."""
    with torch.device("cpu"):
        return _inverse_bound(w, out, **kw)


def _inverse_bound(w, out, *, xlim, ylim, extrap=None, knots_x=None, knots_y=None, eps=EPS32):
    w = torch.as_tensor(np.asarray(w), dtype=torch.float64, device="cpu")
    out = torch.as_tensor(np.asarray(out), dtype=torch.float64, device="cpu")
    knots_x = knots_x.cpu() if torch.is_tensor(knots_x) else knots_x
    knots_y = knots_y.cpu() if torch.is_tensor(knots_y) else knots_y
    N = w.shape[0]
    o = out.t().unsqueeze(0)
    kx, ky, kd = O.knots_from_logits(o, xlim, ylim, knots_x, knots_y)
    kx, ky, kd = (O._bcast_like(k, o) for k in (kx, ky, kd))
    full = (1, kd.shape[1], N)
    kx, ky, kd = (k.expand(full) for k in (kx, ky, kd))
    kx, ky, kd = O.augment_knots(kx, ky, kd, axis=1, **(extrap or {}))
    kx, ky, kd = (k[0].t().contiguous() for k in (kx, ky, kd))
    K = kx.shape[1]
    rx = max(abs(float(xlim[0])), abs(float(xlim[1])))
    ry = max(abs(float(ylim[0])), abs(float(ylim[1])))
    tie = 4 * eps * max(ry, 1.0)
    best = None
    for shift in (0.0, -tie, tie):
        seg = (ky[:, 1:K - 1] < (w + shift).unsqueeze(1)).sum(dim=1, keepdim=True)
        take = lambda t, off: torch.gather(t, 1, seg + off).squeeze(1)
        q = [w, take(kx, 0), take(kx, 1), take(ky, 0), take(ky, 1), take(kd, 0), take(kd, 1), torch.zeros_like(w)]
        q = [t.detach().clone().requires_grad_(True) for t in q]
        val, lg, (s, t1, den, amp) = _segment_invert(*q)
        gv = torch.autograd.grad(val.sum(), q, retain_graph=True)
        gl = torch.autograd.grad(lg.sum(), q)
        wq, x0, x1, y0, y1, d0, d1, _ = [t.detach() for t in q]
        one = torch.ones_like(wq)
        dq = [eps * wq.abs(), eps * rx * one, eps * rx * one, eps * ry * one, eps * ry * one,
              2 * eps * (1 + d0), 2 * eps * (1 + d1), eps * (4 + amp.detach())]
        s, t1, den = s.detach(), t1.detach().abs(), den.detach()
        A = (s.abs() + (d0 + d1 + 2 * s.abs()) * t1) / den.abs()
        vd, ld = val.detach(), lg.detach()
        bv = sum(g.abs() * d for g, d in zip(gv, dq)) + eps * (x0.abs() + vd.abs() + 2 * (vd - x0).abs())
        bl = sum(g.abs() * d for g, d in zip(gl, dq)) + eps * (4 + 4 * A)
        if best is None:
            best = [vd, ld, bv, bl]
        else:
            best[2], best[3] = torch.maximum(best[2], bv), torch.maximum(best[3], bl)
    return tuple(t.numpy() for t in best)


C_SITE = 3.0      # the one constant: every float32 evaluation (HIP kernel, reference, oracle in float32) is held to C * bound


def case_bounds(z, tag, opts, inverse=False):
    """Bounds for a whole atoms.npz case.  Returns dict(val, logd, b_val, b_logd: (B, S, n) float64 arrays over the n
    active sites of the S splines (S = 1 but for multirqs), am: the active-site index, C: logits per spline)."""
    kind = tag.split("/")[0]
    shape, parity = tuple(int(v) for v in z[f"{tag}/shape"]), int(z[f"{tag}/parity"])
    am = O.channel_mask(shape, parity).numpy().reshape(-1).astype(bool)
    inp = np.asarray(z[f"{tag}/y" if inverse else f"{tag}/x_active"])
    out = np.asarray(z[f"{tag}/out"])
    B = inp.shape[0]
    S = 2 if kind == "multirqs" else 1
    inp = inp.reshape(B, S, -1)[:, :, am]
    Ctot = out.shape[1]
    Cs = Ctot // S
    out = out.reshape(B, S, Cs, -1)[:, :, :, am]                       # (B, S, C, n)
    n = inp.shape[2]
    res = {k: np.zeros((B, S, n)) for k in ("val", "logd", "b_val", "b_logd")}
    fn = rqs_inverse_bound if inverse else rqs_forward_bound
    for s in range(S):
        if kind == "multirqs":
            o = dict(xlim=opts["xlims"][s], ylim=opts["ylims"][s], extrap=opts["extraps"][s])
        else:
            o = dict(opts)
        v, l, bv, bl = fn(inp[:, s].reshape(-1), np.moveaxis(out[:, s], 1, 2).reshape(-1, Cs), **o)
        for k, a in zip(("val", "logd", "b_val", "b_logd"), (v, l, bv, bl)):
            res[k][:, s] = a.reshape(B, n)
    res["am"], res["C"] = am, Cs
    return res
