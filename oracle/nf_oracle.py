"""CPU ORACLE for the coupling-layer hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file is a plain-PyTorch (CPU) restatement of the algorithm the reference
(jkomijani/normflow_ @ 2024-10-24) runs on its coupling hot path.  It exists only
to *check* the HIP path: it may be imported by ``tests/``, by
``__graft_entry__.smoke()`` and by ``bench.py``'s ``cpu_baseline`` leg, and by
nothing else.  The product package ``normflow__amd`` never imports it and has no
CPU fallback.

Parity status: PINNED.  Every function below is checked (fp64, <=1e-12) against
golden vectors produced by importing the reference itself in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``); see
``tests/test_oracle_golden.py``.

Every function cites the reference lines it restates (paths relative to the
reference repo root).  The restatement is deliberately *not* structured like the
reference (no spline object, no AugmentKnots class, no mask module): it is a set
of pure functions on explicit tensors so that it can be read next to the HIP
kernels' maths.

Conventions: B = batch, L = lattice shape, V = prod(L), m = knots per spline,
C = 3m-2 channels of raw logits per site, channel axis = 1.
"""

import math

import torch
import torch.nn.functional as F

LN2 = math.log(2.0)


# --------------------------------------------------------------------------- masks
def even_odd_mask(shape, parity=0, dtype=torch.uint8):
    """mask[ind] = (1 - parity + sum(ind)) % 2   (src/mask/mask.py:55-60).

    With parity=0 the mask is 1 on the even-sum sublattice: that sublattice is
    "channel 0" of split/purify (src/mask/mask.py:30-37) and is the one updated by
    coupling layers k = 0, 2, ... (src/nn/scalar/couplings_.py:56-57).
    """
    shape = tuple(int(l) for l in shape)
    total = torch.zeros(shape, dtype=torch.int64, device='cpu')
    for mu, l in enumerate(shape):
        view = [1] * len(shape)
        view[mu] = l
        total = total + torch.arange(l, dtype=torch.int64, device='cpu').reshape(view)
    return ((1 - parity + total) % 2).to(dtype)


def channel_mask(shape, channel, parity=0, dtype=torch.float64):
    """The 0/1 field that `purify(., channel)` multiplies with (mask.py:36-37)."""
    m = even_odd_mask(shape, parity=parity).to(dtype)
    return m if channel == 0 else 1 - m


def sum_density(t):
    """Per-sample reduction over every axis but the batch (src/nn/_core.py:38-42)."""
    return t.reshape(t.shape[0], -1).sum(dim=1) if t.dim() > 1 else t


# ----------------------------------------------------------------- knot construction
def softplus_ln2(t):
    """Softplus(beta=ln 2): log2(1 + 2^t); identity once ln2*t > 20 (torch's
    threshold)  (couplings_.py:172, modules.py:315)."""
    return F.softplus(t, beta=LN2, threshold=20.0)


def coords_from_logits(w, lo, width, axis=1):
    """cat(0, cumsum(softmax(w))) * width + lo   (couplings_.py:227-235)."""
    frac = torch.cumsum(torch.softmax(w, dim=axis), dim=axis)
    zero_shape = list(w.shape)
    zero_shape[axis] = 1
    frac = torch.cat((w.new_zeros(zero_shape), frac), dim=axis)
    return frac * width + lo


def knots_from_logits(out, xlim=(0, 1), ylim=(0, 1), knots_x=None, knots_y=None):
    """Split the raw net output into knot tensors (couplings_.py:211-262).

    out: (B, n, *L).  n = 3m-2 (both free), 2m-1 (one of x/y fixed) or m (both
    fixed).  Returns (knots_x, knots_y, knots_d); free ones are (B, m, *L), fixed
    ones are returned as given (1-D of length m).
    """
    n = out.shape[1]
    xw, yw = xlim[1] - xlim[0], ylim[1] - ylim[0]
    if knots_x is None and knots_y is None:
        m = (n + 2) // 3
        wx, wy, wd = out.split((m - 1, m - 1, m), dim=1)
        kx = coords_from_logits(wx, xlim[0], xw)
        ky = coords_from_logits(wy, ylim[0], yw)
    elif knots_x is not None and knots_y is None:
        m = (n + 2) // 2
        wy, wd = out.split((m - 1, m), dim=1)
        kx, ky = knots_x, coords_from_logits(wy, ylim[0], yw)
    elif knots_x is None and knots_y is not None:
        m = (n + 2) // 2
        wx, wd = out.split((m - 1, m), dim=1)
        kx, ky = coords_from_logits(wx, xlim[0], xw), knots_y
    else:
        kx, ky, wd = knots_x, knots_y, out
    return kx, ky, softplus_ln2(wd)


def _bcast_like(k, ref, axis=1):
    """A 1-D knot vector viewed so that it broadcasts along `axis` of `ref`."""
    if k.dim() > 1:
        return k
    view = [1] * ref.dim()
    view[axis] = -1
    return k.reshape(view)


def augment_knots(kx, ky, kd, left=None, right=None, axis=1):
    """Boundary knots for extrapolation (src/lib/spline/spline.py:458-532).

    'linear': one extra knot one unit outside, on the tangent line (:466-478).
    'anti'  : mirror every *other* knot (including a linear one already added on
              the opposite side, because the reference runs the linear step first,
              :448-456) through the end knot; derivatives are reflected unchanged
              (:498-501, :514-517).
    """
    nd = max(kx.dim(), ky.dim(), kd.dim())
    if nd > 1:
        ref = kx if kx.dim() == nd else (ky if ky.dim() == nd else kd)
        kx, ky, kd = (_bcast_like(k, ref, axis).expand_as(ref) if k.dim() == 1 else k
                      for k in (kx, ky, kd))
        ax = axis
    else:
        ax = 0
    first = lambda t: t.narrow(ax, 0, 1)
    last = lambda t: t.narrow(ax, t.shape[ax] - 1, 1)

    if left == 'linear' or right == 'linear':
        xs, ys, ds = [kx], [ky], [kd]
        if left == 'linear':
            xs.insert(0, first(kx) - 1)
            ys.insert(0, first(ky) - first(kd))
            ds.insert(0, first(kd))
        if right == 'linear':
            xs.append(last(kx) + 1)
            ys.append(last(ky) + last(kd))
            ds.append(last(kd))
        kx, ky, kd = (torch.cat(t, dim=ax) for t in (xs, ys, ds))
        if left is None or right is None:
            return kx, ky, kd

    anti = ('anti', 'anti-periodic')
    if left in anti or right in anti:
        n = kx.shape[ax]
        xs, ys, ds = [kx], [ky], [kd]
        if left in anti:
            rest = lambda t: torch.flip(t.narrow(ax, 1, n - 1), [ax])
            xs.insert(0, 2 * first(kx) - rest(kx))
            ys.insert(0, 2 * first(ky) - rest(ky))
            ds.insert(0, rest(kd))
        if right in anti:
            rest = lambda t: torch.flip(t.narrow(ax, 0, n - 1), [ax])
            xs.append(2 * last(kx) - rest(kx))
            ys.append(2 * last(ky) - rest(ky))
            ds.append(rest(kd))
        kx, ky, kd = (torch.cat(t, dim=ax) for t in (xs, ys, ds))
    for side in (left, right):
        if side not in (None, 'linear') + anti:
            raise ValueError(f"oracle: extrapolation {side!r} is out of scope")
    return kx, ky, kd


# -------------------------------------------------------------- spline evaluate/invert
def _segment_index(knots, v, axis):
    """torch.searchsorted(left) followed by clamp(idx, 1, K-1) - 1
    (spline.py:154-172): equals the number of *interior* knots strictly below v.
    A value equal to knot k therefore lands in segment k-1; values outside the
    range reuse the first / last segment."""
    K = knots.shape[axis]
    inner = knots.narrow(axis, 1, K - 2) if K > 2 else knots.narrow(axis, 0, 0)
    return (inner < v).sum(dim=axis, keepdim=True)


def _gather6(kx, ky, kd, seg, axis):
    take = lambda t, off: torch.gather(t, axis, seg + off)
    return take(kx, 0), take(kx, 1), take(ky, 0), take(ky, 1), take(kd, 0), take(kd, 1)


def _slope_terms(x0, x1, y0, y1, d0, d1):
    s = (y1 - y0) / (x1 - x0)
    return s, d1 + d0 - 2 * s


def _grad_at(theta, s, d0, curv):
    """g1(theta)  (spline.py:209-211)."""
    den = s + curv * theta * (1 - theta)
    return s * s * (d0 + 2 * (s - d0) * theta + curv * theta * theta) / (den * den)


def rqs_evaluate(kx, ky, kd, v, axis=1):
    """Rational-quadratic forward map and its derivative at v
    (spline.py:185-220).  Knot tensors carry K along `axis`; v has size 1 there."""
    kx, ky, kd = _expand_knots(kx, ky, kd, v, axis)
    seg = _segment_index(kx, v, axis)
    x0, x1, y0, y1, d0, d1 = _gather6(kx, ky, kd, seg, axis)
    s, curv = _slope_terms(x0, x1, y0, y1, d0, d1)
    th = (v - x0) / (x1 - x0)
    den = s + curv * th * (1 - th)
    val = y0 + (y1 - y0) * th * (s * th + d0 * (1 - th)) / den
    return val, _grad_at(th, s, d0, curv)


def rqs_invert(kx, ky, kd, w, axis=1):
    """Inverse map x(w) and dx/dw = 1/g1 (spline.py:222-287).

    The reference solves a2 th^2 + a1 th + a0 = 0 with th = (-a1 - delta)/(2 a2),
    which cancels catastrophically as a2 -> 0 (linear tails).
    The oracle uses the algebraically identical, stable forms th = 2 a0/(-a1+delta) for
    -a1 >= 0 and th = (-a1-delta)/(2 a2) otherwise (neither cancels in its branch);
    inside the knot range the two agree to <=1e-12 in fp64 (pinned by the goldens,
    which only compare there), and the stable form is additionally pinned by the
    forward round trip.
    """
    kx, ky, kd = _expand_knots(kx, ky, kd, w, axis)
    seg = _segment_index(ky, w, axis)
    x0, x1, y0, y1, d0, d1 = _gather6(kx, ky, kd, seg, axis)
    s, curv = _slope_terms(x0, x1, y0, y1, d0, d1)
    eta = (w - y0) / (y1 - y0)
    a2 = -curv * eta + d0 - s
    a1 = -a2 - s
    a0 = s * eta
    delta = torch.sqrt(a1 * a1 - 4 * a0 * a2)
    bb = -a1
    safe = lambda t: torch.where(t == 0, torch.ones_like(t), t)
    th = torch.where(bb >= 0, 2 * a0 / safe(bb + delta), (bb - delta) / safe(2 * a2))
    return x0 + (x1 - x0) * th, 1 / _grad_at(th, s, d0, curv)


def _expand_knots(kx, ky, kd, v, axis):
    """Broadcast the three knot tensors against each other and against the value
    tensor on every axis but `axis` (so that torch.gather can index them)."""
    K = max(k.shape[axis] for k in (kx, ky, kd))
    shp = list(torch.broadcast_shapes(*[tuple(1 if i == axis else n for i, n in enumerate(t.shape))
                                        for t in (kx, ky, kd, v)]))
    shp[axis] = K
    return tuple(k.expand(shp) for k in (kx, ky, kd))


# ---------------------------------------------------------------- coupling "atoms"
def rqs_coupling_atom(x_active, out, active_mask, *, xlim=(0, 1), ylim=(0, 1),
                      extrap=None, knots_x=None, knots_y=None, inverse=False, log0=0):
    """One RQ-spline coupling layer given the net output `out`
    (couplings_.py:178-200).  x_active: (B,*L) with zeros off the active
    sublattice; active_mask: (*L) 0/1.  Returns (fx_active, log0 + sum log g)."""
    extrap = extrap or {}
    kx, ky, kd = knots_from_logits(out, xlim, ylim, knots_x, knots_y)
    ref = out
    kx, ky, kd = (_bcast_like(k, ref) for k in (kx, ky, kd))
    full = (out.shape[0], kd.shape[1]) + tuple(out.shape[2:])      # fixed 1-D knots broadcast to every site
    kx, ky, kd = (k.expand(full) for k in (kx, ky, kd))
    kx, ky, kd = augment_knots(kx, ky, kd, axis=1, **extrap)
    v = x_active.unsqueeze(1)
    f = rqs_invert if inverse else rqs_evaluate
    val, g = f(kx, ky, kd, v, axis=1)
    val, g = val.squeeze(1), g.squeeze(1)
    return val * active_mask, log0 + sum_density(torch.log(g) * active_mask)


def multi_rqs_coupling_atom(x_active, out, active_mask, *, xlims, ylims, extraps,
                            inverse=False, log0=0):
    """`num_splines` independent splines, one per data channel
    (couplings_.py:304-329, 342-412).  x_active: (B, n_s, *L); out: (B, n_s*C, *L)."""
    ns = len(xlims)
    outs = torch.tensor_split(out, ns, dim=1)
    xs = torch.tensor_split(x_active, ns, dim=1)
    vals, logs = [], 0
    for i in range(ns):
        kx, ky, kd = knots_from_logits(outs[i], xlims[i], ylims[i])
        kx, ky, kd = augment_knots(kx, ky, kd, axis=1, **(extraps[i] or {}))
        f = rqs_invert if inverse else rqs_evaluate
        val, g = f(kx, ky, kd, xs[i], axis=1)
        vals.append(val * active_mask)
        logs = logs + sum_density(torch.log(g) * active_mask)
    return torch.cat(vals, dim=1), log0 + logs


def affine_coupling_atom(x_active, out, active_mask, *, inverse=False, log0=0):
    """t, s = chunk(out); s = |s|; y = t + x e^{-s}, logJ -= sum s
    (couplings_.py:123-139)."""
    t, s = out[:, 0] * active_mask, out[:, 1] * active_mask
    s = s.abs()
    if inverse:
        return (x_active - t) * torch.exp(s), log0 + sum_density(s)
    return t + x_active * torch.exp(-s), log0 - sum_density(s)


def shift_coupling_atom(x_active, out, active_mask, *, inverse=False, log0=0):
    """y = purify(x +- t)  (couplings_.py:110-116)."""
    t = out[:, 0]
    return ((x_active - t) if inverse else (x_active + t)) * active_mask, log0


# --------------------------------------------------------------- convolution stack
def circular_conv_direct(x, weight, bias=None):
    """Definition of the circular 'same' cross-correlation, any lattice dimension
    (what torch Conv{1,2,3}d(padding='same', padding_mode='circular') and the
    reference Conv4d compute; modules.py:120-145, convNd.py:86-126).

    x: (B, Cin, *L), weight: (Cout, Cin, *k) with odd k, bias: (Cout,) or None.
    out[b,o,n] = bias[o] + sum_{i,j} weight[o,i,j] * x[b,i,(n + j - k//2) mod L]."""
    d = x.dim() - 2
    ks = weight.shape[2:]
    out = None
    import itertools
    for j in itertools.product(*[range(k) for k in ks]):
        shifts = [-(jj - k // 2) for jj, k in zip(j, ks)]
        xs = torch.roll(x, shifts, dims=list(range(2, 2 + d)))
        w = weight[(slice(None), slice(None)) + j]          # (Cout, Cin)
        term = torch.einsum('oi,bi...->bo...', w, xs)
        out = term if out is None else out + term
    if bias is not None:
        out = out + bias.reshape([1, -1] + [1] * d)
    return out


def _circ_pad(x, pads):
    """Wrap-pad the trailing len(pads) axes by pads[i] on each side."""
    d = len(pads)
    for ax, p in zip(range(x.dim() - d, x.dim()), pads):
        if p:
            n = x.shape[ax]
            x = torch.cat((x.narrow(ax, n - p, p), x, x.narrow(ax, 0, p)), dim=ax)
    return x


def circular_conv_fast(x, weight, bias=None):
    """Same result as `circular_conv_direct` through torch's CPU convolution
    kernels (the code path the reference itself takes on CPU): wrap-pad then a
    'valid' conv; 4-D lattices are done as k0 shifted 3-D convolutions
    (the decomposition of convNd.py:104-124, restated)."""
    d = x.dim() - 2
    ks = weight.shape[2:]
    if d <= 3:
        xp = _circ_pad(x, [k // 2 for k in ks])
        return (F.conv1d, F.conv2d, F.conv3d)[d - 1](xp, weight, bias)
    assert d == 4
    B, Ci, L0 = x.shape[0], x.shape[1], x.shape[2]
    rest = x.shape[3:]
    xf = x.movedim(2, 1).reshape(B * L0, Ci, *rest)
    xf = _circ_pad(xf, [k // 2 for k in ks[1:]])
    out = None
    for j in range(ks[0]):
        o = F.conv3d(xf, weight[:, :, j]).reshape(B, L0, weight.shape[0], *rest)
        o = torch.roll(o, -(j - ks[0] // 2), dims=1)
        out = o if out is None else out + o
    out = out.movedim(1, 2)
    if bias is not None:
        out = out + bias.reshape(1, -1, 1, 1, 1, 1)
    return out.contiguous()


_ACTS = {
    None: lambda t: t, 'none': lambda t: t, 'tanh': torch.tanh, 'relu': torch.relu,
    'leaky_relu': lambda t: F.leaky_relu(t), 'softplus': lambda t: F.softplus(t),
    'abs': torch.abs, 'expit': torch.sigmoid,
}


def conv_act(x, layers, acts, conv=circular_conv_fast, pre_act=None):
    """ConvAct: a chain of circular convs and pointwise activations
    (modules.py:120-145).  layers: list of (weight, bias|None) in *standard*
    (Cout, Cin, *k) layout."""
    x = _ACTS[pre_act](x)
    for (w, b), a in zip(layers, acts):
        x = _ACTS[a](conv(x, w, b))
    return x


def conv4d_standard_weight(w_lower, out_channels, k0):
    """(out*k0, in, k,k,k) -> (out, in, k0, k,k,k)  (convNd.py:132-143)."""
    oc_k0, cin = w_lower.shape[0], w_lower.shape[1]
    return w_lower.reshape(out_channels, k0, cin, *w_lower.shape[2:]).movedim(1, 2)


# ------------------------------------------------------------------ coupling block
def coupling_block(x, nets, kind, lattice_shape, *, mask_parity=0, inverse=False,
                   log0=0, **opts):
    """A whole Coupling_ block (couplings_.py:54-78): split by the even-odd mask,
    alternate parity = k % 2 over the nets (reversed for the inverse), recombine.

    nets: list of callables (B,1,*L)->(B,C,*L).  kind in {'affine','shift','rqs'}.
    """
    atom = {'affine': affine_coupling_atom, 'shift': shift_coupling_atom,
            'rqs': rqs_coupling_atom, 'multirqs': multi_rqs_coupling_atom}[kind]
    masks = [channel_mask(lattice_shape, c, parity=mask_parity, dtype=x.dtype) for c in (0, 1)]
    parts = [x * masks[0], x * masks[1]]
    order = range(len(nets))
    for k in (reversed(order) if inverse else order):
        p = k % 2
        frozen = parts[1 - p]
        out = nets[k](frozen if kind == 'multirqs' else frozen.unsqueeze(1))
        parts[p], log0 = atom(parts[p], out, masks[p], inverse=inverse, log0=log0, **opts)
    return parts[0] + parts[1], log0


# ------------------------------------------------------------------ pointwise modules
def expit_(x, log0=0):
    """y = 1/(1+e^-x), logJ = sum(-x + 2 log y)   (modules_.py:93-102)."""
    y = 1 / (1 + torch.exp(-x))
    return y, log0 + sum_density(-x + 2 * torch.log(y))


def logit_(x, log0=0):
    """y = log(x/(1-x)), logJ = -sum log(x(1-x))   (modules_.py:105-114)."""
    return torch.log(x / (1 - x)), log0 - sum_density(torch.log(x * (1 - x)))


def shared_spline_knots(weights_x, weights_y, weights_d, xlim=(0, 1), ylim=(0, 1),
                        extrap=None):
    """SplineNet.make_spline for a single shared spline (modules.py:366-391):
    1-D knots from learned logits; weights_d None => 'smooth' derivatives
    (spline.py:126-152): average of neighbouring segment slopes, end slopes at
    the ends."""
    kx = coords_from_logits(weights_x, xlim[0], xlim[1] - xlim[0], axis=0)
    ky = coords_from_logits(weights_y, ylim[0], ylim[1] - ylim[0], axis=0)
    if weights_d is None:
        slope = (ky[1:] - ky[:-1]) / (kx[1:] - kx[:-1])
        kd = torch.cat((slope[:1], 0.5 * (slope[1:] + slope[:-1]), slope[-1:]))
    else:
        kd = softplus_ln2(weights_d)
    return augment_knots(kx, ky, kd, axis=0, **(extrap or {}))


def shared_spline_(x, knots, inverse=False, log0=0):
    """SplineNet_.forward/backward with spline_shape=[] (modules_.py:284-302):
    ravel, evaluate on the 1-D knots, reshape, logJ = sum log g."""
    kx, ky, kd = (k.reshape(-1, 1) for k in knots)
    v = x.reshape(1, -1)
    f = rqs_invert if inverse else rqs_evaluate
    val, g = f(kx, ky, kd, v, axis=0)
    return val.reshape(x.shape), log0 + sum_density(torch.log(g).reshape(x.shape))


def dist_convertor(x, weights_x, weights_y, weights_d, *, symmetric=False, inverse=False,
                   log0=0):
    """DistConvertor_ = Expit_ -> SplineNet_ -> Logit_ (modules_.py:333-358);
    symmetric => xlim=ylim=(0.5,1), left 'anti' (:345-346).  The inverse runs the
    chain reversed with each stage inverted (nn/_core.py:69-72)."""
    lim = (0.5, 1) if symmetric else (0, 1)
    extrap = {'left': 'anti'} if symmetric else {}
    knots = shared_spline_knots(weights_x, weights_y, weights_d, lim, lim, extrap)
    u, log0 = expit_(x, log0)
    u, log0 = shared_spline_(u, knots, inverse=inverse, log0=log0)
    return logit_(u, log0)


def scale_(x, raw_weight, inverse=False, log0=0):
    """ScaleNet_: x * softplus_ln2(w); logJ = +-V log w  (modules_.py:44-69)."""
    w = softplus_ln2(raw_weight)
    vol = x[0].numel()
    if inverse:
        return x / w, log0 - torch.log(w) * vol
    return x * w, log0 + torch.log(w) * vol


# ------------------------------------------------------------------ physics end points
def phi4_action(cfgs, *, m_sq, lambd, kappa=1.0, a=1.0):
    """S = sum(w2 phi^2 + w4 phi^4) - w0 sum_mu sum phi(x) phi(x - mu)
    (src/action/scalar_action.py:24-46)."""
    d = cfgs.dim() - 1
    kap = kappa * a ** (d - 2)
    w0 = kap
    w2 = 0.5 * (m_sq * a ** d + 2 * kap * d)
    w4 = lambd * a ** d
    S = sum_density(w2 * cfgs ** 2 + w4 * cfgs ** 4)
    for mu in range(1, d + 1):
        S = S - w0 * sum_density(cfgs * torch.roll(cfgs, 1, mu))
    return S


def normal_log_prob(x):
    """Unit-normal log density summed per sample (src/prior/prior.py:30-36,92-101)."""
    return sum_density(-0.5 * x * x - 0.5 * math.log(2 * math.pi))


def kl_loss(logq, logp):
    """Fitter.calc_kl_mean (src/_normflowcore.py:326-329)."""
    return (logq - logp).mean()


# ------------------------------------------------------------------ prior sampling
# The reference draws with torch.distributions.Normal.sample (src/prior/prior.py:26-29, 92-101) and then makes a second
# pass for log_prob (:30-36).  Its random STREAM is whatever torch's generator gives (mt19937 on CPU, torch's own Philox
# schedule on GPU) and is not part of the contract -- the distribution and the identity logr = log N(x) are.  The HIP
# kernel (nf_normal_sample) uses Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
# SC'11 -- the Random123 generator; absent from /root/reference, restated here from the published algorithm and pinned by
# Random123's known-answer vectors in tests/test_oracle_golden.py) with a counter layout of its own, restated below so
# that the kernel's draws can be checked number by number.
PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
PHILOX_KEY_DOMAIN = 0x6E66686B     # NF_PHILOX_KEY_DOMAIN of include/normflow_hip.h: XORed into the high key word


def philox4x32_10(counter, key):
    """counter: (..., 4) uint32, key: (..., 2) uint32 -> (..., 4) uint32 (numpy; 10 rounds)."""
    import numpy as np
    c = [np.asarray(counter[..., i], dtype=np.uint64) for i in range(4)]
    k0 = np.asarray(key[..., 0], dtype=np.uint64)
    k1 = np.asarray(key[..., 1], dtype=np.uint64)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(PHILOX_M0) * c[0]
        p1 = np.uint64(PHILOX_M1) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(PHILOX_W0)) & mask
        k1 = (k1 + np.uint64(PHILOX_W1)) & mask
    return np.stack([v.astype(np.uint32) for v in c], axis=-1)


def normal_prior_sample(seed, offset, B, V, loc=None, scale=None, dtype=torch.float32):
    """(x (B, V), logr (B)) exactly as nf_normal_sample lays its draws out (include/normflow_hip.h):
    group q = element index // 4 of sample b (float32: 4 normals per Philox call) or // 2 (float64: 2 per call);
    counter = (lo32(g), hi32(g), lo32(offset), hi32(offset)) with g = b * ngroups + q, key = (lo32(seed), hi32(seed) ^
    PHILOX_KEY_DOMAIN);
    Box-Muller: float32 u1 = (r + 1) 2^-32, u2 = r' 2^-32 from (r0, r1) -> (z0, z1) = rho (cos, sin)(2 pi u2) and (r2, r3) ->
    (z2, z3); float64 u1 = ((r0 << 21 ^ r1 >> 11) + 1) 2^-53, u2 likewise from (r2, r3) without the + 1.
    x = loc + scale z;  logr = sum_x [-z^2/2 - log scale - log sqrt(2 pi)]."""
    import numpy as np
    per = 4 if dtype == torch.float32 else 2
    ngroups = (V + per - 1) // per
    g = (np.arange(B, dtype=np.uint64)[:, None] * np.uint64(ngroups) + np.arange(ngroups, dtype=np.uint64)[None, :])
    ctr = np.stack([(g & np.uint64(0xFFFFFFFF)), (g >> np.uint64(32)),
                    np.full_like(g, offset & 0xFFFFFFFF), np.full_like(g, (offset >> 32) & 0xFFFFFFFF)], axis=-1).astype(np.uint32)
    key = np.broadcast_to(np.array([seed & 0xFFFFFFFF, ((seed >> 32) & 0xFFFFFFFF) ^ PHILOX_KEY_DOMAIN], dtype=np.uint32),
                          g.shape + (2,))
    r = philox4x32_10(ctr, key).astype(np.float64)
    if dtype == torch.float32:
        def bm(ra, rb):
            u1, u2 = (ra + 1.0) * 2.0 ** -32, rb * 2.0 ** -32
            rho = np.sqrt(-2.0 * np.log(u1))
            return rho * np.cos(2 * np.pi * u2), rho * np.sin(2 * np.pi * u2)
        z0, z1 = bm(r[..., 0], r[..., 1])
        z2, z3 = bm(r[..., 2], r[..., 3])
        z = np.stack([z0, z1, z2, z3], axis=-1)
    else:
        ri = philox4x32_10(ctr, key).astype(np.uint64)
        a = ((ri[..., 0] << np.uint64(21)) ^ (ri[..., 1] >> np.uint64(11))).astype(np.float64)
        b = ((ri[..., 2] << np.uint64(21)) ^ (ri[..., 3] >> np.uint64(11))).astype(np.float64)
        u1, u2 = (a + 1.0) * 2.0 ** -53, b * 2.0 ** -53
        rho = np.sqrt(-2.0 * np.log(u1))
        z = np.stack([rho * np.cos(2 * np.pi * u2), rho * np.sin(2 * np.pi * u2)], axis=-1)
    z = torch.from_numpy(z.reshape(B, ngroups * per)[:, :V]).to(torch.float64)
    loc = torch.zeros(V, dtype=torch.float64, device='cpu') if loc is None else loc.double().reshape(-1)
    scale = torch.ones(V, dtype=torch.float64, device='cpu') if scale is None else scale.double().reshape(-1)
    x = loc + scale * z
    logr = (-0.5 * z * z - torch.log(scale) - 0.5 * math.log(2 * math.pi)).sum(dim=1)
    return x.to(dtype), logr.to(dtype)
