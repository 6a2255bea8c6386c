"""The spline object of a coupling layer (reference: src/lib/spline/spline.py, `RQSpline` = `Pade22Spline`).

In the reference `RQSplineCoupling_.make_spline(out)` (src/nn/scalar/couplings_.py:211-262) turns the net output into knot
tensors and returns an `RQSpline` whose `forward / backward(x, grad=True)` give the map and its derivative at every site
(spline.py:87-123).  Here the same object is a thin handle on the logits: the knots are built and evaluated by the HIP
kernels (`nf_rqs_fwd_sites`, `nf_rqs_inv_sites`, `nf_rqs_knots` of include/normflow_hip.h), with the arithmetic of the
coupling layer itself, so what `_hack` shows is what the layer computes.  An inspection path: no autograd, float32/float64.

The reference's own constructor form, `RQSpline(knots_x=..., knots_y=..., knots_d=..., knots_axis=-1, extrap={})` from
explicit knot tensors (spline.py:39-68), is provided too: the knots are augmented on the host as a layout operation
(spline.py:458-532) and evaluated by `nf_spline_eval` (searchsorted + clamp + segment function / stable inverse root).
"""
import torch

from .. import _hip


class RQSpline:
    """Monotone rational-quadratic spline, one per lattice site, given by a coupling layer's raw logits.

    logits: (B, C, *L) along `knots_axis` = 1 (other axes are moved there), C = 3m-2, or 2m-1 / m with fixed 1-D
    knots_x / knots_y.  `extrap` as the reference: {'left': None|'linear'|'anti', 'right': ...}.
    """

    def __init__(self, logits=None, *, xlim=(0, 1), ylim=(0, 1), knots_x=None, knots_y=None, knots_d=None, extrap=None,
                 knots_axis=None):
        self.extrap = {k: ('anti' if v == 'anti-periodic' else v) for k, v in dict(extrap or {}).items()}
        self._knots = None
        self._explicit = logits is None
        if self._explicit:
            self._init_explicit(knots_x, knots_y, knots_d, -1 if knots_axis is None else knots_axis)
            return
        knots_axis = 1 if knots_axis is None else knots_axis
        if knots_axis not in (1, 1 - logits.dim()):
            logits = logits.movedim(knots_axis, 1)
        self.knots_axis = knots_axis
        self.lattice = tuple(logits.shape[2:])
        self._logits = logits.detach().reshape(logits.shape[0], logits.shape[1], -1).contiguous()
        fixed = lambda k: None if k is None else torch.as_tensor(k).detach().to(
            device=logits.device, dtype=logits.dtype).contiguous()
        self._fx, self._fy = fixed(knots_x), fixed(knots_y)
        n_fixed = (self._fx is not None) + (self._fy is not None)
        C = self._logits.shape[1]
        if (C + 2 - n_fixed) % (3 - n_fixed):
            raise Exception(f"{C} channels do not make a spline: {3 - n_fixed}m-{2 - n_fixed} are needed for m knots")
        self.m = (C + 2 - n_fixed) // (3 - n_fixed)
        self.xlim, self.ylim = tuple(xlim), tuple(ylim)

    # ------------------------------------------------------------------ explicit knots (spline.py:39-68)
    def _init_explicit(self, knots_x, knots_y, knots_d, knots_axis):
        if knots_x is None or knots_y is None:
            raise Exception("RQSpline needs either a coupling layer's logits or knots_x and knots_y")
        kx, ky = torch.as_tensor(knots_x).detach(), torch.as_tensor(knots_y).detach()
        conflict = lambda a, b: (a.shape != b.shape and a.dim() > 1 and b.dim() > 1)
        if conflict(kx, ky):
            raise Exception("x & y must be the same shape unless one is 1 dim.")
        if kx.dtype not in (torch.float32, torch.float64):
            raise TypeError(f"knots must be float32 / float64, got {kx.dtype}")
        ky = ky.to(device=kx.device, dtype=kx.dtype)
        nd = max(kx.dim(), ky.dim())
        ax = knots_axis % nd if nd > 1 else 0
        # working layout: the knots axis first, everything else flattened -> (K, N) planes, or (K,) shared vectors
        flat = lambda t: t if t.dim() == 1 else t.movedim(ax, 0).reshape(t.shape[ax], -1)
        if knots_d is None:
            knots_d = self.smooth_derivatives(kx, ky, ax if nd > 1 else 0)
        kd = torch.as_tensor(knots_d).detach().to(device=kx.device, dtype=kx.dtype)
        if conflict(kd, kx) or conflict(kd, ky):
            raise Exception("shape conflict between d, x, and y.")
        full = max((kx, ky, kd), key=lambda t: t.dim())
        self._axis, self._full_shape = ax, tuple(full.shape)
        self.knots_axis = knots_axis
        kx, ky, kd = flat(kx), flat(ky), flat(kd)
        if any(self.extrap.get(s) for s in ('left', 'right')) and not (kx.dim() == ky.dim() == kd.dim()):
            # the boundary rules mix x, y and d: give every knot tensor the full shape first
            N = max(t.shape[1] for t in (kx, ky, kd) if t.dim() == 2)
            kx, ky, kd = (t if t.dim() == 2 else t.unsqueeze(1).expand(-1, N) for t in (kx, ky, kd))
        kx, ky, kd = self._augment(kx, ky, kd, axis=0)
        self._kx, self._ky, self._kd = (t.contiguous() for t in (kx, ky, kd))
        K = self._kx.shape[0]
        if self._ky.shape[0] != K or self._kd.shape[0] != K:
            raise Exception("shape conflict between d, x, and y.")

    @staticmethod
    def smooth_derivatives(knots_x, knots_y, knots_axis, bc_type='not-ones'):
        """Interior knots: the average of the two neighbouring segment slopes; end knots: the slope of the end segment,
        or 1 with bc_type='ones' (spline.py:125-152)."""
        nd = max(knots_x.dim(), knots_y.dim())
        sl = lambda t, a, b: t.narrow(knots_axis if t.dim() == nd else 0, a, b)
        n = knots_x.shape[knots_axis if knots_x.dim() == nd else 0]
        diff = lambda t: sl(t, 1, n - 1) - sl(t, 0, n - 1)
        dx, dy = diff(knots_x), diff(knots_y)
        if dx.dim() != dy.dim():               # a 1-D vector against an N-D tensor: line it up on the knots axis
            view = [1] * nd
            view[knots_axis] = -1
            dx, dy = (t.reshape(view) if t.dim() == 1 else t for t in (dx, dy))
        m = dy / dx
        ax = knots_axis if m.dim() == nd else 0
        avg = 0.5 * (m.narrow(ax, 1, n - 2) + m.narrow(ax, 0, n - 2))
        if bc_type == 'ones':
            left = right = torch.ones_like(m.narrow(ax, 0, 1))
        else:
            left, right = m.narrow(ax, 0, 1), m.narrow(ax, n - 2, 1)
        return torch.cat((left, avg, right), ax)

    def _map_explicit(self, v, inverse, grad, squeezed):
        v = v.detach()
        _hip._require_device(v, self._kx)
        v = v.to(self._kx.dtype)
        shared = [t.dim() == 1 for t in (self._kx, self._ky, self._kd)]
        K = self._kx.shape[0]
        if all(shared):                        # one spline for every input value (spline.py:163-164)
            flat = v.reshape(1, -1).contiguous()
            out, der = _hip.spline_eval(flat, self._kx, self._ky, self._kd, K, shared, inverse, grad)
            return (out.reshape(v.shape), der.reshape(v.shape)) if grad else out.reshape(v.shape)
        ax = self._axis
        vv = v.unsqueeze(ax) if squeezed else v
        rows = vv.movedim(ax, 0)
        shape_rows = rows.shape
        rows = rows.reshape(rows.shape[0], -1).contiguous()
        N = max(t.shape[1] for t in (self._kx, self._ky, self._kd) if t.dim() == 2)
        if rows.shape[1] != N:
            raise Exception(f"input of shape {tuple(v.shape)} does not match the knots' shape {self._full_shape}"
                            + ("" if squeezed else " (pass squeezed=True for an input without the knots axis)"))
        outs, ders = [], []
        for r in range(rows.shape[0]):         # every entry along the knots axis is a point on the same splines
            o, d = _hip.spline_eval(rows[r:r + 1], self._kx, self._ky, self._kd, K, shared, inverse, grad)
            outs.append(o)
            ders.append(d)
        back = lambda ts: torch.cat(ts).reshape(shape_rows).movedim(0, ax)
        out = back(outs)
        out = out.squeeze(ax) if squeezed else out
        if not grad:
            return out
        der = back(ders)
        return out, (der.squeeze(ax) if squeezed else der)

    def _opts(self):
        return _hip.make_rqs_opts(self.m, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_FULL, self._fx, self._fy)

    # ------------------------------------------------------------------ evaluation
    def _map(self, v, inverse, grad, squeezed, activity=None, log=False):
        if self._explicit:
            return self._map_explicit(v, inverse, grad, squeezed)
        B = self._logits.shape[0]
        shape = v.shape
        flat = v.detach().reshape(B, -1)
        if flat.shape[1] != self._logits.shape[2]:
            raise Exception(f"input of shape {tuple(shape)} does not match the knots' lattice {self.lattice}"
                            + ("" if squeezed else " (pass squeezed=True for an input without the knots axis)"))
        if activity is None:
            activity = torch.ones(flat.shape[1], dtype=torch.uint8, device=flat.device)
        out, _, sites = _hip.rqs_sites(flat, self._logits, activity, None, self._opts(), inverse,
                                       _hip.SITES_LOG if log else _hip.SITES_DERIVATIVE)
        out = out.reshape(shape)
        return (out, sites.reshape(shape)) if grad else out

    def forward(self, x, grad=False, squeezed=False):
        """y(x) at every site, and dy/dx with grad=True (spline.py:87-112).  x: (B, 1, *L), or (B, *L) with squeezed=True."""
        return self._map(x, False, grad, squeezed)

    def backward(self, y, grad=False, squeezed=False):
        """x(y), and dx/dy with grad=True (spline.py:114-123)."""
        return self._map(y, True, grad, squeezed)

    __call__ = forward

    # ------------------------------------------------------------------ knots
    def _knot_tensors(self):
        if self._explicit:
            return self._kx, self._ky, self._kd
        if self._knots is None:
            k = _hip.rqs_knots(self._logits, self._opts())
            B, m = k.shape[0], self.m
            kx, ky, kd = (k[:, i * m:(i + 1) * m].reshape((B, m) + self.lattice) for i in range(3))
            self._knots = self._augment(kx, ky, kd)
        return self._knots

    def _augment(self, kx, ky, kd, axis=1):
        """Boundary knots as the reference stores them (spline.py:458-532): 'linear' = one knot one unit outside on the tangent
        line; 'anti' = every other knot mirrored through the end knot, derivatives unchanged.  (For a spline made from
        logits this is the layout of the stored knots only: the coupling kernels evaluate the tails / reflect the argument
        without materialising these.  For explicit knots it is what nf_spline_eval then evaluates.)"""
        left, right = self.extrap.get('left'), self.extrap.get('right')
        for side in (left, right):
            if side not in (None, 'linear', 'anti'):
                raise Exception(f"extrapolation {side!r} is not supported (supported: None, 'linear', 'anti')")
        first = lambda t: t.narrow(axis, 0, 1)
        last = lambda t: t.narrow(axis, t.shape[axis] - 1, 1)
        if left == 'linear' or right == 'linear':
            xs, ys, ds = [kx], [ky], [kd]
            if left == 'linear':
                xs.insert(0, first(kx) - 1); ys.insert(0, first(ky) - first(kd)); ds.insert(0, first(kd))
            if right == 'linear':
                xs.append(last(kx) + 1); ys.append(last(ky) + last(kd)); ds.append(last(kd))
            kx, ky, kd = (torch.cat(t, dim=axis) for t in (xs, ys, ds))
            if left is None or right is None:
                return kx, ky, kd
        if left == 'anti' or right == 'anti':
            n = kx.shape[axis]
            xs, ys, ds = [kx], [ky], [kd]
            if left == 'anti':
                rest = lambda t: torch.flip(t.narrow(axis, 1, n - 1), [axis])
                xs.insert(0, 2 * first(kx) - rest(kx)); ys.insert(0, 2 * first(ky) - rest(ky)); ds.insert(0, rest(kd))
            if right == 'anti':
                rest = lambda t: torch.flip(t.narrow(axis, 0, n - 1), [axis])
                xs.append(2 * last(kx) - rest(kx)); ys.append(2 * last(ky) - rest(ky)); ds.append(rest(kd))
            kx, ky, kd = (torch.cat(t, dim=axis) for t in (xs, ys, ds))
        return kx, ky, kd

    def _on_axis(self, t):
        if self._explicit:                     # (K, N) working layout -> the caller's shape, knots on their axis
            if t.dim() == 1:
                return t
            shape = list(self._full_shape)
            shape[self._axis] = t.shape[0]
            moved = [shape[self._axis]] + [n for i, n in enumerate(shape) if i != self._axis]
            return t.reshape(moved).movedim(0, self._axis)
        return t if self.knots_axis in (1, 1 - t.dim()) else t.movedim(1, self.knots_axis)

    @property
    def knots_x(self):
        return self._on_axis(self._knot_tensors()[0])

    @property
    def knots_y(self):
        return self._on_axis(self._knot_tensors()[1])

    @property
    def knots_d(self):
        return self._on_axis(self._knot_tensors()[2])

    @property
    def knots_shape(self):
        return self.knots_x.shape

    @property
    def knots_len(self):
        return self._knot_tensors()[0].shape[0 if self._explicit else 1]

    @property
    def segm_len(self):
        return self.knots_len - 1
