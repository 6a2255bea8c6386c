"""The spline object of a coupling layer (reference: src/lib/spline/spline.py, `RQSpline` = `Pade22Spline`).

In the reference `RQSplineCoupling_.make_spline(out)` (src/nn/scalar/couplings_.py:211-262) turns the net output into knot
tensors and returns an `RQSpline` whose `forward / backward(x, grad=True)` give the map and its derivative at every site
(spline.py:87-123).  Here the same object is a thin handle on the logits: the knots are built and evaluated by the HIP
kernels (`nf_rqs_fwd_sites`, `nf_rqs_inv_sites`, `nf_rqs_knots` of include/normflow_hip.h), with the arithmetic of the
coupling layer itself, so what `_hack` shows is what the layer computes.  An inspection path: no autograd, float32/float64.
"""
import torch

from .. import _hip


class RQSpline:
    """Monotone rational-quadratic spline, one per lattice site, given by a coupling layer's raw logits.

    logits: (B, C, *L) along `knots_axis` = 1 (other axes are moved there), C = 3m-2, or 2m-1 / m with fixed 1-D
    knots_x / knots_y.  `extrap` as the reference: {'left': None|'linear'|'anti', 'right': ...}.
    """

    def __init__(self, logits, *, xlim=(0, 1), ylim=(0, 1), knots_x=None, knots_y=None, extrap=None, knots_axis=1):
        if knots_axis not in (1, 1 - logits.dim()):
            logits = logits.movedim(knots_axis, 1)
        self.knots_axis = knots_axis
        self.extrap = dict(extrap or {})
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
        self._knots = None

    def _opts(self):
        return _hip.make_rqs_opts(self.m, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_FULL, self._fx, self._fy)

    # ------------------------------------------------------------------ evaluation
    def _map(self, v, inverse, grad, squeezed, activity=None, log=False):
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
        if self._knots is None:
            k = _hip.rqs_knots(self._logits, self._opts())
            B, m = k.shape[0], self.m
            kx, ky, kd = (k[:, i * m:(i + 1) * m].reshape((B, m) + self.lattice) for i in range(3))
            self._knots = self._augment(kx, ky, kd)
        return self._knots

    def _augment(self, kx, ky, kd):
        """Boundary knots as the reference stores them (spline.py:458-532): 'linear' = one knot one unit outside on the tangent
        line; 'anti' = every other knot mirrored through the end knot, derivatives unchanged.  (Layout of the stored knots
        only: the kernels evaluate the tails / reflect the argument without materialising these.)"""
        left, right = self.extrap.get('left'), self.extrap.get('right')
        first = lambda t: t[:, :1]
        last = lambda t: t[:, -1:]
        if left == 'linear' or right == 'linear':
            xs, ys, ds = [kx], [ky], [kd]
            if left == 'linear':
                xs.insert(0, first(kx) - 1); ys.insert(0, first(ky) - first(kd)); ds.insert(0, first(kd))
            if right == 'linear':
                xs.append(last(kx) + 1); ys.append(last(ky) + last(kd)); ds.append(last(kd))
            kx, ky, kd = (torch.cat(t, dim=1) for t in (xs, ys, ds))
            if left is None or right is None:
                return kx, ky, kd
        if left == 'anti' or right == 'anti':
            n = kx.shape[1]
            xs, ys, ds = [kx], [ky], [kd]
            if left == 'anti':
                rest = lambda t: torch.flip(t[:, 1:n], [1])
                xs.insert(0, 2 * first(kx) - rest(kx)); ys.insert(0, 2 * first(ky) - rest(ky)); ds.insert(0, rest(kd))
            if right == 'anti':
                rest = lambda t: torch.flip(t[:, 0:n - 1], [1])
                xs.append(2 * last(kx) - rest(kx)); ys.append(2 * last(ky) - rest(ky)); ds.append(rest(kd))
            kx, ky, kd = (torch.cat(t, dim=1) for t in (xs, ys, ds))
        return kx, ky, kd

    def _on_axis(self, t):
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
        return self._knot_tensors()[0].shape[1]

    @property
    def segm_len(self):
        return self.knots_len - 1
