"""Coupling layers on the HIP kernels (reference: src/nn/scalar/couplings_.py).

Class names, constructor signatures, the `atomic_forward / atomic_backward` protocol
and the `(value, log0 + log|J|)` return convention are the reference's; what differs
is what happens inside an "atom": where the reference runs ~150 eager ops per layer
(split, softmax, cumsum, cat, softplus, searchsorted on transposed copies, 6 gathers,
~40 pointwise ops, 2 purify, log, sum), an atom here is the parameter net followed by
ONE fused kernel launch (+ a tiny reduction epilogue) through `normflow__amd._hip`.
"""
from abc import ABC, abstractmethod

import torch

from .._core import Module_
from ... import _hip
from ...lib.spline import RQSpline

# Largest raw-logit tensor (bytes) an atom materialises at once; bigger batches are
# cut into slabs (32^4, m=16: one sample's logits are 193 MB in fp32).
PARAM_SLAB_BYTES = 6 << 30
# Largest hidden-activation tensor (fp32-equivalent bytes) a fused atom materialises at once: 1024 samples of 32^4 -- two such
# tensors are alive at a time, 64 of the GPU's 288 GB.  Fewer, longer launches of each kernel: the headline step is 3 % faster
# with 1024-sample slabs than with 256 (tools/slab_ab.py; same bits).  On a device with less memory the budget is an eighth
# of its TOTAL memory (a fixed number per device: a budget that followed the free memory changed the slab sizes from pass to
# pass and with them every allocation), never below HIDDEN_SLAB_FLOOR; a slab never exceeds 2^30 sites (the kernels index
# sites in 32 bits).
HIDDEN_SLAB_BYTES = 32 << 30
HIDDEN_SLAB_FLOOR = 8 << 30
_DEVICE_TOTAL = {}


# Training: an RQ-spline atom whose (B, 3m-2, V/2) logits would exceed this many bytes runs its last layer + spline as ONE
# differentiable node that never materialises them (`_hip.FusedLastRqsFn`: -96 MB per sample and net at 32^4, +17 % step
# time because the backward pass recomputes the layer); smaller atoms keep the faster materialising path.  0 = always fused.
TRAIN_FUSED_MIN_LOGIT_BYTES = 1 << 30


def set_training_fusion(min_logit_bytes):
    """Byte threshold above which a training atom uses the logit-free fused node (0: always; a huge number: never).
    Returns the previous threshold."""
    global TRAIN_FUSED_MIN_LOGIT_BYTES
    old = TRAIN_FUSED_MIN_LOGIT_BYTES
    TRAIN_FUSED_MIN_LOGIT_BYTES = int(min_logit_bytes)
    return old


def set_slab_bytes(params=None, hidden=None):
    """Planner knobs of this module (the product reads no environment variable): byte budgets of the logit slab of an
    unfused atom and of the hidden-activation slab of a fused one.  Returns the previous (params, hidden)."""
    global PARAM_SLAB_BYTES, HIDDEN_SLAB_BYTES
    old = (PARAM_SLAB_BYTES, HIDDEN_SLAB_BYTES)
    if params is not None:
        PARAM_SLAB_BYTES = int(params)
    if hidden is not None:
        HIDDEN_SLAB_BYTES = int(hidden)
    return old


def _activity_bytes(mask, channel, lattice_shape, device):
    """(V,) uint8 on `device`: 1 where the site belongs to `channel`.  Masks of this
    package expose it directly; any other object with `purify` is probed once."""
    if hasattr(mask, 'activity'):
        act = mask.activity(channel)
    else:
        probe = torch.ones(tuple(lattice_shape), dtype=torch.float32, device=device)
        act = (mask.purify(probe, channel) != 0).to(torch.uint8)
    return act.to(device=device, dtype=torch.uint8).reshape(-1).contiguous()


class Coupling_(Module_, ABC):
    """A stack of coupling layers acting alternately on the two partitions of a mask
    (couplings_.py:22-103).

    nets : list of nn.Module mapping (B, 1, *L) -> (B, C, *L)
    mask : object with split / cat / purify (e.g. normflow__amd.mask.EvenOddMask)
    """

    def __init__(self, nets, *, mask, channels_axis=1, label='coupling_'):
        super().__init__(label=label)
        self.nets = torch.nn.ModuleList(nets)
        self.mask = mask
        self.channels_axis = channels_axis
        self._act_cache = {}

    # ---- block level: identical protocol to the reference (couplings_.py:54-78)
    def forward(self, x, log0=0):
        parts, log0 = self.parts_forward(list(self.mask.split(x)), log0)
        return self.mask.cat(*parts), log0

    def backward(self, x, log0=0):
        parts, log0 = self.parts_backward(list(self.mask.split(x)), log0)
        return self.mask.cat(*parts), log0

    # the block on the two parts of the field (`ModuleList_` chains consecutive blocks over one partition through these,
    # without the cat / split passes in between)
    def parts_forward(self, parts, log0=0):
        for k, net in enumerate(self.nets):
            p = k % 2
            parts[p], log0 = self.atomic_forward(x_active=parts[p], x_frozen=parts[1 - p],
                                                 parity=p, net=net, log0=log0)
        return parts, log0

    def parts_backward(self, parts, log0=0):
        for k in reversed(range(len(self.nets))):
            p = k % 2
            parts[p], log0 = self.atomic_backward(x_active=parts[p], x_frozen=parts[1 - p],
                                                  parity=p, net=self.nets[k], log0=log0)
        return parts, log0

    @abstractmethod
    def atomic_forward(self, *, x_active, x_frozen, parity, net, log0=0):
        pass

    @abstractmethod
    def atomic_backward(self, *, x_active, x_frozen, parity, net, log0=0):
        pass

    def preprocess_fz(self, x):
        return x.unsqueeze(self.channels_axis)

    def preprocess(self, x):
        return x.unsqueeze(self.channels_axis)

    def postprocess(self, x):
        return x.squeeze(self.channels_axis)

    def _ctor_kwargs(self):
        return dict(label=self.label, channels_axis=self.channels_axis)

    def transfer(self, scale_factor=1, mask=None, **extra):
        nets = [net.transfer(scale_factor=scale_factor) for net in self.nets]
        return self.__class__(nets, mask=self.mask if mask is None else mask, **self._ctor_kwargs())

    # ---- kernel plumbing
    def _activity(self, parity, lattice_shape, device):
        key = (parity, tuple(lattice_shape), str(device))
        hit = self._act_cache.get(key)
        if hit is None:
            hit = _activity_bytes(self.mask, parity, lattice_shape, device)
            self._act_cache = {key: hit, **{k: v for k, v in self._act_cache.items() if k[0] != parity}}
        return hit

    def _check_density(self):
        if self.propagate_density:
            raise NotImplementedError("propagate_density=True (per-site densities) is not provided by "
                                      "this coupling's kernels, which reduce log|J| per sample")

    def _no_grad_only(self, x_active, x_frozen, net):
        if torch.is_grad_enabled() and (x_active.requires_grad or x_frozen.requires_grad
                                        or any(p.requires_grad for p in net.parameters())):
            raise NotImplementedError("propagate_density=True is an inference path here (per-site densities have no VJP "
                                      "kernel); wrap the call in torch.no_grad()")

    def _density_slabs(self, x_active, x_frozen, net, n_out_hint, one_slab):
        """Per-site densities, slab by slab like `_run_atom` (the logits of a whole batch need not fit in memory):
        `one_slab(x_active[b0:b1], x_frozen[b0:b1]) -> (value, per-site log-derivative)`, written into preallocated rows."""
        B = x_active.shape[0]
        n_out = getattr(net, 'conv_kwargs', {}).get('out_channels', n_out_hint)
        per_sample = n_out * x_active[0].numel() * x_active.element_size()
        val, sites = torch.empty_like(x_active), torch.empty_like(x_active)
        for b0, b1 in self._slabs(B, per_sample):
            val[b0:b1], sites[b0:b1] = one_slab(x_active[b0:b1], x_frozen[b0:b1])
        return val, sites

    def _affine_density_atom(self, inverse, x_active, x_frozen, parity, net, log0):
        """propagate_density (nn/_core.py:19,38-42) for the affine layer: log0 + the log-derivative of every site
        (nf_affine_sites), nothing summed.  Inference only."""
        self._no_grad_only(x_active, x_frozen, net)
        act = self._activity(parity, x_active.shape[1:], x_active.device)

        def one_slab(xa, xf):
            params, layout = self._params(net, xf, None)
            val, _, sites = _hip.affine_sites(xa.reshape(xa.shape[0], -1), params, act, None, layout, inverse)
            return val.reshape(xa.shape), sites.reshape(xa.shape)
        val, sites = self._density_slabs(x_active, x_frozen, net, 2, one_slab)
        return val, log0 + sites

    def _params(self, net, x_frozen, parity=None):
        """Run the parameter net; return raw logits as (B, C, V) [or (B, C, V/2)] and the
        layout code.  A ConvAct on a plain even-odd mask emits only the active sites.
        fp16 fields (BASELINE config 5): the net computes in fp32 on the widened frozen half."""
        if x_frozen.dtype == torch.float16:
            x_frozen = x_frozen.float()
        if (parity is not None and self.channels_axis == 1 and hasattr(net, 'forward_active')
                and getattr(self.mask, 'pairable', False) and hasattr(self.mask, 'checkerboard_parity')):
            a = self.mask.checkerboard_parity(parity)
            if a is not None:
                out = net.forward_active(self.preprocess_fz(x_frozen), a)
                if out is not None:
                    return out, _hip.LAYOUT_PAIR
        out = net(self.preprocess_fz(x_frozen))
        if self.channels_axis not in (1, 1 - out.dim()):
            out = out.movedim(self.channels_axis, 1)
        return out.reshape(out.shape[0], out.shape[1], -1), _hip.LAYOUT_FULL

    def _small_lattice_atom(self, kind, inverse, x_active, x_frozen, parity, net, log0, opts=None):
        """Small lattices (L0, L1 <= 16, 16) or (L1 <= 16, 16) that fit a CU's LDS -- BASELINE configs 3 (16^3) and 2 (16^2) --:
        the WHOLE atom (parameter net 1 -> h -> h -> C and the coupling, kind 0 RQ-spline / 1 affine) is ONE launch of
        nf_small_lattice_coupling; nothing but x and y touches HBM.  None when it does not apply."""
        if (torch.is_grad_enabled() and (x_active.requires_grad or x_frozen.requires_grad
                                         or any(p.requires_grad for p in net.parameters()))):
            return None
        if (self.propagate_density or self.channels_axis != 1 or x_active.dim() not in (3, 4) or x_active.dtype != torch.float32
                or not hasattr(net, 'small3d_plan') or not getattr(self.mask, 'pairable', False)
                or not hasattr(self.mask, 'checkerboard_parity')):
            return None
        a = self.mask.checkerboard_parity(parity)
        if a is None:
            return None
        plan = net.small3d_plan()
        if plan is None:
            return None
        import ctypes as C
        packed, biases, acts, cout = plan
        lat = tuple(x_active.shape[1:])
        m = opts.m if opts is not None else 0
        if not _hip.load().nf_small_lattice_supported((C.c_int32 * len(lat))(*lat), len(lat), kind, cout, m, acts[0], acts[1]):
            return None
        B = x_active.shape[0]
        l0 = _hip._log0_tensor(log0, x_active, B)
        return _hip.small_lattice_coupling(kind, x_frozen, x_active, packed, biases, l0, a, cout, acts, opts, inverse)

    def _slabs(self, B, per_sample_bytes, budget=None):
        step = max(1, min(B, (budget or PARAM_SLAB_BYTES) // max(1, per_sample_bytes)))
        return [(b0, min(B, b0 + step)) for b0 in range(0, B, step)]

    def _hidden_slabs(self, v, hidden):
        """Slabs of a fused atom (hidden activations of `hidden` channels for the samples of v (B, V))."""
        B, V = v.shape
        per_sample = hidden * V * 4
        budget = HIDDEN_SLAB_BYTES
        if B * per_sample > HIDDEN_SLAB_FLOOR and budget > HIDDEN_SLAB_FLOOR and v.is_cuda:
            total = _DEVICE_TOTAL.get(v.device)
            if total is None:
                total = _DEVICE_TOTAL[v.device] = torch.cuda.get_device_properties(v.device).total_memory
            budget = max(HIDDEN_SLAB_FLOOR, min(budget, total // 8))
        step = max(1, min(B, budget // max(1, per_sample), (1 << 30) // max(1, V)))
        return [(b0, min(B, b0 + step)) for b0 in range(0, B, step)]

    def _run_atom(self, kernel, x_active, x_frozen, parity, net, log0, n_out_hint, density_ok=False):
        """Common driver: slab the batch, produce logits, launch `kernel(v, params, l0, act, layout)`."""
        if not density_ok:
            self._check_density()
        B = x_active.shape[0]
        lattice = x_active.shape[1:]
        act = self._activity(parity, lattice, x_active.device)
        v = x_active.reshape(B, -1)
        l0 = _hip._log0_tensor(log0, v, B)
        per_sample = n_out_hint * v.shape[1] * v.element_size()
        vals, logs = [], []
        for b0, b1 in self._slabs(B, per_sample):
            params, layout = self._params(net, x_frozen[b0:b1], parity)
            val, lj = kernel(v[b0:b1], params, None if l0 is None else l0[b0:b1], act, layout)
            vals.append(val)
            logs.append(lj)
        val = vals[0] if len(vals) == 1 else torch.cat(vals)
        lj = logs[0] if len(logs) == 1 else torch.cat(logs)
        return val.reshape(x_active.shape), lj


class ShiftCoupling_(Coupling_):
    """y = purify(x + t) (couplings_.py:107-116); log|J| unchanged."""

    def _atom(self, inverse, *, x_active, x_frozen, parity, net, log0=0):
        k = lambda v, p, l0, act, layout: _hip.AffineCouplingFn.apply(v, p, l0, act, layout, inverse)
        if self.propagate_density:
            # a shift has unit Jacobian: the reference hands log0 back untouched, whatever its shape (couplings_.py:110-116)
            val, _ = self._run_atom(k, x_active, x_frozen, parity, net, 0, 1, density_ok=True)
            return val, log0
        val, lj = self._run_atom(k, x_active, x_frozen, parity, net, log0, 1)
        return val, (lj if torch.is_tensor(log0) or log0 != 0 else log0)

    def atomic_forward(self, **kw):
        return self._atom(False, **kw)

    def atomic_backward(self, **kw):
        return self._atom(True, **kw)


class AffineCoupling_(Coupling_):
    """y = t + x e^{-|s|}, log|J| = -sum|s| over the active sites (couplings_.py:120-139)."""

    def _fused_atom(self, inverse, x_active, x_frozen, parity, net, log0):
        """Inference fast path on the split-fp16 chain: first layer -> pair tensor, hidden layers, and the net's last layer
        (8 -> 2) fused with the affine map (nf_conv_affine_split16); the (t, s) tensor never reaches HBM.  None when it does
        not apply (then: conv stack + nf_affine)."""
        if (torch.is_grad_enabled() and (x_active.requires_grad or x_frozen.requires_grad
                                         or any(p.requires_grad for p in net.parameters()))):
            return None
        if (self.propagate_density or self.channels_axis != 1 or not hasattr(net, 'hidden_and_last')
                or not getattr(self.mask, 'pairable', False) or x_active.dtype not in (torch.float32, torch.float16)
                or not hasattr(self.mask, 'checkerboard_parity') or x_active.dim() != 5):
            return None
        a = self.mask.checkerboard_parity(parity)
        if a is None or net.conv_kwargs['out_channels'] != 2:
            return None
        B = x_active.shape[0]
        v = x_active.reshape(B, -1).contiguous()
        l0 = _hip._log0_tensor(log0, v, B)
        hidden = max(net.conv_kwargs['hidden_sizes'] or [1])
        val = torch.empty_like(v)
        lj = torch.empty(B, dtype=torch.float32, device=v.device)
        lattice = tuple(x_frozen.shape[1:])
        for b0, b1 in self._hidden_slabs(v, hidden):
            xf = x_frozen[b0:b1]
            got = net.hidden_and_last(self.preprocess_fz(xf.float() if xf.dtype == torch.float16 else xf), last_kind='affine')
            if got is None:
                return None
            h16, last, unit, split = got
            _hip.conv_affine_split16(h16, last.weight, last.bias, v[b0:b1], None if l0 is None else l0[b0:b1], a, inverse,
                                     lattice, out=(val[b0:b1], lj[b0:b1]))
        return val.reshape(x_active.shape), lj

    def _atom(self, inverse, *, x_active, x_frozen, parity, net, log0=0):
        if self.propagate_density:
            return self._affine_density_atom(inverse, x_active, x_frozen, parity, net, log0)
        fused = self._fused_atom(inverse, x_active, x_frozen, parity, net, log0)
        if fused is None and getattr(net, 'conv_kwargs', {}).get('out_channels') == 2:
            fused = self._small_lattice_atom(1, inverse, x_active, x_frozen, parity, net, log0)
        if fused is not None:
            return fused
        k = lambda v, p, l0, act, layout: _hip.AffineCouplingFn.apply(v, p, l0, act, layout, inverse)
        return self._run_atom(k, x_active, x_frozen, parity, net, log0, 2)

    def atomic_forward(self, **kw):
        return self._atom(False, **kw)

    def atomic_backward(self, **kw):
        return self._atom(True, **kw)


class RQSplineCoupling_(Coupling_):
    """Rational-quadratic-spline coupling (couplings_.py:143-275).

    Options (as the reference): xlim, ylim, knots_x, knots_y, extrap, e.g.
    extrap={'left': 'anti', 'right': 'linear'}.  The net must emit 3m-2 channels:
    (m-1) width logits, (m-1) height logits, m derivative logits (softplus, beta=ln2).
    """

    def __init__(self, nets, *, mask, xlim=(0, 1), ylim=(0, 1), knots_x=None, knots_y=None,
                 extrap={}, **kwargs):
        super().__init__(nets, mask=mask, **kwargs)
        self.xlim, self.xwidth = xlim, xlim[1] - xlim[0]
        self.ylim, self.ywidth = ylim, ylim[1] - ylim[0]
        self.knots_x = knots_x
        self.knots_y = knots_y
        self.extrap = extrap

    def _fixed(self, knots, like):
        """A fixed 1-D knot vector as a contiguous device tensor of the field's dtype."""
        if knots is None:
            return None
        k = torch.as_tensor(knots)
        if k.dim() != 1:
            raise NotImplementedError("only 1-D fixed knots_x / knots_y are supported")
        return k.detach().to(device=like.device, dtype=like.dtype).contiguous()

    def _opts(self, n_channels, layout, like):
        """knots_len m from the channel count: 3m-2 (free), 2m-1 (x or y fixed), m (both fixed)
        (couplings_.py:236-256)."""
        kx, ky = self._fixed(self.knots_x, like), self._fixed(self.knots_y, like)
        n_fixed = (kx is not None) + (ky is not None)
        div = 3 - n_fixed
        if (n_channels + 2 - n_fixed) % div:
            raise Exception(f"net output has {n_channels} channels; {div}m-{2 - n_fixed} are needed for m knots")
        m = (n_channels + 2 - n_fixed) // div
        for k in (kx, ky):
            if k is not None and k.numel() != m:
                raise Exception(f"fixed knots have {k.numel()} entries but the net output implies m={m}")
        return _hip.make_rqs_opts(m, self.xlim, self.ylim, self.extrap, layout, kx, ky)

    def _fused_atom(self, inverse, x_active, x_frozen, parity, net, log0):
        """Inference fast path: ConvAct's last layer and the spline in ONE kernel (nf_conv_rqs);
        the logits never reach HBM.  Returns None when it does not apply."""
        if (torch.is_grad_enabled() and (x_active.requires_grad or x_frozen.requires_grad
                                         or any(p.requires_grad for p in net.parameters()))):
            return None
        if (self.propagate_density or self.channels_axis != 1 or self.knots_x is not None
                or self.knots_y is not None or not hasattr(net, 'hidden_and_last')
                or not getattr(self.mask, 'pairable', False) or x_active.dtype not in (torch.float32, torch.float16)
                or not hasattr(self.mask, 'checkerboard_parity')):
            return None
        a = self.mask.checkerboard_parity(parity)
        n_out = net.conv_kwargs['out_channels']
        if a is None or (n_out + 2) % 3:
            return None
        small = self._small3d_atom(inverse, x_active, x_frozen, a, net, log0, n_out)
        if small is None and x_active.dim() == 3:           # 2-D lattices (L1, 16): the same kernel, one plane
            small = self._small_lattice_atom(0, inverse, x_active, x_frozen, parity, net, log0,
                                             _hip.make_rqs_opts((n_out + 2) // 3, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_PAIR))
        if small is not None:
            return small
        # hidden widths 9 .. 16: the stack runs the split-fp16 kernels in groups of 8 channels and hands the logits to the coupling
        # kernel (ConvAct.forward_active -> _hip.conv_wide_logits_split16): twice as fast as the fp32 kernels fused
        if (hasattr(net, '_wide_plan') and x_active.dtype == torch.float32 and x_active.dim() == 5
                and net._wide_plan(self.preprocess_fz(x_frozen[:1])) is not None):
            return None
        # knots_len 4 / 8 / 16 fuse on every kernel; any other knots_len <= 16 only on the split-fp16 chain (nf_conv_h.hip)
        any_kernel = bool(_hip.load().nf_conv_rqs_supported(n_out, (n_out + 2) // 3))
        if not any_kernel and not hasattr(net, '_fuse_plan'):
            return None
        B = x_active.shape[0]
        v = x_active.reshape(B, -1)
        l0 = _hip._log0_tensor(log0, v, B)
        hidden = max(net.conv_kwargs['hidden_sizes'] or [1])
        v = v.contiguous()
        val = torch.empty_like(v)               # the slabs write their rows in place: no concatenation
        lj = torch.empty(B, dtype=torch.float32 if v.dtype == torch.float16 else v.dtype, device=v.device)
        for b0, b1 in self._hidden_slabs(v, hidden):
            xf = x_frozen[b0:b1]
            xin = self.preprocess_fz(xf.float() if xf.dtype == torch.float16 else xf)
            planned = net._fuse_plan(xin) if hasattr(net, '_fuse_plan') else None
            if planned is not None and not any_kernel and not planned[4]:
                return None                    # this knots_len is fused by the split-fp16 chain only, and the stack does not run it
            got = net.hidden_and_last(xin, planned=planned) if planned is not None else net.hidden_and_last(xin)
            if got is None:
                return None
            h, last, unit, split = got
            opts = _hip.make_rqs_opts((n_out + 2) // 3, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_PAIR)
            _hip.conv_rqs(h, last.weight, last.bias, v[b0:b1], None if l0 is None else l0[b0:b1], a,
                          opts, inverse, unit_input=unit, lattice=tuple(x_frozen.shape[1:]) if split else None,
                          out=(val[b0:b1], lj[b0:b1]))
        return val.reshape(x_active.shape), lj

    def _small3d_atom(self, inverse, x_active, x_frozen, a, net, log0, n_out):
        """Small 3-D lattices (L0, L1, 16) that fit a CU's LDS -- BASELINE config 3's 16^3 --: the WHOLE atom (parameter net
        1 -> h -> h -> C and the spline coupling) is one launch of nf_small3d_rqs per slab; nothing but x and y touches HBM."""
        if x_active.dim() != 4 or x_active.dtype != torch.float32 or not hasattr(net, 'small3d_plan'):
            return None
        import ctypes as C
        lat = tuple(x_active.shape[1:])
        plan = net.small3d_plan()
        if plan is None:
            return None
        packed, biases, acts, cout = plan
        m = (n_out + 2) // 3
        if cout != n_out or not _hip.load().nf_small3d_rqs_supported((C.c_int32 * 3)(*lat), cout, m, acts[0], acts[1]):
            return None
        B = x_active.shape[0]
        l0 = _hip._log0_tensor(log0, x_active, B)
        opts = _hip.make_rqs_opts(m, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_PAIR)
        return _hip.small3d_rqs(x_frozen, x_active, packed, biases, l0, a, cout, acts, opts, inverse)

    def make_spline(self, out):
        """The spline the net output `out` (B, C, *L) stands for (couplings_.py:211-262): an `RQSpline` whose knots and
        values come from the coupling kernels' own arithmetic (normflow__amd/lib/spline.py)."""
        return RQSpline(out, xlim=self.xlim, ylim=self.ylim, knots_x=self.knots_x, knots_y=self.knots_y,
                        extrap=self.extrap, knots_axis=self.channels_axis)

    def _hack(self, *, x_active, x_frozen, parity, net):
        """(spline, f(x_active), log g) of one layer with nothing summed (couplings_.py:202-209); the last two are zero off
        the active sublattice.  Inspection only: no autograd."""
        with torch.no_grad():
            out = net(self.preprocess_fz(x_frozen))        # the spline object holds the whole batch's logits: no slabs here
            spline = self.make_spline(out)
            act = self._activity(parity, x_active.shape[1:], x_active.device)
            fx, logg = spline._map(x_active, False, True, True, activity=act, log=True)
        return spline, fx, logg

    def _density_atom(self, inverse, x_active, x_frozen, parity, net, log0):
        """propagate_density (nn/_core.py:19,38-42): log0 + the log-derivative of every site, nothing summed."""
        self._no_grad_only(x_active, x_frozen, net)
        act = self._activity(parity, x_active.shape[1:], x_active.device)

        def one_slab(xa, xf):
            spline = self.make_spline(net(self.preprocess_fz(xf)))
            return spline._map(xa, inverse, True, True, activity=act, log=True)
        val, logg = self._density_slabs(x_active, x_frozen, net, 46, one_slab)
        return val, log0 + logg

    def _train_fused_atom(self, inverse, x_active, x_frozen, parity, net, log0):
        """Training (gradients required): the hidden layers run as autograd's conv nodes, the LAST layer and the spline as
        one differentiable node on the split-fp16 kernel (`_hip.FusedLastRqsFn`): the (B, 3m-2, *L) logits are never
        materialised, forward or backward (Fitter.step, _normflowcore.py:275-294).  None when it does not apply (then: conv
        stack with materialised logits + nf_rqs_fwd / _vjp)."""
        if not torch.is_grad_enabled() or not (x_active.requires_grad or x_frozen.requires_grad
                                               or any(p.requires_grad for p in net.parameters())):
            return None
        if (self.channels_axis != 1 or self.knots_x is not None or self.knots_y is not None
                or not hasattr(net, 'hidden_differentiable') or not getattr(self.mask, 'pairable', False)
                or x_active.dtype != torch.float32 or x_active.dim() != 5 or not hasattr(self.mask, 'checkerboard_parity')):
            return None
        a = self.mask.checkerboard_parity(parity)
        n_out = net.conv_kwargs['out_channels']
        hidden = net.conv_kwargs['hidden_sizes'] or []
        if a is None or (n_out + 2) % 3 or not hidden or hidden[-1] != 8:
            return None
        if x_active.shape[0] * n_out * (x_active[0].numel() // 2) * 4 < TRAIN_FUSED_MIN_LOGIT_BYTES:
            return None                        # small atom: the materialising path is faster (no recomputation)
        import ctypes as C
        lat4 = (C.c_int32 * 4)(*x_active.shape[1:])
        if not _hip.load().nf_conv_rqs_split16_supported(lat4, n_out, (n_out + 2) // 3):
            return None
        got = net.hidden_differentiable(self.preprocess_fz(x_frozen))
        if got is None:
            return None
        h, w_last, b_last = got
        if not _hip.fused_last_rqs_trainable(h, w_last):
            return None
        B = x_active.shape[0]
        v = x_active.reshape(B, -1)
        opts = _hip.make_rqs_opts((n_out + 2) // 3, self.xlim, self.ylim, self.extrap, _hip.LAYOUT_PAIR)
        val, lj = _hip.FusedLastRqsFn.apply(h, w_last, b_last, v, _hip._log0_tensor(log0, v, B), a, opts, inverse)
        return val.reshape(x_active.shape), lj

    def _atom(self, inverse, *, x_active, x_frozen, parity, net, log0=0):
        if self.propagate_density:
            return self._density_atom(inverse, x_active, x_frozen, parity, net, log0)
        fused = self._fused_atom(inverse, x_active, x_frozen, parity, net, log0)
        if fused is not None:
            return fused
        fused = self._train_fused_atom(inverse, x_active, x_frozen, parity, net, log0)
        if fused is not None:
            return fused

        def kernel(v, params, l0, act, layout):
            if v.dtype == torch.float16 and params.dtype != torch.float16:
                params = params.half()      # K2's fp16 storage takes x, logits and y as half (fp32 arithmetic and log-det)
            return _hip.RQSCouplingFn.apply(v, params, l0, act, self._opts(params.shape[1], layout, v), inverse)
        return self._run_atom(kernel, x_active, x_frozen, parity, net, log0, 46)

    def atomic_forward(self, **kw):
        return self._atom(False, **kw)

    def atomic_backward(self, **kw):
        return self._atom(True, **kw)

    def _ctor_kwargs(self):
        return dict(super()._ctor_kwargs(), xlim=self.xlim, ylim=self.ylim, knots_x=self.knots_x,
                    knots_y=self.knots_y, extrap=self.extrap)


class MultiRQSplineCoupling_(Coupling_):
    """`num_splines` RQ splines, one per extra data channel, each with its own limits
    and boundary rule (couplings_.py:279-436).  x: (B, n_s, *L); net: -> (B, n_s*C, *L)."""

    def __init__(self, nets, *, mask, xlims=[(0, 1), (0, 1)], ylims=[(0, 1), (0, 1)],
                 knots_x=[None, None], knots_y=[None, None], extraps=[{}, {}], **kwargs):
        super().__init__(nets, mask=mask, **kwargs)
        self.num_splines = len(xlims)
        self.xlims, self.ylims = xlims, ylims
        self.xwidths = [b - a for a, b in xlims]
        self.ywidths = [b - a for a, b in ylims]
        self.knots_x, self.knots_y = knots_x, knots_y
        self.extraps = extraps

    def preprocess_fz(self, x):
        return x

    def _spline_opts(self, Cs, like):
        """One option block per spline; knots_len from the channels each spline gets (couplings_.py:383-401): 3m-2 free,
        2m-1 with knots_x[i] or knots_y[i] fixed, m with both."""
        opts = []
        for i in range(self.num_splines):
            fix = lambda k: None if k is None else torch.as_tensor(k).detach().to(
                device=like.device, dtype=like.dtype).contiguous()
            kx, ky = fix(self.knots_x[i]), fix(self.knots_y[i])
            for k in (kx, ky):
                if k is not None and k.dim() != 1:
                    raise NotImplementedError("only 1-D fixed knots_x / knots_y are supported")
            n_fixed = (kx is not None) + (ky is not None)
            div = 3 - n_fixed
            if (Cs + 2 - n_fixed) % div:
                raise Exception(f"spline {i} gets {Cs} channels; {div}m-{2 - n_fixed} are needed for m knots")
            m = (Cs + 2 - n_fixed) // div
            for k in (kx, ky):
                if k is not None and k.numel() != m:
                    raise Exception(f"fixed knots of spline {i} have {k.numel()} entries but its channels imply m={m}")
            opts.append(_hip.make_rqs_opts(m, self.xlims[i], self.ylims[i], self.extraps[i], _hip.LAYOUT_FULL, kx, ky))
        return opts

    def _atom(self, inverse, *, x_active, x_frozen, parity, net, log0=0):
        ns = self.num_splines
        out = net(self.preprocess_fz(x_frozen))
        ax = self.channels_axis
        on_axis1 = ax in (1, 1 - out.dim())
        if not on_axis1:                       # the kernels address channels on axis 1
            out, x_active = out.movedim(ax, 1), x_active.movedim(ax, 1)
        B = x_active.shape[0]
        if x_active.shape[1] != ns:
            raise Exception(f"x has {x_active.shape[1]} channels on axis {ax} for {ns} splines")
        lattice = x_active.shape[2:]
        act = self._activity(parity, lattice, x_active.device)
        params = out.reshape(B, out.shape[1], -1)
        Cs = params.shape[1] // ns
        if Cs * ns != params.shape[1]:
            raise Exception(f"net output has {params.shape[1]} channels; need num_splines equal parts")
        opts = self._spline_opts(Cs, x_active)
        v = x_active.reshape(B, ns, -1)
        back = (lambda t: t) if on_axis1 else (lambda t: t.movedim(1, ax))
        if self.propagate_density:
            # log0 + the log-derivative of every site of every spline, nothing summed (nn/_core.py:38-42)
            if torch.is_grad_enabled() and (v.requires_grad or params.requires_grad):
                raise NotImplementedError("propagate_density=True is an inference path here (per-site densities have no "
                                          "VJP kernel); wrap the call in torch.no_grad()")
            val, sites = _hip.multi_rqs_sites(v, params, act, opts, inverse)
            return back(val.reshape(x_active.shape)), log0 + back(sites.reshape(x_active.shape))
        val, lj = _hip.MultiRQSCouplingFn.apply(v, params, _hip._log0_tensor(log0, v, B), act, opts, inverse)
        return back(val.reshape(x_active.shape)), lj

    def atomic_forward(self, **kw):
        return self._atom(False, **kw)

    def atomic_backward(self, **kw):
        return self._atom(True, **kw)

    def _ctor_kwargs(self):
        return dict(super()._ctor_kwargs(), xlims=self.xlims, ylims=self.ylims, knots_x=self.knots_x,
                    knots_y=self.knots_y, extraps=self.extraps)
