"""Parameter-producing networks (reference: src/nn/scalar/modules.py): ConvAct,
LinearAct, SplineNet.  These are the dense contractions of the flow."""
import copy
import math

import torch

from .convNd import Conv4d
from ... import _hip

LN2 = math.log(2.0)


class Abs(torch.nn.Module):
    def forward(self, x):
        return torch.abs(x)


class AvgNeighborPool(torch.nn.Module):
    """Mean of the 2d nearest neighbours (periodic)."""

    def forward(self, x):
        dims = range(1, x.ndim)
        return sum(torch.roll(x, s, d) for d in dims for s in (1, -1)) / (2 * len(dims))


ACTIVATIONS = torch.nn.ModuleDict({
    'tanh': torch.nn.Tanh(), 'relu': torch.nn.ReLU(), 'leaky_relu': torch.nn.LeakyReLU(),
    'softplus': torch.nn.Softplus(), 'avg_neighbor_pool': AvgNeighborPool(), 'abs': Abs(),
    'expit': torch.nn.Sigmoid(), 'none': torch.nn.Identity(),
})


_ACT_OF_MODULE = {torch.nn.Tanh: 1, torch.nn.ReLU: 2, torch.nn.LeakyReLU: 3, torch.nn.Softplus: 4, Abs: 5,
                  torch.nn.Sigmoid: 6, torch.nn.Identity: 0}


class PlusBias(torch.nn.Module):
    def __init__(self, out_features):
        super().__init__()
        self.out_features = out_features
        self.bias = torch.nn.Parameter(torch.randn(out_features))

    def forward(self, x):
        return x + self.bias


class ConvAct(torch.nn.Sequential):
    """A stack of circular 'same' convolutions with activations, lattice dimension 1-4
    (modules.py:68-154).  `acts` has len(hidden_sizes)+1 entries (None = no activation).
    Input (B, C_in, *L) -> output (B, C_out, *L)."""

    Conv = {1: torch.nn.Conv1d, 2: torch.nn.Conv2d, 3: torch.nn.Conv3d, 4: Conv4d}

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, conv_dim: int = 2,
                 hidden_sizes=[], acts=[None], pre_act=None, **extra_kwargs):
        sizes = [in_channels, *hidden_sizes, out_channels]
        assert len(acts) == len(hidden_sizes) + 1
        conv_kwargs = dict(padding='same', padding_mode='circular')
        conv_kwargs.update(extra_kwargs)
        layers = [] if pre_act is None else [ACTIVATIONS[pre_act]]
        for i, act in enumerate(acts):
            layers.append(self.Conv[conv_dim](sizes[i], sizes[i + 1], kernel_size, **conv_kwargs))
            if act is not None:
                layers.append(ACTIVATIONS[act])
        super().__init__(*layers)
        self.conv_kwargs = dict(conv_kwargs, in_channels=in_channels, out_channels=out_channels,
                                kernel_size=kernel_size, conv_dim=conv_dim, hidden_sizes=hidden_sizes,
                                acts=acts, pre_act=pre_act)

    def set_param2zero(self):
        for p in self.parameters():
            torch.nn.init.zeros_(p)

    def transfer(self, scale_factor=1, **extra):
        if scale_factor != 1:
            raise NotImplementedError("kernel rescaling on transfer is not supported")
        return copy.deepcopy(self)

    # ---- fp16 parameter storage (BASELINE config 5): the kernels compute in fp32 on widened copies, cached per version
    def _wb(self, conv):
        w, b = conv.weight, conv.bias
        if w.dtype != torch.float16:
            return w, b
        cache = self.__dict__.setdefault('_w32', {})
        key = id(conv)
        ver = (w._version, w.data_ptr(), None if b is None else b._version)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            hit = (ver, w.detach().float(), None if b is None else b.detach().float())
            cache[key] = hit
        return hit[1], hit[2]

    # ---- fused execution: every (conv, activation) pair is ONE launch of the MFMA kernel
    def _plan(self):
        """[(conv module, act code)] if the whole stack maps onto nf_conv_fwd, else None."""
        mods, plan, i = list(self), [], 0
        while i < len(mods):
            conv = mods[i]
            if not isinstance(conv, (torch.nn.Conv1d, torch.nn.Conv2d, torch.nn.Conv3d, Conv4d)):
                return None
            if not isinstance(conv, Conv4d):
                same = conv.padding == 'same' or all(p == k // 2 for p, k in zip(conv.padding, conv.kernel_size))
                ok = (conv.padding_mode == 'circular' and same and all(k % 2 for k in conv.kernel_size)
                      and all(s == 1 for s in conv.stride) and all(dl == 1 for dl in conv.dilation)
                      and conv.groups == 1)
                if not ok:
                    return None
            act = 0
            if i + 1 < len(mods) and not any(True for _ in mods[i + 1].parameters()):
                code = _ACT_OF_MODULE.get(type(mods[i + 1]))
                if code is None:
                    return None
                act = code
                i += 1
            plan.append((conv, act))
            i += 1
        return plan

    def _run_fused(self, x, compact_parity=None):
        plan = self._plan()
        if plan is not None and plan[0][0].weight.dtype == torch.float16 and x.dtype in (torch.float16, torch.float32):
            x = x.float()                      # half parameters: fp32 compute on widened copies
        if plan is None or not _hip.conv_supported(x, self._wb(plan[0][0])[0]) or x.dim() - 2 != self.conv_kwargs['conv_dim']:
            return None
        for n, (conv, act) in enumerate(plan):
            last = n == len(plan) - 1
            w, b = self._wb(conv)
            x = _hip.conv_layer(x, w, b, act,
                                compact=last and compact_parity is not None,
                                parity=compact_parity or 0)
        return x

    # ---- hidden widths below 8 on the split-fp16 chain: the kernels exchange 8-channel pair tensors, so a narrower stack runs
    # on weights zero-padded to 8 channels (padded hidden channels are act(0) and meet zero weights in the next layer)
    def _pad8(self, conv, w, b, pad_out, pad_in):
        co, ci = w.shape[:2]
        no, ni = (8 if pad_out else co), (8 if pad_in else ci)
        if (no, ni) == (co, ci):
            return w, b
        cache = self.__dict__.setdefault('_w8', {})
        key = (id(conv), pad_out, pad_in)
        ver = (w._version, w.data_ptr(), None if b is None else b._version)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            w8 = w.new_zeros((no, ni) + tuple(w.shape[2:]))
            w8[:co, :ci] = w.detach()
            b8 = None
            if b is not None:
                b8 = b.new_zeros(no)
                b8[:co] = b.detach()
            hit = (ver, w8, b8)
            cache[key] = hit
        return hit[1], hit[2]

    def _fuse_plan(self, x, last_kind='rqs'):
        """How `hidden_and_last` will run this stack on input x: (plan, x as the kernels take it, unit, split, chain), or None
        when the stack does not map onto the MFMA kernels with a last layer without activation.  Planning only: no launch."""
        if self.conv_kwargs.get('pre_act') is not None or x.dim() - 2 != self.conv_kwargs['conv_dim']:
            return None
        plan = self._plan()
        if plan is None or plan[-1][1] != 0:
            return None
        # (weight, bias) as the kernels take them: the parameters themselves, or fp32 copies of half parameters
        wbs = [self._wb(conv) for conv, _ in plan]
        got = self._fuse_plan_for(x, plan, wbs, last_kind)
        hidden = [w.shape[0] for w, _ in wbs[:-1]]
        if ((got is None or not got[4]) and len(plan) > 2 and x.dim() == 6 and hidden and all(h == hidden[0] for h in hidden)
                and 1 <= hidden[0] < 8 and all(w.dim() == 6 and tuple(w.shape[2:]) == (3, 3, 3, 3) for w, _ in wbs)
                and wbs[0][0].shape[1] == 1 and wbs[0][0].dtype == torch.float32):
            # a narrower stack: does it run the split-fp16 chain on weights zero-padded to 8 channels?
            n = len(plan)
            wb8 = [self._pad8(plan[i][0], w, b, pad_out=i < n - 1, pad_in=i > 0) for i, (w, b) in enumerate(wbs)]
            got8 = self._fuse_plan_for(x, plan, wb8, last_kind)
            if got8 is not None and got8[4]:
                return got8
        return got

    def _fuse_plan_for(self, x, plan, wbs, last_kind):
        import types
        plan = [(types.SimpleNamespace(weight=w, bias=b), act) for (w, b), (_, act) in zip(wbs, plan)]
        if plan[0][0].weight.dtype == torch.float32 and x.dtype == torch.float16:
            x = x.float()
        if not _hip.conv_supported(x, plan[0][0].weight):
            return None
        # |hidden| <= 1 when the last hidden activation is tanh or the logistic function: lets the fused kernel use
        # split-fp16 products (nf_conv_h.hip); when it will, the last hidden layer writes its output already split
        # into fp16 (hi, lo) pairs, channel-last (one conversion per site instead of one per halo copy downstream)
        unit = len(plan) > 1 and plan[-2][1] in (_hip.ACT_CODES['tanh'], _hip.ACT_CODES['expit'])
        if last_kind == 'affine':
            lw = plan[-1][0].weight
            split = (unit and len(plan) > 2 and x.dtype == torch.float32 and lw.dim() == 6 and tuple(lw.shape) == (2, 8, 3, 3, 3, 3)
                     and _hip._weights_fit_fp16(lw))
        else:
            split = unit and len(plan) > 2 and self._wants_split16(x, plan)
        # when every hidden layer after the first is an 8 -> 8 layer the split-fp16 two-site kernel takes, the pairs
        # are produced once by the first layer and flow through the whole stack
        chain = split and self._split16_chain(x, plan)
        return plan, x, unit, split, chain

    def hidden_and_last(self, x, last_kind='rqs', planned=None):
        """(hidden activations after all but the last conv, last conv module, |hidden| <= 1?, pair tensor?) when the stack
        maps onto the MFMA kernel and the last layer has no activation; else None.  Lets a coupling fuse the last layer
        with its own kernel: last_kind = 'rqs' (nf_conv_rqs) or 'affine' (nf_conv_affine_split16: only the split-fp16 chain)."""
        got = planned if planned is not None else self._fuse_plan(x, last_kind)
        if got is None:
            return None
        plan, x, unit, split, chain = got
        if last_kind == 'affine' and not chain:
            return None                        # the fused affine layer exists on the pair tensor only
        lat = tuple(x.shape[2:])
        for n, (conv, act) in enumerate(plan[:-1]):
            last_hidden = n == len(plan) - 2
            if chain and n > 0:
                x = _hip.conv_layer_split16(x, conv.weight, conv.bias, act, lat)
            else:
                x = _hip.conv_layer(x, conv.weight, conv.bias, act,
                                    compact=2 if ((chain and n == 0) or (split and last_hidden)) else False)
        return x, plan[-1][0], unit, split

    def small3d_plan(self):
        """The fragment-packed weights of this stack for the small-lattice fused kernel (nf_small3d_rqs /
        nf_small_lattice_coupling: ONE launch per coupling layer, the sample resident in LDS), or None when the stack is not
        1 -> h -> h -> C with 3^3 (or, on 2-D lattices, 3^2) circular kernels, h <= 8, tanh / logistic hidden activations and
        fp16-range weights.  Cached per parameter version."""
        cd = self.conv_kwargs['conv_dim']
        if self.conv_kwargs.get('pre_act') is not None or cd not in (2, 3):
            return None
        plan = self._plan()
        ok_act = (_hip.ACT_CODES['tanh'], _hip.ACT_CODES['expit'])
        if plan is None or len(plan) != 3 or plan[0][1] not in ok_act or plan[1][1] not in ok_act or plan[2][1] != 0:
            return None
        wbs = [self._wb(conv) for conv, _ in plan]
        (w1, b1), (w2, b2), (w3, b3) = wbs
        h = w1.shape[0]
        if (w1.dtype != torch.float32 or not w1.is_cuda or any(tuple(w.shape[2:]) != (3,) * cd for w, _ in wbs) or w1.shape[1] != 1
                or not 1 <= h <= 8 or tuple(w2.shape[:2]) != (h, h) or w3.shape[1] != h or w3.shape[0] > 46):
            return None
        if not all(_hip._weights_fit_fp16(w) for w, _ in wbs):
            return None
        ver = tuple((w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr())) for w, b in wbs)
        hit = self.__dict__.get('_small3d')
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                def pad(w, no, ni):
                    w = w.detach().float()
                    if cd == 2:                # a 3^2 kernel = the middle plane of a 3^3 one
                        w3 = w.new_zeros(tuple(w.shape[:2]) + (3, 3, 3))
                        w3[:, :, 1] = w
                        w = w3
                    return torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, ni - w.shape[1], 0, no - w.shape[0]))
                padb = lambda b, no: None if b is None else torch.nn.functional.pad(b.detach().float(), (0, no - b.shape[0]))
                packed = _hip.pack_small3d_weights(pad(w1, 8, 1), pad(w2, 8, 8), pad(w3, w3.shape[0], 8))
                biases = (padb(b1, 8), padb(b2, 8), padb(b3, w3.shape[0]))
            hit = (ver, packed, biases, (plan[0][1], plan[1][1]), int(w3.shape[0]))
            self.__dict__['_small3d'] = hit
        return hit[1:]

    @staticmethod
    def _split16_chain(x, plan):
        import ctypes as C
        lib = _hip.load()
        lat = list(x.shape[2:])
        lat4 = (C.c_int32 * 4)(*lat)
        first, fact = plan[0]
        if tuple(first.weight.shape[:2]) != (8, 1) or fact not in (_hip.ACT_CODES['tanh'], _hip.ACT_CODES['expit']):
            return False
        if not lib.nf_conv_two_site(8, 0, lat[-1], first.weight.shape[-1]):
            return False
        for conv, act in plan[1:-1]:
            k4 = (C.c_int32 * 4)(*list(conv.weight.shape[2:]))
            if conv.weight.dim() != 6 or not _hip._weights_fit_fp16(conv.weight):
                return False
            if not lib.nf_conv_split16_supported(lat4, k4, conv.weight.shape[1], conv.weight.shape[0], act):
                return False
        return True

    @staticmethod
    def _wants_split16(x, plan):
        import ctypes as C
        last, prev = plan[-1][0], plan[-2][0]
        lat = list(x.shape[2:])
        d = len(lat)
        if x.dtype != torch.float32 or prev.weight.shape[0] != 8 or prev.weight.shape[1] % 4 or d != 4:
            return False
        lib = _hip.load()
        lat4 = (C.c_int32 * 4)(*lat)
        k4 = (C.c_int32 * 4)(*list(last.weight.shape[2:]))
        if not _hip._weights_fit_fp16(last.weight):
            return False
        if lib.nf_conv_weight_layout(lat4, k4, last.weight.shape[1], last.weight.shape[0], 1, 7, _hip.NF_F32) != 2:      # fused | unit input | pair-tensor input
            return False
        kp = list(prev.weight.shape[2:])
        return bool(lib.nf_conv_two_site(8, 0, lat[-1], kp[-1])) and lat[-1] % 4 == 0

    def forward(self, x):
        out = self._run_fused(x) if self.conv_kwargs.get('pre_act') is None else None
        return out if out is not None else super().forward(x)

    def hidden_differentiable(self, x):
        """(hidden activations (B, h, *L) through autograd's ConvFn nodes, last conv's (weight, bias)) for a coupling that
        differentiates its own fused [last layer + map] node (FusedLastRqsFn); None when the stack does not map onto the
        MFMA kernels with a plain last layer."""
        if self.conv_kwargs.get('pre_act') is not None or x.dim() - 2 != self.conv_kwargs['conv_dim']:
            return None
        plan = self._plan()
        if plan is None or len(plan) < 2 or plan[-1][1] != 0 or plan[0][0].weight.dtype != torch.float32:
            return None
        if not _hip.conv_supported(x, plan[0][0].weight):
            return None
        for conv, act in plan[:-1]:
            x = _hip.conv_layer(x, conv.weight, conv.bias, act)
        return x, plan[-1][0].weight, plan[-1][0].bias

    def forward_active(self, x, active_parity):
        """Raw output at the ACTIVE sites only, pair-compact (B, C, V/2): the sites whose
        coordinate sum has parity `active_parity`.  None if the fused path does not apply."""
        if self.conv_kwargs.get('pre_act') is not None or x.shape[-1] % 2:
            return None
        wide = self._wide_plan(x)
        if wide is not None:
            return _hip.conv_wide_logits_split16(x, wide[0], wide[1], active_parity)
        return self._run_fused(x, compact_parity=active_parity)

    def _wide_plan(self, x):
        """(packed weights, first activation) when this stack is 1 -> h -> h -> C with 8 < h <= 16, 3^4 circular kernels, tanh
        hidden activations (the first may be logistic) and fp16-range weights, on a lattice the split-fp16 kernels take, under
        no_grad: `_hip.conv_wide_logits_split16` composes it from those kernels in groups of 8 channels.  Cached per parameter
        version.  None otherwise (the fp32 MFMA kernels run the stack)."""
        import ctypes as C
        if (torch.is_grad_enabled() or x.dim() != 6 or x.shape[1] != 1 or x.dtype != torch.float32 or not x.is_cuda
                or self.conv_kwargs['conv_dim'] != 4):
            return None
        plan = self._plan()
        T, S = _hip.ACT_CODES['tanh'], _hip.ACT_CODES['expit']
        if plan is None or len(plan) != 3 or plan[0][1] not in (T, S) or plan[1][1] != T or plan[2][1] != 0:
            return None
        wbs = [self._wb(conv) for conv, _ in plan]
        (w1, b1), (w2, b2), (w3, b3) = wbs
        h = w1.shape[0]
        if (not 8 < h <= 16 or w1.shape[1] != 1 or tuple(w2.shape[:2]) != (h, h) or w3.shape[1] != h or w3.shape[0] > 46
                or w1.dtype != torch.float32 or any(tuple(w.shape[2:]) != (3, 3, 3, 3) for w, _ in wbs)):
            return None
        lib = _hip.load()
        lat4 = (C.c_int32 * 4)(*x.shape[2:])
        k4 = (C.c_int32 * 4)(3, 3, 3, 3)
        if (not lib.nf_get_option(_hip.OPT_SPLIT16) or not lib.nf_conv_split16_supported(lat4, k4, 8, 8, T)
                or not lib.nf_conv_first_split16_supported(lat4, k4, 8, plan[0][1]) or not lib.nf_conv_rqs_split16_supported(lat4, 46, 16)):
            return None
        if not all(_hip._weights_fit_fp16(w) for w, _ in wbs):
            return None
        ver = tuple((w._version, w.data_ptr(), None if b is None else (b._version, b.data_ptr())) for w, b in wbs)
        hit = self.__dict__.get('_wide16')
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, _hip.pack_wide_split16(w1, b1, w2, b2, w3, b3))
            self.__dict__['_wide16'] = hit
        return hit[1], plan[0][1]


class LinearAct(torch.nn.Sequential):
    """A stack of Linear layers with activations acting on `features_axis`
    (modules.py:197-273)."""

    def __init__(self, in_features: int, out_features: int, hidden_sizes=[], acts=[None], pre_act=None,
                 final_bias=False, features_axis=-1, **linear_kwargs):
        sizes = [in_features, *hidden_sizes, out_features]
        assert len(acts) == len(hidden_sizes) + 1
        layers = [] if pre_act is None else [ACTIVATIONS[pre_act]]
        for i, act in enumerate(acts):
            layers.append(torch.nn.Linear(sizes[i], sizes[i + 1], **linear_kwargs))
            if act is not None:
                layers.append(ACTIVATIONS[act])
        if final_bias:
            layers.append(PlusBias(out_features))
        super().__init__(*layers)
        self.linear_kwargs = dict(linear_kwargs, in_features=in_features, out_features=out_features,
                                  hidden_sizes=hidden_sizes, acts=acts, pre_act=pre_act,
                                  final_bias=final_bias, features_axis=features_axis)

    def forward(self, x):
        ax = self.linear_kwargs['features_axis']
        if ax == -1:
            return super().forward(x)
        return torch.movedim(super().forward(torch.movedim(x, ax, -1)), -1, ax)

    def set_param2zero(self):
        for p in self.parameters():
            torch.nn.init.zeros_(p)


def softplus_ln2(t):
    """log2(1 + 2^t): equals 1 at 0 (modules.py:315)."""
    return torch.nn.functional.softplus(t, beta=LN2, threshold=20.0)


class SplineNet(torch.nn.Module):
    """ONE learned rational-quadratic spline shared by every element of the input
    (modules.py:276-391, spline_shape=[] only -- the reference's class-level
    Softmax(dim=0) makes other shapes ill-defined).

    Parameters (all zero-initialised => identity map): weights_x, weights_y (m-1 logits
    each, softmax -> bin widths / heights), weights_d (m logits, softplus(beta=ln2) ->
    knot derivatives; absent when smooth=True: derivatives from neighbouring slopes,
    spline.py:126-152).
    """

    def __init__(self, knots_len, xlim=(0, 1), ylim=(0, 1), knots_x=None, knots_y=None, knots_d=None,
                 spline_shape=[], knots_axis=-1, smooth=False, Spline=None, label='spline',
                 **spline_kwargs):
        super().__init__()
        if len(spline_shape) > 0:
            raise NotImplementedError("only a single shared spline (spline_shape=[]) is supported")
        need_len = knots_x is None or knots_y is None or knots_d is None
        assert not (need_len and knots_len < 2), "oops: knots_len < 2 for splines"
        self.label = label
        self.knots_len = knots_len
        self.knots_x, self.knots_y, self.knots_d = knots_x, knots_y, knots_d
        self.spline_shape, self.knots_axis = spline_shape, knots_axis
        self.spline_kwargs = spline_kwargs
        self.smooth = smooth
        zeros = lambda n: torch.nn.Parameter(torch.zeros(n))
        if knots_x is None:
            self.xlim, self.xwidth = xlim, xlim[1] - xlim[0]
            self.weights_x = zeros(knots_len - 1)
        if knots_y is None:
            self.ylim, self.ywidth = ylim, ylim[1] - ylim[0]
            self.weights_y = zeros(knots_len - 1)
        if knots_d is None:
            self.weights_d = None if smooth else zeros(knots_len)

    # O(m) host-side (device tensors, differentiable) knot construction; the O(B*V)
    # field pass is the HIP kernel.
    def knots(self):
        """(3, K) tensor of boundary-augmented knots x | y | d."""
        def coords(w, lo, width):
            frac = torch.cumsum(torch.softmax(w, dim=0), dim=0)
            return torch.cat((frac.new_zeros(1), frac)) * width + lo

        kx = self.knots_x if self.knots_x is not None else coords(self.weights_x, self.xlim[0], self.xwidth)
        ky = self.knots_y if self.knots_y is not None else coords(self.weights_y, self.ylim[0], self.ywidth)
        if self.knots_d is not None:
            kd = self.knots_d
        elif self.weights_d is not None:
            kd = softplus_ln2(self.weights_d)
        else:
            slope = (ky[1:] - ky[:-1]) / (kx[1:] - kx[:-1])
            kd = torch.cat((slope[:1], 0.5 * (slope[1:] + slope[:-1]), slope[-1:]))
        extrap = self.spline_kwargs.get('extrap', {}) or {}
        return torch.stack(_augment(kx, ky, kd, extrap.get('left'), extrap.get('right')))

    def _field(self, x, inverse, log0=None):
        v = x.reshape(1, -1) if x.dim() < 2 else x.reshape(x.shape[0], -1)
        val, lj = _hip.DistConvFn.apply(v, self.knots(), log0, _hip.STAGE_SPLINE, inverse)
        return val.reshape(x.shape), lj

    def forward(self, x):
        return self._field(x, False)[0]

    def backward(self, x):
        return self._field(x, True)[0]


def _augment(kx, ky, kd, left, right):
    """Boundary knots of a 1-D knot set (spline.py:458-532): 'linear' adds a tangent-line
    knot one unit outside; 'anti' point-mirrors all other knots through the end knot
    (after any linear knot on the opposite side has been added)."""
    anti = ('anti', 'anti-periodic')
    for side in (left, right):
        if side not in (None, 'linear') + anti:
            raise NotImplementedError(f"extrapolation {side!r} is not supported")
    if left == 'linear':
        kx, ky, kd = (torch.cat((kx[:1] - 1, kx)), torch.cat((ky[:1] - kd[:1], ky)), torch.cat((kd[:1], kd)))
    if right == 'linear':
        kx, ky, kd = (torch.cat((kx, kx[-1:] + 1)), torch.cat((ky, ky[-1:] + kd[-1:])), torch.cat((kd, kd[-1:])))
    if (left == 'linear' or right == 'linear') and (left is None or right is None):
        return kx, ky, kd
    lx = ly = ld = rx = ry = rd = None
    if left in anti:
        lx, ly, ld = 2 * kx[0] - kx[1:].flip(0), 2 * ky[0] - ky[1:].flip(0), kd[1:].flip(0)
    if right in anti:
        rx, ry, rd = 2 * kx[-1] - kx[:-1].flip(0), 2 * ky[-1] - ky[:-1].flip(0), kd[:-1].flip(0)
    join = lambda l, c, r: torch.cat([t for t in (l, c, r) if t is not None])
    return join(lx, kx, rx), join(ly, ky, ry), join(ld, kd, rd)
