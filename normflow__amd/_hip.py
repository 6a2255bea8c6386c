"""ctypes binding of libnormflow_hip.so (the C ABI in include/normflow_hip.h) and the
`torch.autograd.Function`s that put its kernels behind differentiable tensor ops.

There is NO fallback here: if the shared library is missing, or a tensor is not on a
HIP device, the call raises.  PyTorch is used for device memory, the current
stream and autograd bookkeeping only.
"""
import ctypes as C
import os
import threading
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnormflow_hip.so")

NF_F32, NF_F64, NF_F16, NF_F16_FIELD = 0, 1, 2, 3
LAYOUT_FULL, LAYOUT_PAIR = 0, 1
EXTRAP = {None: 0, 'none': 0, 'linear': 1, 'anti': 2, 'anti-periodic': 2}


OPT_SPLIT16, OPT_PIPE, OPT_SMALL8 = 0, 1, 2


class NormflowHipError(RuntimeError):
    pass


class RqsOpts(C.Structure):
    _fields_ = [("m", C.c_int32), ("extrap_left", C.c_int32), ("extrap_right", C.c_int32),
                ("layout", C.c_int32), ("xlo", C.c_double), ("xhi", C.c_double),
                ("ylo", C.c_double), ("yhi", C.c_double),
                ("fixed_knots_x", C.c_void_p), ("fixed_knots_y", C.c_void_p)]


class Strides(C.Structure):
    _fields_ = [("x_batch", C.c_int64), ("y_batch", C.c_int64), ("params_batch", C.c_int64)]


_P, _I64, _I, _SZ, _D = C.c_void_p, C.c_int64, C.c_int, C.c_size_t, C.c_double
_MAP_ARGS = [_P, _P, _P, _P, _P, _P, _I64, _I64, C.POINTER(RqsOpts), C.POINTER(Strides), _P, _SZ, _I, _P]
_VJP_ARGS = [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, C.POINTER(RqsOpts), C.POINTER(Strides), _I, _P]
_SITES_ARGS = [_P, _P, _P, _P, _P, _P, _P, _I, _I64, _I64, C.POINTER(RqsOpts), C.POINTER(Strides), _P, _SZ, _I, _P]
PROTOTYPES = {
    "nf_version": (C.c_int, []),
    "nf_last_error_string": (C.c_char_p, []),
    "nf_set_option": (C.c_int, [C.c_int, C.c_int]),
    "nf_get_option": (C.c_int, [C.c_int]),
    "nf_workspace_bytes": (_SZ, [_I64, _I64]),
    "nf_rqs_fwd": (_I, _MAP_ARGS),
    "nf_rqs_inv": (_I, _MAP_ARGS),
    "nf_rqs_fwd_sites": (_I, _SITES_ARGS),
    "nf_rqs_inv_sites": (_I, _SITES_ARGS),
    "nf_rqs_knots": (_I, [_P, _P, _I64, _I64, C.POINTER(RqsOpts), _I, _P]),
    "nf_spline_eval": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _I, _I, _I, _I, _P]),
    "nf_rqs_fwd_vjp": (_I, _VJP_ARGS),
    "nf_rqs_inv_vjp": (_I, _VJP_ARGS),
    "nf_affine_fwd": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _P, _SZ, _I, _P]),
    "nf_affine_inv": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _P, _SZ, _I, _P]),
    "nf_affine_sites": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _I, _P, _SZ, _I, _P]),
    "nf_affine_vjp": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _I, _I, _I, _P]),
    "nf_distconv": (_I, [_P, _P, _I, _P, _P, _P, _I64, _I64, _I, _I, _P, _SZ, _I, _P]),
    "nf_distconv_vjp": (_I, [_P, _P, _I, _P, _P, _P, _P, _I64, _I64, _I, _I, _P, _SZ, _I, _P]),
    "nf_conv_two_site": (_I, [_I, _I, _I, _I]),
    "nf_conv_cin_pad": (_I, [_I]),
    "nf_conv_ntiles": (_I, [_I]),
    "nf_conv_packed_steps": (_I, [_I, _I]),
    "nf_phi4_action": (_I, [_P, _P, _I64, C.POINTER(C.c_int32), _D, _D, _D, _P, _SZ, _I, _P]),
    "nf_phi4_action_vjp": (_I, [_P, _P, _P, _I64, C.POINTER(C.c_int32), _D, _D, _D, _I, _P]),
    "nf_normal_logprob": (_I, [_P, _P, _P, _P, _I64, _I64, _P, _SZ, _I, _P]),
    "nf_normal_logprob_vjp": (_I, [_P, _P, _P, _P, _P, _I64, _I64, _I, _P]),
    "nf_normal_sample": (_I, [_P, _P, _P, _P, _I64, _I64, C.c_uint64, C.c_uint64, _P, _SZ, _I, _P]),
    "nf_act_vjp": (_I, [_P, _P, _P, _I64, _I, _I, _P]),
    "nf_conv_wgrad_cols": (_I, [_I, _I]),
    "nf_conv_wgrad": (_I, [_P, _P, _P, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I, _P]),
    "nf_planes_to_split16": (_I, [_P, _P, _P, _I64, _I, C.POINTER(C.c_int32), _I, _P]),
    "nf_conv_dgrad_split16": (_I, [_P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _P, _I, _I, _P]),
    "nf_absmax_bits": (_I, [_P, _I64, _P, _P]),
    "nf_conv_last_logits_split16": (_I, [_P, _I, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _P, _P]),
    "nf_conv_last_logits_split16_acc": (_I, [_P, _I, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _P, _I, _I, _P]),
    "nf_expand_pairs": (_I, [_P, _P, _I64, C.POINTER(C.c_int32), _I, _I, _P]),
    "nf_gather_pad": (_I, [_P, _P, _P, _I64, _I64, _I, _P]),
    "nf_conv_wgrad_sites_supported": (_I, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I]),
    "nf_conv_wgrad_sites_workspace": (_SZ, [C.POINTER(C.c_int32), _I, _I, _I]),
    "nf_conv_wgrad_sites": (_I, [_P, _P, _P, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I, _P, _SZ, _I, _P]),
    "nf_conv_wgrad_split16_supported": (_I, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I]),
    "nf_conv_wgrad_split16_workspace": (_SZ, [_I64, C.POINTER(C.c_int32), _I]),
    "nf_conv_wgrad_split16": (_I, [_P, _P, _P, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _P, _I, _P, _SZ, _P]),
    "nf_small3d_rqs_supported": (_I, [C.POINTER(C.c_int32), _I, _I, _I, _I]),
    "nf_small_lattice_supported": (_I, [C.POINTER(C.c_int32), _I, _I, _I, _I, _I, _I]),
    "nf_small_lattice_coupling": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _I, _I, _I, _I,
                                       C.POINTER(RqsOpts), _I, _P]),
    "nf_small3d_rqs": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _I, _I, _I,
                            C.POINTER(RqsOpts), _I, _P]),
    "nf_conv_rqs_split16_train": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _P,
                                       C.POINTER(RqsOpts), _I, _P, _SZ, _P]),
    "nf_conv_rqs_split16_vjp": (_I, [_P, _I, _P, _P, _I, _P, _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), _I, _P,
                                     C.POINTER(RqsOpts), _I, _P]),
    "nf_conv_rqs_supported": (_I, [_I, _I]),
    "nf_conv_rqs_split16_supported": (_I, [C.POINTER(C.c_int32), _I, _I]),
    "nf_conv_last_path": (_I, []),
    "nf_conv_split16_supported": (_I, [_P, _P, _I, _I, _I]),
    "nf_conv_fwd_split16": (_I, [_P, _P, _P, _P, _I64, _P, _I, _P]),
    "nf_conv_affine_split16": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, _P, _I, _I, _I, _P, _SZ, _P]),
    "nf_conv_first_split16_supported": (_I, [_P, _P, _I, _I]),
    "nf_conv_first_split16": (_I, [_P, _P, _P, _P, _I64, _P, _I, _P]),
    "nf_conv_weight_layout": (_I, [_P, _P, _I, _I, _I, _I, _I]),
    "nf_conv_rqs": (_I, [_P, _P, _P, _P, _P, _P, _P, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I,
                         C.POINTER(RqsOpts), _I, _I, _P, _SZ, _I, _P]),
    "nf_conv_fwd": (_I, [_P, _P, _P, _P, _I64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I, _I, _I, _I, _I,
                         _I, _P]),
}

_lib = None
_lock = threading.Lock()


def load():
    """Load the HIP library once; raise loudly if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise NormflowHipError(
                    f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` or `make -C normflow__amd/csrc` (hipcc --offload-arch=gfx950). "
                    "normflow__amd has no CPU or eager-PyTorch fallback for its kernels.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in PROTOTYPES.items():
                fn = getattr(lib, name)
                fn.restype, fn.argtypes = res, args
            _lib = lib
    return _lib


class options:
    """Context manager / setter for the library's kernel-selection options (include/normflow_hip.h, nf_set_option):
    `with _hip.options(split16=False): ...` runs the block with exact fp32 MFMA products everywhere."""
    _CODES = {"split16": OPT_SPLIT16, "pipe": OPT_PIPE, "small8": OPT_SMALL8}

    def __init__(self, **kw):
        self._new = {self._CODES[k]: int(bool(v)) for k, v in kw.items()}
        self._old = {}

    def __enter__(self):
        lib = load()
        for code, val in self._new.items():
            self._old[code] = lib.nf_set_option(code, val)
        return self

    def __exit__(self, *exc):
        lib = load()
        for code, val in self._old.items():
            lib.nf_set_option(code, val)
        return False


def _check(rc, what):
    if rc != 0:
        msg = load().nf_last_error_string().decode("utf-8", "replace")
        raise NormflowHipError(f"{what} failed (code {rc}): {msg}")


def _dtype_code(t):
    if t.dtype == torch.float32:
        return NF_F32
    if t.dtype == torch.float64:
        return NF_F64
    if t.dtype == torch.float16:      # storage only (RQ-spline map kernels): fp32 arithmetic, fp32 log-det
        return NF_F16
    raise TypeError(f"normflow__amd kernels support float32 and float64 (and float16 storage for the spline map), got {t.dtype}")


def _require_device(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise NormflowHipError(
                "normflow__amd kernels need tensors on an MI355X (torch device 'cuda'); got a "
                f"{t.device} tensor and there is no CPU fallback")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _workspace(B, V, device):
    """Scratch buffer for the per-workgroup log-det partials of ONE call, taken from torch's caching allocator:
    stream-aware (two streams never share partials) and graph-pool safe (a buffer captured into a HIP graph belongs
    to the graph's private pool and is never handed out again while the graph lives)."""
    need = load().nf_workspace_bytes(B, V)
    return torch.empty(max(int(need), 256), dtype=torch.uint8, device=device)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _log0_tensor(log0, like, B):
    """log0 may be the python number 0 (nn/_core.py:25 default) or a (B,) tensor."""
    dt = torch.float32 if like.dtype == torch.float16 else like.dtype       # fp16 fields accumulate log|J| in fp32
    if torch.is_tensor(log0):
        if log0.dim() == 0:
            log0 = log0.expand(B)
        return log0.to(dtype=dt, device=like.device).contiguous()
    if log0 == 0:
        return None
    return torch.full((B,), float(log0), dtype=dt, device=like.device)


MAX_B = 32768  # the batch is the grid's y extent; larger batches are cut into slabs


# =============================================================================== RQS
def make_rqs_opts(m, xlim, ylim, extrap, layout, knots_x=None, knots_y=None):
    """knots_x / knots_y: optional contiguous 1-D device tensors of m fixed knot coordinates (of
    the call's dtype); the limits are then the end knots."""
    extrap = extrap or {}
    for side in ('left', 'right'):
        if extrap.get(side) not in EXTRAP:
            raise NotImplementedError(f"extrapolation {extrap.get(side)!r} is not supported "
                                      "(supported: None, 'linear', 'anti')")
    if knots_x is not None:
        xlim = (float(knots_x[0]), float(knots_x[-1]))
    if knots_y is not None:
        ylim = (float(knots_y[0]), float(knots_y[-1]))
    opts = RqsOpts(int(m), EXTRAP[extrap.get('left')], EXTRAP[extrap.get('right')], int(layout),
                   float(xlim[0]), float(xlim[1]), float(ylim[0]), float(ylim[1]),
                   knots_x.data_ptr() if knots_x is not None else None,
                   knots_y.data_ptr() if knots_y is not None else None)
    opts._keepalive = (knots_x, knots_y)
    return opts


def _rqs_call(fn_name, v, params, mask, log0, opts, strides, B, V):
    lib = load()
    out = torch.empty_like(v)
    logj = torch.empty(B, dtype=torch.float32 if v.dtype == torch.float16 else v.dtype, device=v.device)
    if log0 is not None and log0.dtype != logj.dtype:
        raise TypeError(f"log0 must be {logj.dtype} for a {v.dtype} field")
    ws = _workspace(min(B, MAX_B), V, v.device)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        l0 = log0[b0:b1] if log0 is not None else None
        _check(getattr(lib, fn_name)(_ptr(v[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(l0),
                                      _ptr(out[b0:b1]), _ptr(logj[b0:b1]), b1 - b0, V, C.byref(opts),
                                      C.byref(strides) if strides is not None else None, _ptr(ws),
                                      ws.numel(), _dtype_code(v), _stream()), fn_name)
    return out, logj


SITES_LOG, SITES_DERIVATIVE = 1, 2       # nf_sites_mode


def rqs_sites(v, params, mask, log0, opts, inverse, mode=SITES_LOG):
    """nf_rqs_fwd_sites / nf_rqs_inv_sites: (value, logJ, per-site log-derivative or derivative).  Inference only
    (no autograd): the spline-object / propagate_density path of the reference (couplings_.py:202-209, _core.py:38-42)."""
    _require_device(v, params, mask, log0)
    if v.dtype not in (torch.float32, torch.float64) or params.dtype != v.dtype:
        raise TypeError(f"per-site derivatives are built for float32 / float64 fields; got {v.dtype} / {params.dtype}")
    lib = load()
    B, V = v.shape
    v, params = v.detach().contiguous(), params.detach().contiguous()
    out, sites = torch.empty_like(v), torch.empty_like(v)
    logj = torch.empty(B, dtype=v.dtype, device=v.device)
    ws = _workspace(min(B, MAX_B), V, v.device)
    fn = lib.nf_rqs_inv_sites if inverse else lib.nf_rqs_fwd_sites
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        l0 = log0[b0:b1] if log0 is not None else None
        _check(fn(_ptr(v[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(l0), _ptr(out[b0:b1]), _ptr(logj[b0:b1]),
                  _ptr(sites[b0:b1]), int(mode), b1 - b0, V, C.byref(opts), None, _ptr(ws), ws.numel(),
                  _dtype_code(v), _stream()), "nf_rqs_sites")
    return out, logj, sites


def multi_rqs_sites(v, params, mask, opts_list, inverse, mode=SITES_LOG):
    """Per-site values and log-derivatives (or derivatives) of `n_s` splines, one per data channel
    (MultiRQSplineCoupling_ with propagate_density, couplings_.py:304-329 + nn/_core.py:38-42).
    v: (B, n_s, V), params: (B, n_s*C, V) -> (value (B, n_s, V), sites (B, n_s, V)).  Inference only."""
    _require_device(v, params, mask)
    if v.dtype not in (torch.float32, torch.float64) or params.dtype != v.dtype:
        raise TypeError(f"per-site derivatives are built for float32 / float64 fields; got {v.dtype} / {params.dtype}")
    lib = load()
    B, ns, V = v.shape
    v, params = v.detach().contiguous(), params.detach().contiguous()
    Ctot, Vp = params.shape[1], params.shape[2]
    Cs = Ctot // ns
    out = torch.empty_like(v)
    sites = torch.empty(ns, B, V, dtype=v.dtype, device=v.device)
    logj = torch.empty(B, dtype=v.dtype, device=v.device)
    ws = _workspace(min(B, MAX_B), V, v.device)
    fn = lib.nf_rqs_inv_sites if inverse else lib.nf_rqs_fwd_sites
    esz = v.element_size()
    st = Strides(ns * V, ns * V, Ctot * Vp)
    for i, opts in enumerate(opts_list):
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(fn(C.c_void_p(v[b0:b1].data_ptr() + i * V * esz),
                      C.c_void_p(params[b0:b1].data_ptr() + i * Cs * Vp * esz), _ptr(mask), None,
                      C.c_void_p(out[b0:b1].data_ptr() + i * V * esz), _ptr(logj[b0:b1]), _ptr(sites[i, b0:b1]),
                      int(mode), b1 - b0, V, C.byref(opts), C.byref(st), _ptr(ws), ws.numel(), _dtype_code(v),
                      _stream()), "nf_rqs_sites (multi)")
    return out, sites.movedim(0, 1)


def spline_eval(v, kx, ky, kd, K, shared, inverse, grad):
    """nf_spline_eval: a spline given by explicit knots at every site.  v: (B, V); each knot tensor (K, B*V) planes... laid
    out (K, V) per batch row when B == 1, or a shared (K,) vector (`shared` = three flags).  Returns (value, derivative|None)."""
    _require_device(v, kx, ky, kd)
    B, V = v.shape
    if B != 1 and not all(shared):
        raise NotImplementedError("per-site explicit knots are addressed as one (K, V) block: pass B == 1")
    out = torch.empty_like(v)
    der = torch.empty_like(v) if grad else None
    _check(load().nf_spline_eval(_ptr(v), _ptr(kx), _ptr(ky), _ptr(kd), _ptr(out), _ptr(der), B, V, int(K),
                                 int(shared[0]), int(shared[1]), int(shared[2]), int(bool(inverse)), _dtype_code(v),
                                 _stream()), "nf_spline_eval")
    return out, der


def rqs_knots(params, opts):
    """nf_rqs_knots: (B, C, V) logits -> (B, 3m, V) = knots_x | knots_y | knots_d, before boundary augmentation."""
    _require_device(params)
    if params.dtype not in (torch.float32, torch.float64):
        raise TypeError(f"knots are built for float32 / float64 logits; got {params.dtype}")
    params = params.detach().contiguous()
    B, _, V = params.shape
    knots = torch.empty(B, 3 * opts.m, V, dtype=params.dtype, device=params.device)
    for b0 in range(0, B, 65535):
        b1 = min(B, b0 + 65535)
        _check(load().nf_rqs_knots(_ptr(params[b0:b1]), _ptr(knots[b0:b1]), b1 - b0, V, C.byref(opts),
                                   _dtype_code(params), _stream()), "nf_rqs_knots")
    return knots


def _rqs_vjp_call(fn_name, x, params, mask, gout, glogj, opts, strides, B, V):
    lib = load()
    gin = torch.empty_like(x)
    gpar = torch.empty_like(params)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        _check(getattr(lib, fn_name)(_ptr(x[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(gout[b0:b1]),
                                      _ptr(glogj[b0:b1]), _ptr(gin[b0:b1]), _ptr(gpar[b0:b1]), b1 - b0, V,
                                      C.byref(opts), C.byref(strides) if strides is not None else None,
                                      _dtype_code(x), _stream()), fn_name)
    return gin, gpar


class RQSCouplingFn(torch.autograd.Function):
    """(value, logJ) of one RQ-spline coupling layer on the active sublattice.

    v: (B, V) field (only active sites are read); params: (B, C, V) or (B, C, V/2) raw
    logits; mask: (V,) uint8 activity; log0: (B,) tensor or None.
    """

    @staticmethod
    def forward(ctx, v, params, log0, mask, opts, inverse):
        _require_device(v, params, mask, log0)
        B, V = v.shape
        v, params = v.contiguous(), params.contiguous()
        if params.dtype != v.dtype:
            raise TypeError(f"field is {v.dtype} but net output is {params.dtype}")
        out, logj = _rqs_call("nf_rqs_inv" if inverse else "nf_rqs_fwd", v, params, mask, log0, opts,
                              None, B, V)
        ctx.save_for_backward(out if inverse else v, params, mask)
        ctx.opts, ctx.inverse, ctx.has_log0 = opts, inverse, log0 is not None
        return out, logj

    @staticmethod
    def backward(ctx, gout, glogj):
        x, params, mask = ctx.saved_tensors
        if x.dtype == torch.float16:
            raise NotImplementedError("fp16 storage is an inference path (nf_rqs_fwd / nf_rqs_inv); train in fp32")
        B, V = x.shape
        gin, gpar = _rqs_vjp_call("nf_rqs_inv_vjp" if ctx.inverse else "nf_rqs_fwd_vjp", x, params, mask,
                                  gout.contiguous(), glogj.contiguous(), ctx.opts, None, B, V)
        return gin, gpar, (glogj if ctx.has_log0 else None), None, None, None


class MultiRQSCouplingFn(torch.autograd.Function):
    """`n_s` independent splines, one per data channel, addressed in place through batch
    strides (MultiRQSplineCoupling_).  v: (B, n_s, V); params: (B, n_s*C, Vp)."""

    @staticmethod
    def forward(ctx, v, params, log0, mask, opts_list, inverse):
        _require_device(v, params, mask, log0)
        B, ns, V = v.shape
        v, params = v.contiguous(), params.contiguous()
        if params.dtype != v.dtype:        # the channels are addressed by byte offsets: a silent reinterpretation otherwise
            raise TypeError(f"field is {v.dtype} but net output is {params.dtype}")
        if log0 is not None and log0.dtype != v.dtype:
            raise TypeError(f"log0 must be {v.dtype} for a {v.dtype} field")
        Ctot, Vp = params.shape[1], params.shape[2]
        Cs = Ctot // ns
        out = torch.empty_like(v)
        lib = load()
        fn = lib.nf_rqs_inv if inverse else lib.nf_rqs_fwd
        ws = _workspace(min(B, MAX_B), V, v.device)
        esz = v.element_size()
        st = Strides(ns * V, ns * V, Ctot * Vp)
        logj = log0
        for i, opts in enumerate(opts_list):
            nxt = torch.empty(B, dtype=v.dtype, device=v.device)
            for b0 in range(0, B, MAX_B):       # the batch is the grid's y extent
                b1 = min(B, b0 + MAX_B)
                l0 = logj[b0:b1] if logj is not None else None
                _check(fn(C.c_void_p(v[b0:b1].data_ptr() + i * V * esz),
                          C.c_void_p(params[b0:b1].data_ptr() + i * Cs * Vp * esz), _ptr(mask), _ptr(l0),
                          C.c_void_p(out[b0:b1].data_ptr() + i * V * esz), _ptr(nxt[b0:b1]), b1 - b0, V,
                          C.byref(opts), C.byref(st), _ptr(ws), ws.numel(), _dtype_code(v), _stream()),
                       "nf_rqs (multi)")
            logj = nxt
        ctx.save_for_backward(out if inverse else v, params, mask)
        ctx.opts_list, ctx.inverse, ctx.has_log0 = opts_list, inverse, log0 is not None
        return out, logj

    @staticmethod
    def backward(ctx, gout, glogj):
        x, params, mask = ctx.saved_tensors
        B, ns, V = x.shape
        Ctot, Vp = params.shape[1], params.shape[2]
        Cs = Ctot // ns
        gout, glogj = gout.contiguous(), glogj.contiguous()
        gin, gpar = torch.empty_like(x), torch.empty_like(params)
        lib = load()
        fn = lib.nf_rqs_inv_vjp if ctx.inverse else lib.nf_rqs_fwd_vjp
        esz = x.element_size()
        st = Strides(ns * V, ns * V, Ctot * Vp)
        for i, opts in enumerate(ctx.opts_list):
            xo, po = i * V * esz, i * Cs * Vp * esz
            for b0 in range(0, B, MAX_B):
                b1 = min(B, b0 + MAX_B)
                _check(fn(C.c_void_p(x[b0:b1].data_ptr() + xo), C.c_void_p(params[b0:b1].data_ptr() + po), _ptr(mask),
                          C.c_void_p(gout[b0:b1].data_ptr() + xo), _ptr(glogj[b0:b1]),
                          C.c_void_p(gin[b0:b1].data_ptr() + xo), C.c_void_p(gpar[b0:b1].data_ptr() + po), b1 - b0, V,
                          C.byref(opts), C.byref(st), _dtype_code(x), _stream()), "nf_rqs_vjp (multi)")
        return gin, gpar, (glogj if ctx.has_log0 else None), None, None, None


# ============================================================================ affine
def affine_sites(v, params, mask, log0, layout, inverse):
    """nf_affine_sites: (value, logJ, per-site log-derivative) of an affine / shift layer; inference only."""
    _require_device(v, params, mask, log0)
    if v.dtype not in (torch.float32, torch.float64) or params.dtype != v.dtype:
        raise TypeError(f"per-site densities are built for float32 / float64 fields; got {v.dtype} / {params.dtype}")
    B, V = v.shape
    v, params = v.detach().contiguous(), params.detach().contiguous()
    out, sites = torch.empty_like(v), torch.empty_like(v)
    logj = torch.empty(B, dtype=v.dtype, device=v.device)
    ws = _workspace(min(B, MAX_B), V, v.device)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        l0 = log0[b0:b1] if log0 is not None else None
        _check(load().nf_affine_sites(_ptr(v[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(l0), _ptr(out[b0:b1]),
                                      _ptr(logj[b0:b1]), _ptr(sites[b0:b1]), b1 - b0, V, params.shape[1], layout,
                                      int(bool(inverse)), _ptr(ws), ws.numel(), _dtype_code(v), _stream()),
               "nf_affine_sites")
    return out, logj, sites


class AffineCouplingFn(torch.autograd.Function):
    """Affine (n_ch = 2) or shift (n_ch = 1) coupling on the active sublattice."""

    @staticmethod
    def forward(ctx, v, params, log0, mask, layout, inverse):
        _require_device(v, params, mask, log0)
        B, V = v.shape
        v, params = v.contiguous(), params.contiguous()
        if v.dtype == torch.float16:
            # fp16 field storage (BASELINE config 5): params either half as well (NF_F16) or the fp32 a conv layer wrote
            # (NF_F16_FIELD); fp32 arithmetic, fp32 log-det
            if params.dtype == torch.float16:
                code = NF_F16
            elif params.dtype == torch.float32:
                code = NF_F16_FIELD
            else:
                raise TypeError(f"half field with {params.dtype} parameters")
            ldt = torch.float32
        else:
            if params.dtype != v.dtype:
                raise TypeError(f"field is {v.dtype} but net output is {params.dtype}")
            code, ldt = _dtype_code(v), v.dtype
        if log0 is not None and log0.dtype != ldt:
            raise TypeError(f"log0 must be {ldt} for a {v.dtype} field")
        n_ch = params.shape[1]
        lib = load()
        out = torch.empty_like(v)
        logj = torch.empty(B, dtype=ldt, device=v.device)
        ws = _workspace(min(B, MAX_B), V, v.device)
        fn = lib.nf_affine_inv if inverse else lib.nf_affine_fwd
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            l0 = log0[b0:b1] if log0 is not None else None
            _check(fn(_ptr(v[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(l0), _ptr(out[b0:b1]),
                      _ptr(logj[b0:b1]), b1 - b0, V, n_ch, layout, _ptr(ws), ws.numel(), code,
                      _stream()), "nf_affine")
        ctx.save_for_backward(v, params, mask)
        ctx.layout, ctx.inverse, ctx.has_log0 = layout, inverse, log0 is not None
        return out, logj

    @staticmethod
    def backward(ctx, gout, glogj):
        v, params, mask = ctx.saved_tensors
        if v.dtype == torch.float16:
            raise NotImplementedError("fp16 storage is an inference path (nf_affine_fwd / nf_affine_inv); train in fp32")
        B, V = v.shape
        gout, glogj = gout.contiguous(), glogj.contiguous()
        gin, gpar = torch.empty_like(v), torch.empty_like(params)
        lib = load()
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(lib.nf_affine_vjp(_ptr(v[b0:b1]), _ptr(params[b0:b1]), _ptr(mask), _ptr(gout[b0:b1]),
                                     _ptr(glogj[b0:b1]), _ptr(gin[b0:b1]), _ptr(gpar[b0:b1]), b1 - b0, V,
                                     params.shape[1], ctx.layout, int(ctx.inverse), _dtype_code(v),
                                     _stream()), "nf_affine_vjp")
        return gin, gpar, (glogj if ctx.has_log0 else None), None, None, None


# ========================================================================== distconv
STAGE_EXPIT, STAGE_SPLINE, STAGE_LOGIT = 1, 2, 4


class DistConvFn(torch.autograd.Function):
    """Expit_ -> shared-knot spline -> Logit_ (any subset, `stages` bit mask) in one pass.

    v: (B, V); knots: (3, K) augmented knots (x | y | d), differentiable.
    """

    @staticmethod
    def forward(ctx, v, knots, log0, stages, inverse):
        _require_device(v, knots, log0)
        B, V = v.shape
        v = v.contiguous()
        if knots is not None:
            knots = knots.to(v.dtype).contiguous()
        K = knots.shape[1] if knots is not None else 0
        lib = load()
        out = torch.empty_like(v)
        logj = torch.empty(B, dtype=v.dtype, device=v.device)
        ws = _workspace(min(B, MAX_B), V, v.device)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            l0 = log0[b0:b1] if log0 is not None else None
            _check(lib.nf_distconv(_ptr(v[b0:b1]), _ptr(knots), K, _ptr(l0), _ptr(out[b0:b1]),
                                   _ptr(logj[b0:b1]), b1 - b0, V, stages, int(inverse), _ptr(ws),
                                   ws.numel(), _dtype_code(v), _stream()), "nf_distconv")
        ctx.save_for_backward(out if inverse else v, knots)
        ctx.stages, ctx.inverse, ctx.has_log0 = stages, inverse, log0 is not None
        return out, logj

    @staticmethod
    def backward(ctx, gout, glogj):
        x, knots = ctx.saved_tensors
        B, V = x.shape
        K = knots.shape[1] if knots is not None else 0
        gout, glogj = gout.contiguous(), glogj.contiguous()
        gin = torch.empty_like(x)
        gk = torch.zeros(3, max(K, 1), dtype=torch.float64, device=x.device)
        lib = load()
        ws = _workspace(min(B, MAX_B), V, x.device)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            part = torch.zeros(3, max(K, 1), dtype=torch.float64, device=x.device)
            _check(lib.nf_distconv_vjp(_ptr(x[b0:b1]), _ptr(knots), K, _ptr(gout[b0:b1]), _ptr(glogj[b0:b1]),
                                       _ptr(gin[b0:b1]), _ptr(part), b1 - b0, V, ctx.stages,
                                       int(ctx.inverse), _ptr(ws), ws.numel(), _dtype_code(x), _stream()),
                   "nf_distconv_vjp")
            gk += part
        gk = gk.to(knots.dtype) if knots is not None else None
        return gin, gk, (glogj if ctx.has_log0 else None), None, None


# ============================================================================== conv
ACT_CODES = {None: 0, 'none': 0, 'tanh': 1, 'relu': 2, 'leaky_relu': 3, 'softplus': 4, 'abs': 5, 'expit': 6}
_TORCH_ACT = {0: lambda t: t, 1: torch.tanh, 2: torch.relu,
              3: lambda t: torch.nn.functional.leaky_relu(t), 4: lambda t: torch.nn.functional.softplus(t),
              5: torch.abs, 6: torch.sigmoid}


def pack_conv_weight(w):
    """(Cout, Cin, *k) -> MFMA fragment order [tap][cin/4][cout/16][4][16] (zero padded)."""
    cout, cin = w.shape[:2]
    ntaps = 1
    for k in w.shape[2:]:
        ntaps *= k
    nt = (cout + 15) // 16
    ns = load().nf_conv_packed_steps(cin, ntaps)
    if ns:      # cin % 4 != 0: K = (tap, ci) flattened, 4 per step
        wk = w.new_zeros(nt * 16, 4 * ns)
        wk[:cout, :cin * ntaps] = w.reshape(cout, cin, ntaps).permute(0, 2, 1).reshape(cout, ntaps * cin)
        return wk.reshape(nt, 16, ns, 4).permute(2, 0, 3, 1).contiguous()
    cin_pad = (cin + 3) // 4 * 4
    wp = w.new_zeros(nt * 16, cin_pad, ntaps)
    wp[:cout, :cin] = w.reshape(cout, cin, ntaps)
    return wp.reshape(nt, 16, cin_pad // 4, 4, ntaps).permute(4, 2, 0, 3, 1).contiguous()


def conv_weight_for_layer(w, lat4, k4, cin, cout, compact, fused, dtype_code):
    """Pack (Cout', Cin, *k) weights (already two-site expanded where that applies) in the layout the
    library will read for this layer (nf_conv_weight_layout): fragment order, or row-packed for the
    persistent kernel -- [row][cin/4][lane][K3*ntiles -> multiple of 4]."""
    frag = pack_conv_weight(w)
    code = load().nf_conv_weight_layout(lat4, k4, cin, cout, int(compact), int(fused), dtype_code)
    if code < 0:
        raise RuntimeError("nf_conv_weight_layout: invalid layer description")
    if code == 0:
        return frag
    if code == 2:
        return pack_conv_weight_split16(w)
    ntaps, kq, nt = frag.shape[:3]
    k3 = w.shape[-1]
    rows = ntaps // k3
    vals = frag.reshape(rows, k3, kq, nt, 64).permute(0, 2, 4, 1, 3).reshape(rows, kq, 64, k3 * nt)
    nv = (k3 * nt + 3) // 4 * 4
    out = vals.new_zeros(rows, kq, 64, nv)
    out[..., :k3 * nt] = vals
    return out.contiguous()


# The weights are scaled by 2^10 before the split (the kernel scales the accumulators back, exactly): typical |w| ~ 0.1
# would put the lo parts into fp16's subnormal range -- absolute resolution 6e-8, i.e. 5e-7 RELATIVE to w and, unlike
# activation rounding, the same error at every site: it showed up as a coherent 1.7e-5 shift of log|J| on 32^4.
SPLIT16_WEIGHT_SCALE = 1024.0


def pack_conv_weight_split16(w):
    """(46, 8, 3, 3, 3, 3) fp32 weights -> NF_WLAYOUT_SPLIT16 (include/normflow_hip.h): fp16 hi / lo pairs in the
    B-fragment order of v_mfma_f32_16x16x32_f16, [column tile (3)][K slice (21)][hi|lo][lane (64)][8]:
    slice 7*j3 + i = tap j3 of kernel rows 4i..4i+3; lane 16*g + n holds the 8 input channels of row 4i+g for
    column 16*tile + n (zero for row 27 and for columns >= cout)."""
    cout, cin = w.shape[:2]
    assert cin == 8 and tuple(w.shape[2:]) == (3, 3, 3, 3) and cout <= 48
    wp = w.new_zeros(48, 8, 28, 3, dtype=torch.float32)
    wp[:cout, :, :27] = w.reshape(cout, 8, 27, 3).float() * SPLIT16_WEIGHT_SCALE
    hi = wp.half()
    lo = (wp - hi.float()).half()
    out = torch.empty(3, 21, 2, 64, 8, dtype=torch.float16, device=w.device)
    for k, part in enumerate((hi, lo)):
        # part[co = 16 t + n, ch, row = 4 i + g, j3]  ->  [t, j3, i, g, n, ch]
        v = part.reshape(3, 16, 8, 7, 4, 3).permute(0, 5, 3, 4, 1, 2)       # t, j3, i, g, n, ch
        out[:, :, k] = v.reshape(3, 21, 64, 8)
    return out.contiguous()


def pack_conv_weight_split16_two_site(w):
    """(8, 8, 3, 3, 3, 3) fp32 weights of a hidden layer -> the B fragments of conv_g_kernel (include/normflow_hip.h,
    nf_conv_fwd_split16): [kernel row (27)][hi|lo][lane (64)][8]; lane 16*g + n: column n = 8*shift + co, tap g - shift."""
    assert tuple(w.shape) == (8, 8, 3, 3, 3, 3)
    wr = w.reshape(8, 8, 27, 3).float() * SPLIT16_WEIGHT_SCALE          # co, ch, row, j3
    w2 = wr.new_zeros(16, 8, 27, 4)                                     # column, ch, row, tap g
    w2[:8, :, :, :3] = wr
    w2[8:, :, :, 1:] = wr
    hi = w2.half()
    lo = (w2 - hi.float()).half()
    out = torch.empty(27, 2, 64, 8, dtype=torch.float16, device=w.device)
    for k, part in enumerate((hi, lo)):
        out[:, k] = part.permute(2, 3, 0, 1).reshape(27, 64, 8)         # row, (g, n), ch
    return out.contiguous()


def pack_conv_weight_split16_first(w):
    """(8, 1, 3, 3, 3, 3) fp32 weights of a first ConvAct layer -> the B fragments of conv_c2_kernel (include/normflow_hip.h,
    nf_conv_first_split16): [K slice (4)][hi|lo][lane (64)][8]; K index = 4 r + t with r the kernel row (j0, j1, j2) row-major
    (27, zero-padded to 32) and t the tap of the site pair; lane 16*g + n (column n = 8*shift + co) holds rows 8*slice + 2*g + h
    (h = 0, 1), taps t = 0..3, as value 4*h + t = W[co][r][t - shift] (zero outside 0..2), scaled by 2^10 and split."""
    assert tuple(w.shape) == (8, 1, 3, 3, 3, 3)
    wr = w.reshape(8, 27, 3).float() * SPLIT16_WEIGHT_SCALE              # co, row, j3
    w2 = wr.new_zeros(16, 32, 4)                                         # column, row (padded), tap
    w2[:8, :27, :3] = wr
    w2[8:, :27, 1:] = wr
    hi = w2.half()
    lo = (w2 - hi.float()).half()
    out = torch.empty(4, 2, 64, 8, dtype=torch.float16, device=w.device)
    for k, part in enumerate((hi, lo)):
        # part[n, r = 8 sl + 2 g + h, t] -> [sl, g, n, h, t]
        v = part.reshape(16, 4, 4, 2, 4).permute(1, 2, 0, 3, 4)          # sl, g, n, h, t
        out[:, k] = v.reshape(4, 64, 8)
    return out.contiguous()


def conv_first_split16(x, weight, bias, act, weight_src=None):
    """First ConvAct layer 1 -> 8 on the split-fp16 kernel (nf_conv_first_split16): x (B, 1, *L) fp32 -> the fp16 pair
    tensor (B, V, 16); inference only."""
    lib = load()
    B = x.shape[0]
    lat = list(x.shape[2:])
    lat4 = (C.c_int32 * 4)(*lat)
    V = 1
    for n in lat:
        V *= n
    wsp = _cached_pack(weight if weight_src is None else weight_src, 'first16', pack_conv_weight_split16_first)
    bias = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((B, V, 16), dtype=torch.float16, device=x.device)
    step = max(1, min(MAX_B, ((1 << 31) - 1) // V))
    for b0 in range(0, B, step):
        b1 = min(B, b0 + step)
        _check(lib.nf_conv_first_split16(_ptr(x[b0:b1]), _ptr(wsp), _ptr(bias), _ptr(out[b0:b1]), b1 - b0, lat4,
                                         int(act), _stream()), "nf_conv_first_split16")
    return out


def _split16_site_index(L3, device):
    """idx[par, slot] = the site x3 of a lattice row stored in slot `slot` of parity block `par` of the fp16 pair layout
    (include/normflow_hip.h, NF_OUT_SPLIT16): even sites in order, odd sites rotated by one slot (slot s holds site 2s-1)."""
    slot = torch.arange(L3 // 2, device=device)
    return torch.stack((2 * slot, (2 * slot - 1) % L3))


def to_split16(h):
    """(B, 8, *L) fp32 hidden activations (|h| <= 1) -> the fp16 pair tensor the split-fp16 kernels exchange, shape
    (B, V, 16) halfs = per lattice row (fastest axis, L3 sites) [hi | lo][even sites | odd sites][L3/2 slots][8 channels]
    with hi = fp16(h), lo = fp16(h - hi) (layout: include/normflow_hip.h, NF_OUT_SPLIT16).  Host-side helper for tests and
    benches: in the pipeline the producing kernel's epilogue writes this format itself."""
    B, Cc = h.shape[:2]
    L3 = h.shape[-1]
    if Cc != 8 or L3 % 2:
        raise NormflowHipError("to_split16: 8 channels and an even fastest axis expected")
    hp = h.reshape(B, 8, -1, L3).float()
    idx = _split16_site_index(L3, h.device)                           # (2, L3/2)
    g = hp[..., idx].permute(0, 2, 3, 4, 1)                           # (B, R, par, slot, channel)
    hi = g.half()
    lo = (g - hi.float()).half()
    return torch.stack((hi, lo), dim=2).reshape(B, -1, 16).contiguous()


def from_split16(h16, lattice):
    """Inverse of to_split16 up to the split's rounding: (B, V, 16) halfs -> (B, 8, *L) fp32 (hi + lo)."""
    B = h16.shape[0]
    L3 = lattice[-1]
    t = h16.reshape(B, -1, 2, 2, L3 // 2, 8).float()
    v = t[:, :, 0] + t[:, :, 1]                                       # (B, R, par, slot, channel)
    idx = _split16_site_index(L3, h16.device)
    out = torch.empty((B, v.shape[1], L3, 8), dtype=torch.float32, device=h16.device)
    out[:, :, idx.reshape(-1)] = v.reshape(B, v.shape[1], L3, 8)
    return out.permute(0, 3, 1, 2).reshape((B, 8) + tuple(lattice)).contiguous()


def conv_layer_split16(h16, weight, bias, act, lattice):
    """Hidden 8 -> 8 layer on fp16 (hi, lo) pairs in and out (nf_conv_fwd_split16); inference only."""
    lib = load()
    B = h16.shape[0]
    lat4 = (C.c_int32 * 4)(*lattice)
    wsp = _cached_pack(weight, 'two_site16', pack_conv_weight_split16_two_site)
    bias = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty_like(h16)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        _check(lib.nf_conv_fwd_split16(_ptr(h16[b0:b1]), _ptr(wsp), _ptr(bias), _ptr(out[b0:b1]), b1 - b0, lat4,
                                       int(act), _stream()), "nf_conv_fwd_split16")
    return out


_UNIT_OK = {}


def invalidate_weight_checks():
    """Forget every cached verdict of `_weights_fit_fp16` and every cached fragment packing (call after editing weights
    through `.data`, which does not bump a tensor's version counter)."""
    _UNIT_OK.clear()
    _PACKED.clear()


_PACKED = {}
_PACK_IN_CAPTURE = [False]


class pack_inside_capture:
    """While active, a stream capture never takes a cached fragment packing: the repacking kernels are captured with the
    pass, so a graph that is replayed across optimiser steps (graphs.GraphedTrainStep) always multiplies by the CURRENT
    weights.  (GraphedFlow, an inference graph, keeps the cached packing out of its graph and re-captures instead.)"""

    def __enter__(self):
        self._old = _PACK_IN_CAPTURE[0]
        _PACK_IN_CAPTURE[0] = True

    def __exit__(self, *exc):
        _PACK_IN_CAPTURE[0] = self._old
        return False


def _cached_pack(w, kind, fn):
    """fn(w.detach()) -- a weight tensor repacked into a kernel's fragment layout -- cached per live tensor OBJECT (the
    parameter a view like Conv4d.weight is based on, held by weak reference as in `_weights_fit_fp16`) and version, so that
    an inference pass does not repack (a dozen small torch kernels per layer and launch) weights that have not changed.
    A detached alias or a temporary has no live base: it is packed every time."""
    if _PACK_IN_CAPTURE[0] and torch.cuda.is_current_stream_capturing():
        return fn(w.detach())          # a training graph: the packing is part of the graph, replayed on the current values
    base = w._base if w._base is not None else w
    key = (id(base), kind)
    state = (w._version, w.data_ptr(), tuple(w.shape), w.dtype)
    hit = _PACKED.get(key)
    if hit is not None and hit[0]() is base and hit[1] == state:
        return hit[2]
    out = fn(w.detach())
    _PACKED[key] = (weakref.ref(base, lambda _, k=key: _PACKED.pop(k, None)), state, out)
    return out


def _weights_fit_fp16(w):
    """finite and inside the fp16 range (the split-fp16 kernel's precondition on the weights); cached per
    parameter version so that the device->host sync happens once per optimiser step, not per launch.
    An entry belongs to one live tensor OBJECT (the parameter a view like Conv4d.weight is based on), held by weak
    reference: when that tensor dies its entry goes with it, so a new tensor that reuses the freed address never inherits
    a verdict, and temporaries (a flipped / transposed copy made for one call) are checked every time.
    In-place ops under no_grad (optimizers, `p.copy_`) bump the version; writes through `p.data` do NOT -- after
    those call `invalidate_weight_checks()` (ModelDeviceHandler.broadcast_parameters does).  Under stream capture
    (GraphedFlow) the check cannot run: only an already cached verdict is accepted, so GraphedFlow validates the
    weights eagerly before it captures and must be re-captured after the weights change."""
    base = w._base if w._base is not None else w
    key = id(base)
    state = (w._version, w.data_ptr(), tuple(w.shape))
    hit = _UNIT_OK.get(key)
    if hit is not None and hit[0]() is base and hit[1] == state:
        return hit[2]
    if torch.cuda.is_current_stream_capturing():
        return False            # cannot synchronise here; the fp32 kernels are always valid
    ok = bool(torch.isfinite(w.detach()).all()) and float(w.detach().abs().max()) * SPLIT16_WEIGHT_SCALE < 3.0e4
    _UNIT_OK[key] = (weakref.ref(base, lambda _, k=key: _UNIT_OK.pop(k, None)), state, ok)
    return ok


def conv_supported(x, weight):
    return (x.is_cuda and x.dtype in (torch.float32, torch.float64) and weight.dtype == x.dtype
            and 1 <= x.dim() - 2 <= 4)


_GATHER_MAPS = {}


def _pack_by_gather(weight, build, key):
    """build(weight) -- a pure re-arrangement of the weights with zero padding (two-site expansion, fragment order, row
    packing: a dozen small torch kernels) -- as ONE gather: the source index of every output element is found once per layer
    shape by running `build` on a tensor of element numbers, then every packing is one nf_gather_pad launch.  (A training step
    re-packs every layer's weights twice, forward and flipped / transposed for the input gradient: on small lattices those
    launches were a quarter of the step.)"""
    key = (tuple(weight.shape), weight.device) + key
    ent = _GATHER_MAPS.get(key)
    if ent is None:
        probe = torch.arange(1, weight.numel() + 1, dtype=torch.float64, device=weight.device).reshape(weight.shape)
        packed = build(probe)
        idx = (packed.reshape(-1).round().to(torch.int64) - 1).to(torch.int32).contiguous()      # -1: a zero of the padding
        ent = (idx, tuple(packed.shape))
        _GATHER_MAPS[key] = ent
    idx, shape = ent
    w = weight.contiguous()
    out = torch.empty(shape, dtype=w.dtype, device=w.device)
    _check(load().nf_gather_pad(_ptr(w), _ptr(idx), _ptr(out), idx.numel(), w.numel(), w.element_size(), _stream()), "nf_gather_pad")
    return out


def _conv_launch(x, weight, bias, act, compact, parity, weight_src=None, transposed=False):
    """One launch sequence of nf_conv_fwd for a (cout, cin, *k) weight tensor; packs the weights in
    the fragment layout the library will use (two-site column packing for cout <= 8).  transposed: the layer is the INPUT
    GRADIENT of the layer with these weights (flipped taps, channels exchanged)."""
    lib = load()
    B, cin = x.shape[:2]
    lat = list(x.shape[2:])
    cout, ksize = weight.shape[1 if transposed else 0], list(weight.shape[2:])
    d = len(lat)
    lat4 = (C.c_int32 * 4)(*([1] * (4 - d) + lat))
    k4 = (C.c_int32 * 4)(*([1] * (4 - d) + ksize))
    split16 = compact == 2                 # NF_OUT_SPLIT16: the fp16 (hi, lo) pair tensor, (B, V, 16) halfs
    if (split16 and cin == 1 and cout == 8 and d == 4 and x.dtype == torch.float32
            and _weights_fit_fp16(weight if weight_src is None else weight_src)     # (the caller's tensor: its verdict is cached)
            and lib.nf_conv_first_split16_supported(lat4, k4, cout, act)):
        return conv_first_split16(x, weight, bias, act, weight_src)
    if split16 and not (lib.nf_conv_two_site(cout, 0, lat[-1], ksize[-1]) and cout == 8 and x.dtype == torch.float32):
        raise NormflowHipError("split-fp16 output needs an fp32 two-site layer with 8 output channels")
    two_site = bool(lib.nf_conv_two_site(cout, 0 if split16 else int(compact), lat[-1], ksize[-1]))
    eff_compact = False if (two_site and split16) else compact

    def build(w):
        """the layer's weights (or its flipped / transposed weights: the input gradient's) in the library's fragment layout"""
        if transposed:
            w = w.flip(list(range(2, w.dim()))).transpose(0, 1)
        if two_site:
            # 16 columns: [0, cout) = the layer at site 2p (taps 0..k3-1), [8, 8+cout) = the same
            # channels at site 2p+1 (taps 1..k3): one extra tap along the fastest axis
            w2 = w.new_zeros((16, cin) + tuple(ksize[:-1]) + (ksize[-1] + 1,))
            w2[:cout, ..., :ksize[-1]] = w
            w2[8:8 + cout, ..., 1:] = w
            w = w2
        return conv_weight_for_layer(w, lat4, k4, cin, cout, eff_compact, False, _dtype_code(x))

    wfrag = _pack_by_gather(weight, build, (transposed, two_site, tuple(lat4), tuple(k4), cin, cout, int(eff_compact), _dtype_code(x)))
    V = 1
    for n in lat:
        V *= n
    if split16:
        out = torch.empty((B, V, 16), dtype=torch.float16, device=x.device)
    else:
        out = torch.empty((B, cout, V // 2) if compact else (B, cout) + tuple(lat), dtype=x.dtype, device=x.device)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        _check(lib.nf_conv_fwd(_ptr(x[b0:b1]), _ptr(wfrag), _ptr(bias), _ptr(out[b0:b1]), b1 - b0, lat4, k4,
                               cin, cout, act, int(compact), int(parity), _dtype_code(x), _stream()), "nf_conv_fwd")
    return out


def _compact_to_full(t, lattice, parity):
    """(B, C, V/2) pair-compact -> (B, C, *L) with zeros at the other sites (nf_expand_pairs: the checkerboard is
    rebuilt from the coordinate sum)."""
    _require_device(t)
    B, Cc, Vh = t.shape
    t = t.contiguous()
    full = torch.empty((B, Cc) + tuple(lattice), dtype=t.dtype, device=t.device)
    lat4 = (C.c_int32 * 4)(*([1] * (4 - len(lattice)) + list(lattice)))
    _check(load().nf_expand_pairs(_ptr(t), _ptr(full), B * Cc, lat4, int(parity), _dtype_code(t), _stream()),
           "nf_expand_pairs")
    return full


def _lat4(lat, ksize):
    d = len(lat)
    return (C.c_int32 * 4)(*([1] * (4 - d) + list(lat))), (C.c_int32 * 4)(*([1] * (4 - d) + list(ksize)))


def absmax_bits(t):
    """nf_absmax_bits: max |t| of an fp32 tensor as float bits in a 1-element int32 device tensor (no host sync)."""
    bits = torch.empty(1, dtype=torch.int32, device=t.device)
    _check(load().nf_absmax_bits(_ptr(t), t.numel(), _ptr(bits), _stream()), "nf_absmax_bits")
    return bits


def conv_weight_grad(x, gz, ksize, bits=None, compact_parity=-1):
    """(grad_weight (cout, cin, *k), grad_bias (cout)) of a circular conv layer from its input x
    (B, cin, *L) and the full-lattice pre-activation cotangent gz (B, cout, *L): nf_conv_wgrad_split16 (4-D lattice networks;
    `bits`: absmax_bits(gz) if the caller has it already), nf_conv_wgrad_sites (layers of few columns: 1- to 3-D kernels;
    bitwise reproducible) or nf_conv_wgrad.  compact_parity 0 / 1: gz is the pair-compact (B, cout, V/2) cotangent of an
    active-site-only layer; the first two kernels read that form: None is returned when the shape does not qualify (the
    caller expands gz and calls again)."""
    lib = load()
    B, cin = x.shape[:2]
    cout = gz.shape[1]
    ntaps = 1
    for k in ksize:
        ntaps *= k
    lat4, k4 = _lat4(x.shape[2:], ksize)
    ncols = lib.nf_conv_wgrad_cols(cin, ntaps)
    gws, gbs = [], []
    for c0 in range(0, cout, 48):
        c1 = min(cout, c0 + 48)
        part = gz if (c0 == 0 and c1 == cout) else gz[:, c0:c1]
        part = part.contiguous()
        buf = torch.zeros(((c1 - c0 + 15) // 16) * 16, ncols, dtype=x.dtype, device=x.device)
        split = (x.dtype == torch.float32 and lib.nf_get_option(OPT_SPLIT16)
                 and lib.nf_conv_wgrad_split16_supported(lat4, k4, cin, c1 - c0))
        sites = not split and bool(lib.nf_conv_wgrad_sites_supported(lat4, k4, cin, c1 - c0, _dtype_code(x)))
        if compact_parity >= 0 and not (split or sites):
            return None
        if compact_parity >= 0 and sites and x.shape[-1] % 2:
            return None
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            if split:       # fp16 matrix cores, three products per fp32 product (nf_conv_w.hip)
                need = lib.nf_conv_wgrad_split16_workspace(b1 - b0, lat4, cin)
                ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
                if bits is None:
                    bits = absmax_bits(gz)      # (of the whole cotangent: any power of two that fits the maximum will do)
                _check(lib.nf_conv_wgrad_split16(_ptr(x[b0:b1]), _ptr(part[b0:b1]), _ptr(buf), b1 - b0, lat4, k4, cin,
                                                 c1 - c0, _ptr(bits), int(compact_parity), _ptr(ws), ws.numel(), _stream()),
                       "nf_conv_wgrad_split16")
            elif sites:     # few columns (1- to 3-D kernels): the waves split the sites, fixed-order sums (nf_conv.hip)
                need = lib.nf_conv_wgrad_sites_workspace(k4, cin, c1 - c0, _dtype_code(x))
                ws = torch.empty(int(need), dtype=torch.uint8, device=x.device)
                _check(lib.nf_conv_wgrad_sites(_ptr(x[b0:b1]), _ptr(part[b0:b1]), _ptr(buf), b1 - b0, lat4, k4, cin,
                                               c1 - c0, int(compact_parity), _ptr(ws), ws.numel(), _dtype_code(x), _stream()),
                       "nf_conv_wgrad_sites")
            else:
                _check(lib.nf_conv_wgrad(_ptr(x[b0:b1]), _ptr(part[b0:b1]), _ptr(buf), b1 - b0, lat4, k4, cin,
                                         c1 - c0, _dtype_code(x), _stream()), "nf_conv_wgrad")
        gws.append(buf[:c1 - c0, :ntaps * cin])
        gbs.append(buf[:c1 - c0, ntaps * cin])
    gw = torch.cat(gws) if len(gws) > 1 else gws[0]
    gb = torch.cat(gbs) if len(gbs) > 1 else gbs[0]
    gw = gw.reshape(cout, ntaps, cin).permute(0, 2, 1).reshape((cout, cin) + tuple(ksize))
    return gw, gb


def conv_input_grad_split16(gz, wt, bits=None, compact_parity=-1, lattice=None, weight=None):
    """Gradient w.r.t. the input of a 3^4 layer with up to 8 input channels on the split-fp16 chain: gz (B, C, *L) fp32 cotangent
    of the layer's pre-activation, wt (cin, C, 3, 3, 3, 3) its weights flipped and transposed.  The C channels go through the
    hidden-layer kernel in groups of 8 (nf_planes_to_split16 + nf_conv_dgrad_split16).  None when the shape does not qualify
    (the caller then runs the fp32 kernel)."""
    lib = load()
    cin = wt.shape[0]                  # channels of the layer's input = of the gradient (fewer than 8: zero weight rows)
    if compact_parity < 0:
        lattice = tuple(gz.shape[2:])
    if (gz.dtype != torch.float32 or lattice is None or len(lattice) != 4 or cin > 8 or tuple(wt.shape[2:]) != (3, 3, 3, 3)
            or not lib.nf_get_option(OPT_SPLIT16)):
        return None
    B, Cc = gz.shape[:2]
    lat4, k4 = _lat4(lattice, (3, 3, 3, 3))
    if not lib.nf_conv_split16_supported(lat4, k4, 8, 8, ACT_CODES['tanh']) or B > 65535:
        return None
    if not _weights_fit_fp16(wt if weight is None else weight):      # (`weight`: the layer's parameter, whose verdict is cached)
        return None
    G = (Cc + 7) // 8
    V = 1
    for n in lattice:
        V *= n
    gz = gz.contiguous()
    g16 = torch.empty((G, B, V, 16), dtype=torch.float16, device=gz.device)
    if bits is None:
        bits = absmax_bits(gz)
    _check(lib.nf_planes_to_split16(_ptr(gz), _ptr(g16), _ptr(bits), B, Cc, lat4, int(compact_parity), _stream()),
           "nf_planes_to_split16")
    gx = torch.empty((B, 8) + tuple(lattice), dtype=torch.float32, device=gz.device)
    wpad = wt.new_zeros((8, 8 * G, 3, 3, 3, 3), dtype=torch.float32)
    wpad[:cin, :Cc] = wt.float()
    for g in range(G):
        wsp = pack_conv_weight_split16_two_site(wpad[:, 8 * g:8 * g + 8].contiguous())
        _check(lib.nf_conv_dgrad_split16(_ptr(g16[g]), _ptr(wsp), None, _ptr(gx), B, lat4, _ptr(bits), int(g > 0), 0,
                                         _stream()), "nf_conv_dgrad_split16")
    return gx if cin == 8 else gx[:, :cin].contiguous()


def conv_hidden_planes_split16(x, weight, bias, act):
    """A forward 8 -> 8 layer (3^4, tanh or no activation) on the split-fp16 chain with fp32 planes in and out -- the form
    autograd keeps (ConvFn): nf_absmax_bits + nf_planes_to_split16 + nf_conv_dgrad_split16.  None when the shape does not qualify."""
    lib = load()
    if (x.dtype != torch.float32 or x.dim() != 6 or tuple(weight.shape) != (8, 8, 3, 3, 3, 3) or act not in (0, ACT_CODES['tanh'])
            or not lib.nf_get_option(OPT_SPLIT16)):
        return None
    B = x.shape[0]
    lattice = tuple(x.shape[2:])
    lat4, k4 = _lat4(lattice, (3, 3, 3, 3))
    if not lib.nf_conv_split16_supported(lat4, k4, 8, 8, ACT_CODES['tanh']) or B > 65535 or not _weights_fit_fp16(weight):
        return None
    V = x[0, 0].numel()
    x16 = torch.empty((1, B, V, 16), dtype=torch.float16, device=x.device)
    bits = absmax_bits(x)                        # (any input range: the pair tensor is scaled like a cotangent's)
    _check(lib.nf_planes_to_split16(_ptr(x), _ptr(x16), _ptr(bits), B, 8, lat4, -1, _stream()), "nf_planes_to_split16")
    wsp = pack_conv_weight_split16_two_site(weight.detach().float())
    out = torch.empty_like(x)
    b = None if bias is None else bias.detach().float().contiguous()
    _check(lib.nf_conv_dgrad_split16(_ptr(x16[0]), _ptr(wsp), _ptr(b), _ptr(out), B, lat4, _ptr(bits), 0, int(act), _stream()),
           "nf_conv_dgrad_split16")
    return out


def conv_last_logits_split16(x, weight, bias, parity):
    """The forward 8 -> 46 layer at the active sites of parity `parity` on the split-fp16 kernel, logits pair-compact
    (B, 46, V/2) -- the form autograd keeps (ConvFn with compact=True): nf_absmax_bits + nf_conv_last_logits_split16.  None
    when the shape does not qualify (a fastest axis of 32 + 16 n sites; other than 32: through a pair tensor)."""
    lib = load()
    if (x.dtype != torch.float32 or x.dim() != 6 or tuple(weight.shape) != (46, 8, 3, 3, 3, 3) or x.shape[1] != 8
            or x.shape[-1] < 32 or x.shape[-1] % 16 or any(n < 2 or n % 2 for n in x.shape[2:5])
            or not lib.nf_get_option(OPT_SPLIT16) or not _weights_fit_fp16(weight)):
        return None
    B = x.shape[0]
    lattice = tuple(x.shape[2:])
    lat4 = (C.c_int32 * 4)(*lattice)
    V = x[0, 0].numel()
    bits = absmax_bits(x)
    wsp = pack_conv_weight_split16(weight.detach().float())
    b = None if bias is None else bias.detach().float().contiguous()
    out = torch.empty((B, 46, V // 2), dtype=torch.float32, device=x.device)
    src, is16 = x, 0
    if lattice[-1] != 32:            # the kernel stages fp32 planes only for whole-row segments: hand it the pair tensor
        src = torch.empty((1, B, V, 16), dtype=torch.float16, device=x.device)
        _check(lib.nf_planes_to_split16(_ptr(x), _ptr(src), _ptr(bits), B, 8, lat4, -1, _stream()), "nf_planes_to_split16")
        is16 = 1
    _check(lib.nf_conv_last_logits_split16(_ptr(src), is16, _ptr(wsp), _ptr(b), _ptr(out), B, lat4, int(parity), _ptr(bits),
                                           _stream()), "nf_conv_last_logits_split16")
    return out


def pack_wide_split16(w1, b1, w2, b2, w3, b3):
    """Weights of a stack 1 -> h -> h -> C with 8 < h <= 16 (3^4 kernels) for `conv_wide_logits_split16`: the hidden channels
    zero-padded to 16 and cut into two groups of 8 -- [first-layer fragments x 2, biases x 2], [two-site fragments of the four
    8 x 8 blocks of the hidden layer, biases x 2], [last-layer fragments x 2, bias]."""
    h, cout = w1.shape[0], w3.shape[0]
    f = lambda t: t.detach().float()
    w1p = f(w1).new_zeros((16, 1, 3, 3, 3, 3)); w1p[:h] = f(w1)
    w2p = f(w2).new_zeros((16, 16, 3, 3, 3, 3)); w2p[:h, :h] = f(w2)
    w3p = f(w3).new_zeros((cout, 16, 3, 3, 3, 3)); w3p[:, :h] = f(w3)
    padb = lambda b, n: None if b is None else torch.nn.functional.pad(f(b), (0, n - b.shape[0]))
    b1p, b2p = padb(b1, 16), padb(b2, 16)
    cut = lambda b, g: None if b is None else b[8 * g:8 * g + 8].contiguous()
    first = [(pack_conv_weight_split16_first(w1p[8 * g:8 * g + 8].contiguous()), cut(b1p, g)) for g in (0, 1)]
    hidden = [[pack_conv_weight_split16_two_site(w2p[8 * go:8 * go + 8, 8 * gi:8 * gi + 8].contiguous()) for gi in (0, 1)] for go in (0, 1)]
    last = [pack_conv_weight_split16(w3p[:, 8 * gi:8 * gi + 8].contiguous()) for gi in (0, 1)]
    return first, (hidden, [cut(b2p, 0), cut(b2p, 1)]), (last, None if b3 is None else f(b3).contiguous()), cout


def conv_wide_logits_split16(x, packed, act1, parity):
    """The raw output (B, C, V/2) at the active sites of a ConvAct stack 1 -> h -> h -> C with hidden widths 9 .. 16 and tanh
    hidden activations on a 4-D lattice, composed from the split-fp16 kernels in groups of 8 channels: the first layer twice
    (nf_conv_first_split16), the four 8 x 8 blocks of the hidden layer through nf_conv_dgrad_split16 (fp32 planes out, the second
    block of a group added to the first, tanh of the sum) + nf_planes_to_split16, the last layer twice (nf_conv_last_logits_split16,
    the second group added).  x: (B, 1, *L) fp32.  Twice as fast as the fp32 MFMA kernels on this shape (tools/hidden16_bench.py)."""
    lib = load()
    first, (hidden, hb), (last, b3), cout = packed
    B = x.shape[0]
    lattice = tuple(x.shape[2:])
    lat4 = (C.c_int32 * 4)(*lattice)
    V = x[0, 0].numel()
    x = x.contiguous()
    tanh = ACT_CODES['tanh']
    h1 = []
    for wsp, b in first:
        out = torch.empty((B, V, 16), dtype=torch.float16, device=x.device)
        _check(lib.nf_conv_first_split16(_ptr(x), _ptr(wsp), _ptr(b), _ptr(out), B, lat4, int(act1), _stream()), "nf_conv_first_split16")
        h1.append(out)
    h2 = []
    z = torch.empty((B, 8) + lattice, dtype=torch.float32, device=x.device)
    for go in (0, 1):
        for gi in (0, 1):
            _check(lib.nf_conv_dgrad_split16(_ptr(h1[gi]), _ptr(hidden[go][gi]), _ptr(hb[go]) if gi == 0 else None, _ptr(z), B, lat4, None,
                                             int(gi > 0), tanh if gi == 1 else 0, _stream()), "nf_conv_dgrad_split16")
        out = torch.empty((1, B, V, 16), dtype=torch.float16, device=x.device)
        _check(lib.nf_planes_to_split16(_ptr(z), _ptr(out), None, B, 8, lat4, -1, _stream()), "nf_planes_to_split16")
        h2.append(out)
    del h1
    logits = torch.empty((B, cout, V // 2), dtype=torch.float32, device=x.device)
    for gi in (0, 1):
        _check(lib.nf_conv_last_logits_split16_acc(_ptr(h2[gi]), 1, _ptr(last[gi]), _ptr(b3) if gi == 0 else None, _ptr(logits), B, lat4,
                                                   int(parity), None, int(gi > 0), int(cout), _stream()), "nf_conv_last_logits_split16_acc")
    return logits


class ConvFn(torch.autograd.Function):
    """One circular conv layer + activation, forward and backward on the MFMA kernels:
    forward nf_conv_fwd; backward nf_act_vjp, nf_conv_fwd with flipped / transposed weights
    (grad_input) and the persistent nf_conv_wgrad kernel (grad_weight, grad_bias)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, compact, parity):
        _require_device(x, weight, bias)
        x = x.contiguous()
        out = None
        if not compact and x.dim() == 6 and x.shape[1] == 8 and weight.shape[0] == 8:
            out = conv_hidden_planes_split16(x, weight, bias, act)      # 8 -> 8 layer of a lattice network: the split-fp16 kernel
        elif compact and act == 0 and x.dim() == 6 and weight.shape[0] == 46:
            out = conv_last_logits_split16(x, weight, bias, parity)     # ... and its 8 -> 46 layer at the active sites
        if out is None:
            out = _conv_launch(x, weight.detach(), bias, act, compact, parity, weight_src=weight)
        ctx.save_for_backward(x, weight, out if act else None)
        ctx.act, ctx.compact, ctx.parity, ctx.has_bias = act, compact, parity, bias is not None
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight, y = ctx.saved_tensors
        lib = load()
        gout = gout.contiguous()
        gz = gout
        if ctx.act:
            gz = torch.empty_like(gout)
            _check(lib.nf_act_vjp(_ptr(gout), _ptr(y), _ptr(gz), gout.numel(), ctx.act, _dtype_code(gout),
                                  _stream()), "nf_act_vjp")
        lattice = tuple(x.shape[2:])
        split_ok = gz.dtype == torch.float32 and len(lattice) == 4 and bool(lib.nf_get_option(OPT_SPLIT16))
        bits = absmax_bits(gz) if split_ok else None
        want_w = ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2])
        kdims = list(range(2, weight.dim()))
        # the input gradient's weights (flipped taps, channels exchanged), made when a kernel needs them as a tensor
        wt = (lambda: weight.detach().flip(kdims).transpose(0, 1).contiguous()) if ctx.needs_input_grad[0] else None
        gx = gw = gb = None
        if ctx.compact:
            # the pair-compact cotangent as it is: the split-fp16 kernels and the few-column weight-gradient kernel read that
            # form, no expanded copy
            gzc = gz.reshape(x.shape[0], weight.shape[0], -1).contiguous()
            if wt is not None and split_ok:
                gx = conv_input_grad_split16(gzc, wt(), bits, ctx.parity, lattice, weight)
            if want_w:
                got = conv_weight_grad(x, gzc, weight.shape[2:], bits, ctx.parity)
                if got is not None:
                    gw, gb = got
        need_full = (wt is not None and gx is None) or (want_w and gw is None)
        if need_full:
            if ctx.compact:
                gz = _compact_to_full(gz, lattice, ctx.parity)
            gz = gz.reshape((x.shape[0], weight.shape[0]) + lattice).contiguous()
            if wt is not None and gx is None:
                gx = conv_input_grad_split16(gz, wt(), bits, weight=weight) if split_ok else None
                if gx is None:
                    gx = _conv_launch(gz, weight.detach(), None, 0, False, 0, transposed=True)
            if want_w and gw is None:
                gw, gb = conv_weight_grad(x, gz, weight.shape[2:], bits)
        if not ctx.has_bias:
            gb = None
        return gx, gw, gb, None, None, None


def fused_last_rqs_trainable(h, weight):
    """Does the differentiable fused node (FusedLastRqsFn) take this last layer?  h: (B, 8, *L) fp32 hidden activations."""
    lib = load()
    if (h.dtype != torch.float32 or h.dim() != 6 or h.shape[1] != 8 or weight.dim() != 6 or weight.shape[1] != 8
            or tuple(weight.shape[2:]) != (3, 3, 3, 3) or weight.shape[0] > 46 or (weight.shape[0] + 2) % 3
            or not lib.nf_get_option(OPT_SPLIT16)):
        return False
    lat4 = (C.c_int32 * 4)(*h.shape[2:])
    return bool(lib.nf_conv_rqs_split16_supported(lat4, weight.shape[0], (weight.shape[0] + 2) // 3)) and _weights_fit_fp16(weight)


class FusedLastRqsFn(torch.autograd.Function):
    """The last ConvAct layer (8 -> 3m-2 at the active sites) and the RQ-spline coupling as ONE differentiable node on the
    split-fp16 kernel: forward nf_conv_rqs_split16_train, backward nf_conv_rqs_split16_vjp (logits recomputed in the
    kernel, cotangents through the spline) + the weight / input gradient kernels on the pair-compact logit cotangent.
    The (B, 3m-2, V/2) logits are never materialised -- only their cotangent is, once, in the backward pass."""

    @staticmethod
    def _source(h, bits, lat4):
        """The kernel's view of the hidden activations: fp32 planes for whole-row segments, else the pair tensor."""
        if h.shape[-1] == 32:
            return h, 0
        B, V = h.shape[0], h[0, 0].numel()
        src = torch.empty((1, B, V, 16), dtype=torch.float16, device=h.device)
        _check(load().nf_planes_to_split16(_ptr(h), _ptr(src), _ptr(bits), B, 8, lat4, -1, _stream()), "nf_planes_to_split16")
        return src, 1

    @staticmethod
    def forward(ctx, h, weight, bias, x_active, log0, parity, opts, inverse):
        _require_device(h, weight, bias, x_active, log0)
        lib = load()
        h, x_active = h.contiguous(), x_active.contiguous()
        B, V = x_active.shape
        lattice = tuple(h.shape[2:])
        lat4 = (C.c_int32 * 4)(*lattice)
        bits = absmax_bits(h)
        src, is16 = FusedLastRqsFn._source(h, bits, lat4)
        wsp = pack_conv_weight_split16(weight.detach().float())
        b = None if bias is None else bias.detach().float().contiguous()
        y = torch.empty_like(x_active)
        logj = torch.empty(B, dtype=torch.float32, device=h.device)
        ws = _workspace(min(B, MAX_B), V, h.device)
        _check(lib.nf_conv_rqs_split16_train(_ptr(src), is16, _ptr(wsp), _ptr(b), weight.shape[0], _ptr(x_active), _ptr(log0),
                                             _ptr(y), _ptr(logj), B, lat4, int(parity), _ptr(bits), C.byref(opts),
                                             int(bool(inverse)), _ptr(ws), ws.numel(), _stream()), "nf_conv_rqs_split16_train")
        ctx.save_for_backward(h, weight, bias, y if inverse else x_active)
        ctx.parity, ctx.opts, ctx.inverse, ctx.has_log0 = parity, opts, inverse, log0 is not None
        return y, logj

    @staticmethod
    def backward(ctx, gy, glogj):
        h, weight, bias, xpt = ctx.saved_tensors
        lib = load()
        B, V = xpt.shape
        lattice = tuple(h.shape[2:])
        lat4 = (C.c_int32 * 4)(*lattice)
        cout = weight.shape[0]
        bits = absmax_bits(h)
        src, is16 = FusedLastRqsFn._source(h, bits, lat4)
        wsp = pack_conv_weight_split16(weight.detach().float())
        b = None if bias is None else bias.detach().float().contiguous()
        gz = torch.empty((B, cout, V // 2), dtype=torch.float32, device=h.device)
        gx = torch.empty_like(xpt)
        _check(lib.nf_conv_rqs_split16_vjp(_ptr(src), is16, _ptr(wsp), _ptr(b), cout, _ptr(xpt), _ptr(gy.contiguous()),
                                           _ptr(glogj.contiguous()), _ptr(gz), _ptr(gx), B, lat4, int(ctx.parity), _ptr(bits),
                                           C.byref(ctx.opts), int(bool(ctx.inverse)), _stream()), "nf_conv_rqs_split16_vjp")
        gh = gw = gb = None
        gbits = absmax_bits(gz)
        if ctx.needs_input_grad[0]:
            wt = weight.detach().flip([2, 3, 4, 5]).transpose(0, 1).contiguous()
            gh = conv_input_grad_split16(gz, wt, gbits, ctx.parity, lattice, weight)
        if ctx.needs_input_grad[1] or (bias is not None and ctx.needs_input_grad[2]):
            got = conv_weight_grad(h, gz, weight.shape[2:], gbits, ctx.parity)
            if got is not None:
                gw, gb = got
        if (ctx.needs_input_grad[0] and gh is None) or ((ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) and gw is None):
            raise NormflowHipError("FusedLastRqsFn.backward: the split-fp16 gradient kernels refused a layer the forward pass took")
        if bias is None:
            gb = None
        return gh, gw, gb, gx, (glogj if ctx.has_log0 else None), None, None, None


def conv_layer(x, weight, bias, act=0, compact=False, parity=0):
    """Circular 'same' conv + activation on the HIP kernel; differentiable."""
    if torch.is_grad_enabled() and (x.requires_grad or weight.requires_grad or (bias is not None and bias.requires_grad)):
        if act == 5:      # |.| hides the sign its derivative needs: keep the activation outside
            return torch.abs(ConvFn.apply(x, weight, bias, 0, compact, parity))
        return ConvFn.apply(x, weight, bias, act, compact, parity)
    x = x.contiguous()
    return _conv_launch(x, weight.detach(), None if bias is None else bias.detach(), act, compact, parity, weight_src=weight)


def conv_affine_split16(h16, weight, bias, x_active, log0, parity, inverse, lattice, out=None):
    """Fused last layer of an affine coupling's net + the coupling (nf_conv_affine_split16); inference only.
    h16: the (B, V, 16) fp16 pair tensor of hidden activations; weight (2, 8, 3, 3, 3, 3), bias (2) | None; x_active (B, V)
    fp32 or half; returns (y (B, V), logJ (B) fp32 / the field's fp32-or-wider dtype)."""
    _require_device(h16, weight, bias, x_active, log0)
    lib = load()
    B, V = x_active.shape
    x_active = x_active.contiguous()
    lat4 = (C.c_int32 * 4)(*lattice)
    def pack_affine(w):
        w8 = w.new_zeros((8, 8, 3, 3, 3, 3), dtype=torch.float32)
        w8[:2] = w.float()
        return pack_conv_weight_split16_two_site(w8)
    wsp = _cached_pack(weight, 'affine16', pack_affine)
    b8 = torch.zeros(8, dtype=torch.float32, device=x_active.device)
    if bias is not None:
        b8[:2] = bias.detach().float()
    field16 = x_active.dtype == torch.float16
    if out is None:
        y = torch.empty_like(x_active)
        logj = torch.empty(B, dtype=torch.float32, device=x_active.device)
    else:
        y, logj = out
    ws = _workspace(min(B, MAX_B), V, x_active.device)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        l0 = log0[b0:b1] if log0 is not None else None
        _check(lib.nf_conv_affine_split16(_ptr(h16[b0:b1]), _ptr(wsp), _ptr(b8), _ptr(x_active[b0:b1]), _ptr(l0),
                                          _ptr(y[b0:b1]), _ptr(logj[b0:b1]), b1 - b0, lat4, int(parity), int(inverse),
                                          4 if field16 else 0, _ptr(ws), ws.numel(), _stream()), "nf_conv_affine_split16")
    return y, logj


def conv_rqs(h, weight, bias, x_active, log0, parity, opts, inverse, unit_input=False, lattice=None, out=None):
    """Fused last conv layer + RQ-spline coupling (nf_conv_rqs); inference only.
    h: (B, cin, *L) fp32 hidden activations, or -- with `lattice` given -- the (B, V, 16) fp16 (hi, lo) pairs a
    previous conv_layer(..., compact=2) wrote; x_active: (B, V); returns (y (B, V), logJ (B)).  `out` = (y, logJ)
    tensors to fill instead of new ones (contiguous slices of a caller's batch: no concatenation afterwards)."""
    _require_device(h, weight, bias, x_active, log0)
    lib = load()
    h, x_active = h.contiguous(), x_active.contiguous()
    split_in = lattice is not None
    B = h.shape[0]
    cin = weight.shape[1] if split_in else h.shape[1]
    lat = list(lattice) if split_in else list(h.shape[2:])
    d = len(lat)
    lat4 = (C.c_int32 * 4)(*([1] * (4 - d) + lat))
    k4 = (C.c_int32 * 4)(*([1] * (4 - d) + list(weight.shape[2:])))
    V = x_active.shape[1]
    flags = 1 if (unit_input and _weights_fit_fp16(weight)) else 0          # NF_CONV_UNIT_INPUT: |h| <= 1 (tanh outputs)
    field16 = x_active.dtype == torch.float16                                # NF_CONV_FIELD_F16: fp16 field storage
    if field16:
        flags |= 4
    ldt = torch.float32 if field16 else x_active.dtype
    if split_in:
        if not flags or h.dtype != torch.float16 or tuple(h.shape) != (B, V, 16):
            raise NormflowHipError("split-fp16 hidden activations need unit_input and a (B, V, 16) half tensor")
        flags |= 2                                                           # NF_CONV_SPLIT16_INPUT
    fused_code = 1 | ((flags & 1) << 1) | ((flags & 2) << 1)
    kind = ('fused_last', tuple(lat4), tuple(k4), cin, fused_code, lib.nf_get_option(OPT_SPLIT16), lib.nf_get_option(OPT_PIPE))
    wfrag = _cached_pack(weight, kind, lambda w: conv_weight_for_layer(w, lat4, k4, cin, w.shape[0], True, fused_code, NF_F32))
    bias = None if bias is None else bias.detach().contiguous()
    if out is None:
        y = torch.empty_like(x_active)
        logj = torch.empty(B, dtype=ldt, device=x_active.device)
    else:
        y, logj = out
        if (tuple(y.shape) != (B, V) or tuple(logj.shape) != (B,) or y.dtype != x_active.dtype or logj.dtype != ldt
                or not y.is_contiguous() or not logj.is_contiguous()):
            raise NormflowHipError("conv_rqs: `out` must be contiguous (B, V) and (B,) tensors of the input's dtype (log|J| fp32 for a half field)")
        _require_device(y, logj)
    ws = _workspace(min(B, MAX_B), V, h.device)
    for b0 in range(0, B, MAX_B):
        b1 = min(B, b0 + MAX_B)
        l0 = log0[b0:b1] if log0 is not None else None
        _check(lib.nf_conv_rqs(_ptr(h[b0:b1]), _ptr(wfrag), _ptr(bias), _ptr(x_active[b0:b1]), _ptr(l0),
                               _ptr(y[b0:b1]), _ptr(logj[b0:b1]), b1 - b0, lat4, k4, cin, weight.shape[0],
                               int(parity), C.byref(opts), int(inverse), flags, _ptr(ws), ws.numel(), NF_F32,
                               _stream()), "nf_conv_rqs")
    return y, logj


# ============================================================ small 3-D lattices: one kernel per layer
def _split_hi_lo(w):
    w = w.float() * SPLIT16_WEIGHT_SCALE
    hi = w.half()
    return hi, (w - hi.float()).half()


def pack_small3d_weights(w1, w2, w3):
    """(8, 1, 3, 3, 3), (8, 8, 3, 3, 3), (cout <= 48, 8, 3, 3, 3) fp32 weights -> the three fragment tensors of
    nf_small3d_rqs (include/normflow_hip.h): scaled by 2^10, split into fp16 (hi, lo)."""
    dev = w1.device
    cout = w3.shape[0]
    assert tuple(w1.shape) == (8, 1, 3, 3, 3) and tuple(w2.shape) == (8, 8, 3, 3, 3) and tuple(w3.shape[1:]) == (8, 3, 3, 3) and cout <= 48
    # w1: A operand rows = output channel (16, 8 used), K = tap (32, 27 used); lane 16 g + row holds k = 8g .. 8g+7
    a1 = torch.zeros(16, 32, device=dev)
    a1[:8, :27] = w1.reshape(8, 27).float()
    p1 = torch.stack([t.reshape(16, 4, 8).permute(1, 0, 2).reshape(64, 8) for t in _split_hi_lo(a1)])            # (2, 64, 8)
    # w2: per kernel row (j0, j1): rows = (site-in-pair s, co), K = (tap g of the pair, ci): w[co][ci][j0][j1][g - s]
    a2 = torch.zeros(9, 16, 4, 8, device=dev)                       # row, m = 8 s + co, g, ci
    wr = w2.float().permute(2, 3, 0, 4, 1).reshape(9, 8, 3, 8)      # (j0 j1), co, j2, ci
    a2[:, :8, 0:3] = wr
    a2[:, 8:, 1:4] = wr
    p2 = torch.stack([t.permute(0, 2, 1, 3).reshape(9, 64, 8) for t in _split_hi_lo(a2)], dim=1)                  # (9, 2, 64, 8)
    # w3: per column tile t and slice i: cols = channel 16 t + n, K = (tap 4 i + g, ci)
    b3 = torch.zeros(48, 28, 8, device=dev)                         # channel, tap, ci
    b3[:cout, :27] = w3.float().reshape(cout, 8, 27).permute(0, 2, 1)
    p3 = torch.stack([t.reshape(3, 16, 7, 4, 8).permute(0, 2, 3, 1, 4).reshape(3, 7, 64, 8) for t in _split_hi_lo(b3)], dim=2)
    return p1.contiguous(), p2.contiguous(), p3.contiguous()        # (2,64,8), (9,2,64,8), (3,7,2,64,8)


def small_lattice_coupling(kind, x_frozen, x_active, packed, biases, log0, parity, cout, acts, opts, inverse):
    """One launch per coupling layer of a small lattice (nf_small_lattice_coupling): kind 0 RQ-spline, 1 affine;
    x_frozen, x_active: (B, L0, L1, 16) or (B, L1, 16) fp32; inference only."""
    _require_device(x_frozen, x_active, log0, *packed)
    lib = load()
    B = x_active.shape[0]
    lat = tuple(x_active.shape[1:])
    latc = (C.c_int32 * len(lat))(*lat)
    x_frozen, x_active = x_frozen.contiguous(), x_active.contiguous()
    y = torch.empty_like(x_active)
    logj = torch.empty(B, dtype=torch.float32, device=x_active.device)
    b1, b2, b3 = (None if b is None else b.detach().float().contiguous() for b in biases)
    _check(lib.nf_small_lattice_coupling(int(kind), _ptr(x_frozen), _ptr(x_active), _ptr(packed[0]), _ptr(b1), _ptr(packed[1]),
                                         _ptr(b2), _ptr(packed[2]), _ptr(b3), _ptr(log0), _ptr(y), _ptr(logj), B, latc, len(lat),
                                         int(parity), int(cout), int(acts[0]), int(acts[1]),
                                         C.byref(opts) if opts is not None else None, int(bool(inverse)), _stream()),
           "nf_small_lattice_coupling")
    return y, logj


def small3d_rqs(x_frozen, x_active, packed, biases, log0, parity, cout, acts, opts, inverse, out=None):
    """One launch per RQ-spline coupling layer of a small 3-D lattice (nf_small3d_rqs); inference only.
    x_frozen, x_active: (B, L0, L1, 16) fp32; packed = pack_small3d_weights(...); biases = (b1, b2, b3) fp32 or None each."""
    _require_device(x_frozen, x_active, log0, *packed)
    lib = load()
    B = x_active.shape[0]
    lat3 = (C.c_int32 * 3)(*x_active.shape[1:])
    x_frozen, x_active = x_frozen.contiguous(), x_active.contiguous()
    if out is None:
        y = torch.empty_like(x_active)
        logj = torch.empty(B, dtype=torch.float32, device=x_active.device)
    else:
        y, logj = out
    b1, b2, b3 = (None if b is None else b.detach().float().contiguous() for b in biases)
    _check(lib.nf_small3d_rqs(_ptr(x_frozen), _ptr(x_active), _ptr(packed[0]), _ptr(b1), _ptr(packed[1]), _ptr(b2),
                              _ptr(packed[2]), _ptr(b3), _ptr(log0), _ptr(y), _ptr(logj), B, lat3, int(parity), int(cout),
                              int(acts[0]), int(acts[1]), C.byref(opts), int(bool(inverse)), _stream()), "nf_small3d_rqs")
    return y, logj


# ========================================================================= end points
def endpoint_supported(t):
    return t.is_cuda and t.dtype in (torch.float32, torch.float64) and 1 <= t.dim() - 1 <= 4


class Phi4ActionFn(torch.autograd.Function):
    """Per-sample phi^4 action of (B, *L) configurations, one pass (nf_phi4_action)."""

    @staticmethod
    def forward(ctx, cfgs, w0, w2, w4):
        cfgs = cfgs.contiguous()
        B, lat = cfgs.shape[0], list(cfgs.shape[1:])
        lat4 = (C.c_int32 * 4)(*([1] * (4 - len(lat)) + lat))
        out = torch.empty(B, dtype=cfgs.dtype, device=cfgs.device)
        ws = _workspace(min(B, MAX_B), cfgs[0].numel(), cfgs.device)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(load().nf_phi4_action(_ptr(cfgs[b0:b1]), _ptr(out[b0:b1]), b1 - b0, lat4, w0, w2, w4, _ptr(ws),
                                         ws.numel(), _dtype_code(cfgs), _stream()), "nf_phi4_action")
        ctx.save_for_backward(cfgs)
        ctx.w = (w0, w2, w4)
        return out

    @staticmethod
    def backward(ctx, g):
        (cfgs,) = ctx.saved_tensors
        B, lat = cfgs.shape[0], list(cfgs.shape[1:])
        lat4 = (C.c_int32 * 4)(*([1] * (4 - len(lat)) + lat))
        g = g.contiguous()
        gc = torch.empty_like(cfgs)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(load().nf_phi4_action_vjp(_ptr(cfgs[b0:b1]), _ptr(g[b0:b1]), _ptr(gc[b0:b1]), b1 - b0, lat4,
                                             *ctx.w, _dtype_code(cfgs), _stream()), "nf_phi4_action_vjp")
        return gc, None, None, None


class NormalLogProbFn(torch.autograd.Function):
    """Per-sample log-density of independent normals (nf_normal_logprob); loc/scale (V) or None."""

    @staticmethod
    def forward(ctx, x, loc, scale):
        x = x.contiguous()
        B = x.shape[0]
        V = x[0].numel() if B else 0
        out = torch.empty(B, dtype=x.dtype, device=x.device)
        ws = _workspace(min(B, MAX_B), V, x.device)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(load().nf_normal_logprob(_ptr(x[b0:b1]), _ptr(loc), _ptr(scale), _ptr(out[b0:b1]), b1 - b0, V,
                                            _ptr(ws), ws.numel(), _dtype_code(x), _stream()), "nf_normal_logprob")
        ctx.save_for_backward(x, loc, scale)
        return out

    @staticmethod
    def backward(ctx, g):
        x, loc, scale = ctx.saved_tensors
        B = x.shape[0]
        V = x[0].numel() if B else 0
        g = g.contiguous()
        gx = torch.empty_like(x)
        for b0 in range(0, B, MAX_B):
            b1 = min(B, b0 + MAX_B)
            _check(load().nf_normal_logprob_vjp(_ptr(x[b0:b1]), _ptr(loc), _ptr(scale), _ptr(g[b0:b1]),
                                                _ptr(gx[b0:b1]), b1 - b0, V, _dtype_code(x), _stream()),
                   "nf_normal_logprob_vjp")
        return gx, None, None


def normal_sample(loc, scale, batch_size, shape, dtype, device, generator=None):
    """(x (B, *shape), logr (B)) of a NormalPrior in one launch (nf_normal_sample).  Seed and stream offset come from
    torch's CUDA generator of `device` (or `generator`), which is advanced by one Philox call's worth per launch -- so
    torch.manual_seed(s) makes this kernel reproducible exactly as it does torch's own samplers."""
    gen = generator if generator is not None else torch.cuda.default_generators[device.index if device.index is not None
                                                                                  else torch.cuda.current_device()]
    seed, offset = gen.initial_seed(), gen.get_offset()
    gen.set_offset(offset + 4)            # torch keeps offsets in multiples of 4
    V = 1
    for n in shape:
        V *= n
    x = torch.empty((batch_size,) + tuple(shape), dtype=dtype, device=device)
    logr = torch.empty(batch_size, dtype=dtype, device=device)
    ws = _workspace(min(batch_size, MAX_B), V, device)
    loc = None if loc is None else loc.to(device=device, dtype=dtype).contiguous()
    scale = None if scale is None else scale.to(device=device, dtype=dtype).contiguous()
    for b0 in range(0, batch_size, MAX_B):
        b1 = min(batch_size, b0 + MAX_B)
        # slabs of one call use disjoint counter ranges through the high word of the offset
        _check(load().nf_normal_sample(_ptr(x[b0:b1]), _ptr(logr[b0:b1]), _ptr(loc), _ptr(scale), b1 - b0, V,
                                       C.c_uint64(seed & (2 ** 64 - 1)), C.c_uint64((offset // 4) + ((b0 // MAX_B) << 40)),
                                       _ptr(ws), ws.numel(), _dtype_code(x), _stream()), "nf_normal_sample")
    return x, logr
