#!/usr/bin/env python3
"""bench.py -- lattice configs/sec of forward + log|det J| (Posterior.sample_ minus the prior
draw) on BASELINE.json's headline workload: 32^4 phi^4 lattice, 8 RQ-spline coupling layers
(knots_len 16, ConvAct hidden [8,8], kernel 3, tanh), batch 1024 per GPU, fp32, synthetic
inputs already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU (RCCL for the timing barriers only); the batch shards over ranks with no data-path
collective (forward / sampling needs none).  With `--gpus N` and no torchrun environment the parent starts the
N ranks itself (plain child processes, before it makes any GPU call) and relays rank 0's line.

Rank 0 prints ONE JSON line (contract in the task statement).  `value` is the WEAK-scaling aggregate (every rank
runs the full per-GPU batch); extra keys:
  strong               : the same network on a FIXED global batch (--batch, 1024) cut into N shards (BASELINE config 4)
  value_fp32_products  : the same timed loop with exact fp32 MFMA products everywhere (nf_set_option(NF_OPT_SPLIT16, 0));
                         `value` itself forms the conv products of tanh-fed layers from three fp16 MFMA products
  selfcheck            : the two arithmetics compared on a slab of the timed input before timing (the kernels did the work)
  roofline             : the dominant kernel of the timed region, HIP-event timed in here; roofline_kernels: the others
  cpu_baseline         : the CPU oracle (oracle/nf_oracle.py, kind "port") on a bounded sample, fp32 and fp64 legs
"""
import argparse
import hashlib
import json
import os
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")   # bench manages dtype/device itself

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (= vector f32 peak)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (task statement: ~2.5 PFLOP/s; 16x the f32-input rate)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1024, help="batch per GPU (weak scaling) = global batch of the strong-scaling leg")
    ap.add_argument("--lattice", type=str, default="32,32,32,32")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--knots", type=int, default=16)
    ap.add_argument("--scaling", choices=["weak", "strong", "both"], default="both",
                    help="which leg(s) to time at N > 1; `value`/`scaling` of the line are the weak leg unless 'strong'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-products", action="store_true", help="skip the exact-fp32-product leg")
    ap.add_argument("--no-selfcheck", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the short timings of BASELINE configs 2, 3 and 5")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline sample")
    ap.add_argument("--kernel-reps", type=int, default=10)
    ap.add_argument("--master-port", type=int, default=0, help="rendezvous port of a self-launched run (0 = pick a free one)")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ self-launch
def self_launch(a):
    """`python bench.py --gpus N` without a torchrun environment: start N ranks as child processes.  Nothing in this
    function (or before it) initialises the GPU in the parent: torch.cuda.device_count() does not."""
    import socket
    import torch
    rehearsal = os.environ.get("NF_BENCH_REHEARSAL", "0") == "1"
    if not rehearsal:
        n_dev = torch.cuda.device_count()
        if n_dev < a.gpus:
            print(f"bench.py: --gpus {a.gpus} but only {n_dev} GPU(s) are visible (NF_BENCH_REHEARSAL=1 rehearses the "
                  "multi-rank path on one device with gloo)", file=sys.stderr)
            return 2
    port = a.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    procs = []
    for rank in range(a.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(a.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    return rc


# ------------------------------------------------------------------------------------------ workload
def build_net(lattice, layers, m, dev, seed):
    import torch
    from normflow__amd.mask import EvenOddMask
    from normflow__amd.nn import ConvAct, RQSplineCoupling_, ModuleList_
    torch.manual_seed(seed)
    d = len(lattice)
    nets = [ConvAct(1, 3 * m - 2, 3, conv_dim=d, hidden_sizes=[8, 8], acts=['tanh', 'tanh', None])
            for _ in range(layers)]
    cpl = RQSplineCoupling_(nets, mask=EvenOddMask(shape=lattice), xlim=(-5.0, 5.0), ylim=(-5.0, 5.0),
                            extrap={'left': 'linear', 'right': 'linear'})
    net_ = ModuleList_([cpl])
    net_.to(device=dev, dtype=torch.float32)
    # rescale the last conv so that the coupling logits have std ~ 0.5
    with torch.no_grad():
        probe = torch.randn((1, 1) + tuple(lattice), device=dev, dtype=torch.float32)
        for net in cpl.nets:
            last = [mod for mod in net if any(True for _ in mod.parameters())][-1]
            std = float(net(probe).std())
            for p in last.parameters():
                p.mul_(0.5 / max(std, 1e-6))
    return net_, cpl


def _events_ms(f, reps, warm=2):
    """Average duration of f() by HIP events on the stream the kernels are launched on (torch's current stream)."""
    import torch
    for _ in range(warm):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def inloop_kernel_ms(net_, xb):
    """Average duration of the three kernels of a coupling layer INSIDE the timed loop: one more pass of that loop, right after
    the timed steps, with a pair of HIP events around every call of the three C entry points (on torch's current stream, the
    one the library launches on).  A kernel timed in a burst of its own launches (time_fused_last_layer) runs at another
    clock than in the three-kernel sequence; this is the duration the step time is made of.  Small launches (a remainder
    slab) are left out of the mean."""
    import torch
    from normflow__amd import _hip
    lib = _hip.load()
    names = ("nf_conv_rqs", "nf_conv_fwd_split16", "nf_conv_first_split16")
    orig = {n: getattr(lib, n) for n in names}
    rec = {n: [] for n in names}

    def wrap(n):
        f = orig[n]

        def g(*args):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = f(*args)
            e1.record()
            rec[n].append((e0, e1))
            return rc
        return g
    try:
        for n in names:
            setattr(lib, n, wrap(n))
        with torch.no_grad():
            net_(xb)
        torch.cuda.synchronize()
    finally:
        for n in names:
            setattr(lib, n, orig[n])
    out = {}
    for n in names:
        ms = [e0.elapsed_time(e1) for e0, e1 in rec[n]]
        full = [t for t in ms if t > 0.5 * max(ms)] if ms else []
        out[n] = {"ms": sum(full) / len(full), "launches": len(full)} if full else None
    return out


def _vol(lattice):
    V = 1
    for n in lattice:
        V *= n
    return V


def time_rqs_kernel(cpl, lattice, m, dev, reps, layout_pair):
    """One nf_rqs_fwd launch (stand-alone RQ-spline coupling kernel) on the slab shape the unfused pipeline uses."""
    import torch
    from normflow__amd import _hip
    from normflow__amd.nn.scalar import couplings_ as cp
    V = _vol(lattice)
    C = 3 * m - 2
    Vp = V // 2 if layout_pair else V
    slab = max(1, min(64, cp.PARAM_SLAB_BYTES // (C * V * 4)))
    act = cpl.mask.activity(0).reshape(-1).to(dev)
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.randn(slab, V, device=dev, dtype=torch.float32, generator=g) * act.float()
    params = 0.5 * torch.randn(slab, C, Vp, device=dev, dtype=torch.float32, generator=g)
    opts = _hip.make_rqs_opts(m, (-5.0, 5.0), (-5.0, 5.0), {'left': 'linear', 'right': 'linear'},
                              _hip.LAYOUT_PAIR if layout_pair else _hip.LAYOUT_FULL)
    sec = 1e-3 * _events_ms(lambda: _hip.RQSCouplingFn.apply(x, params, None, act, opts, False), reps)
    algo_bytes = slab * (V // 2) * (C + 2) * 4          # B*(V/2)*(C+2)*sizeof
    return dict(seconds=sec, slab=slab, algo_bytes=algo_bytes, gbs=algo_bytes / sec / 1e9)


def _pipeline_slab(cpl, lattice, batch):
    net = cpl.nets[0]
    hidden = max(net.conv_kwargs['hidden_sizes'])
    from normflow__amd.nn.scalar import couplings_
    return max(1, min(batch, couplings_.HIDDEN_SLAB_BYTES // (hidden * _vol(lattice) * 4))), hidden


def time_fused_last_layer(cpl, lattice, m, dev, reps, batch):
    """One nf_conv_rqs launch (last conv layer 8 -> 3m-2 at the active sites + fused RQ-spline epilogue) on the slab
    shape the pipeline uses.  Algorithmic flops: 2 * 3^d * cin * cout per ACTIVE site (last layer)."""
    import torch
    from normflow__amd import _hip
    V = _vol(lattice)
    net = cpl.nets[0]
    slab, hidden = _pipeline_slab(cpl, lattice, batch)
    last = [mod for mod in net if any(True for _ in mod.parameters())][-1]
    g = torch.Generator(device=dev).manual_seed(98)
    h = torch.tanh(torch.randn((slab, hidden) + tuple(lattice), device=dev, dtype=torch.float32, generator=g))
    x = torch.randn(slab, V, device=dev, dtype=torch.float32, generator=g)
    opts = _hip.make_rqs_opts(m, (-5.0, 5.0), (-5.0, 5.0), {'left': 'linear', 'right': 'linear'}, _hip.LAYOUT_PAIR)
    w, b = last.weight.detach(), last.bias.detach()
    f = lambda: _hip.conv_rqs(h, w, b, x, None, 0, opts, False, unit_input=True)    # h = tanh(...): |h| <= 1
    got = net.hidden_and_last(torch.zeros((1, 1) + tuple(lattice), device=dev, dtype=torch.float32))
    if got is not None and got[3]:      # the pipeline hands the kernel fp16 (hi, lo) pairs: time that form
        h16 = _hip.to_split16(h)
        del h
        f = lambda: _hip.conv_rqs(h16, w, b, x, None, 0, opts, False, unit_input=True, lattice=tuple(lattice))
    sec = 1e-3 * _events_ms(f, reps)
    flops = 2.0 * (3 ** len(lattice)) * hidden * (3 * m - 2) * (V // 2) * slab
    return dict(seconds=sec, slab=slab, flops=flops, tflops=flops / sec / 1e12, path=_hip.load().nf_conv_last_path())


def time_hidden_layers(cpl, lattice, dev, reps, batch):
    """The other two kernels of a coupling layer on the pipeline's slab: first ConvAct layer (1 -> 8, HBM-bound: 4 B in +
    32 B out per site) and the hidden 8 -> 8 layer (MFMA-bound: 2*81*8*8 flop per site; 32 B in + 32 B out per site)."""
    import torch
    from normflow__amd import _hip
    V = _vol(lattice)
    net = cpl.nets[0]
    slab, hidden = _pipeline_slab(cpl, lattice, batch)
    convs = [mod for mod in net if any(True for _ in mod.parameters())]
    got = net.hidden_and_last(torch.zeros((1, 1) + tuple(lattice), device=dev, dtype=torch.float32))
    chain = bool(got is not None and got[3])
    g = torch.Generator(device=dev).manual_seed(97)
    x = torch.randn((slab, 1) + tuple(lattice), device=dev, dtype=torch.float32, generator=g)
    out = []
    c0, c1 = convs[0], convs[1]
    tanh = _hip.ACT_CODES['tanh']
    f0 = lambda: _hip.conv_layer(x, c0.weight.detach(), c0.bias.detach(), tanh, compact=2 if chain else False)
    ms0 = _events_ms(f0, reps)
    by0 = slab * V * (4 + 32)
    out.append({"kernel": "first ConvAct layer 1->8 + tanh (nf::conv_c2_kernel, split-fp16 products)" + (", fp16 (hi,lo) pair output" if chain else ""),
                "bound": "hbm", "achieved": by0 / ms0 / 1e6, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": by0 / ms0 / 1e6 / HBM_PEAK_GBS, "launch_ms": ms0, "slab_batch": slab,
                "algorithmic_bytes_per_launch": by0, "traffic": None})
    fl1 = 2.0 * (3 ** len(lattice)) * hidden * hidden * V * slab
    if chain:
        h16 = f0()
        f1 = lambda: _hip.conv_layer_split16(h16, c1.weight.detach(), c1.bias.detach(), tanh, tuple(lattice))
        peak, name = MFMA_F16_PEAK_TFLOPS / 3.0, "hidden ConvAct layer 8->8 + tanh (nf::conv_g2_kernel, split-fp16 products)"
    else:
        hin = torch.tanh(torch.randn((slab, hidden) + tuple(lattice), device=dev, dtype=torch.float32, generator=g))
        f1 = lambda: _hip.conv_layer(hin, c1.weight.detach(), c1.bias.detach(), tanh)
        peak, name = MFMA_F32_PEAK_TFLOPS, "hidden ConvAct layer 8->8 + tanh (fp32 MFMA)"
    ms1 = _events_ms(f1, reps)
    out.append({"kernel": name, "bound": "mfma", "achieved": fl1 / ms1 / 1e9, "peak": peak, "unit": "TFLOP/s",
                "frac": fl1 / ms1 / 1e9 / peak, "launch_ms": ms1, "slab_batch": slab,
                "algorithmic_flops_per_launch": fl1, "traffic": None})
    return out


# ------------------------------------------------------------------------------------------ profiles <-> code identity
def kernel_src_sha():
    """Identity of the kernel sources this process runs (the csrc tree travels to the GPU box; .git does not)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "normflow__amd", "csrc")
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(csrc, name), "rb").read())
    return h.hexdigest()[:16]


def profiled_traffic(kernel_key, slab, lattice, knots):
    """HBM/fabric bytes of one launch of `kernel_key` from the newest committed PMC profile whose recorded
    `kernel_src_sha` equals the sources of THIS build; (None, reason) otherwise -- a stale profile is never quoted."""
    pdir = os.path.join(ROOT, "profiles")
    sha = kernel_src_sha()
    try:
        names = sorted(n for n in os.listdir(pdir) if n.endswith(".json") and "_pmc_" in n)
    except OSError:
        return None, "no profiles/ directory"
    reason = "no PMC profile records this build's kernel_src_sha " + sha
    for name in reversed(names):
        try:
            prof = json.load(open(os.path.join(pdir, name)))
        except (OSError, ValueError):
            continue
        if prof.get("kernel_src_sha") != sha:
            continue
        if tuple(prof.get("lattice", ())) != tuple(lattice) or prof.get("knots") != knots:
            reason = f"{name}: profiled on another workload"
            continue
        k = next((v for kn, v in prof.get("kernels", {}).items() if kernel_key in kn), None)
        if not k or "traffic_bytes_per_launch" not in k:
            continue
        return k["traffic_bytes_per_launch"] * slab / k.get("slab_batch", prof.get("slab_batch", slab)), f"profiles/{name}"
    return None, reason


# ------------------------------------------------------------------------------------------ CPU baseline
def cpu_model_string():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cpl, lattice, m, budget_s):
    """The CPU oracle restatement (plain PyTorch CPU ops, the box's CPU share, fp32 AND fp64) on a bounded sample of the
    same workload: one whole coupling layer (conv stack + RQ spline + log-det) of the bench network at batch 1 on the
    full lattice, one warm-up pass then >= 3 timed repetitions per dtype, median reported, extrapolated linearly to
    all layers (the layers are identical in cost).  `value` is the fp32 figure (the target is judged against fp32)."""
    import torch
    from oracle import nf_oracle as O
    # the GPU box gives one GPU's job a 16-CPU share (task statement); more threads than that only oversubscribes
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))
    torch.set_num_threads(nthreads)
    net = cpl.nets[0]
    convs = [mod for mod in net if any(True for _ in mod.parameters())]
    opts = dict(xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
    n_layers = len(cpl.nets)
    legs = {}
    t_start = time.time()
    for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        layers = [(c.weight.detach().to('cpu', dt).contiguous(), c.bias.detach().to('cpu', dt)) for c in convs]
        g = torch.Generator(device='cpu').manual_seed(4321)
        x = torch.randn((1,) + tuple(lattice), generator=g, dtype=dt, device='cpu')
        masks = [O.channel_mask(lattice, c, dtype=dt) for c in (0, 1)]

        def one_layer():
            with torch.no_grad():
                out = O.conv_act((x * masks[1]).unsqueeze(1), layers, ['tanh', 'tanh', None])
                return O.rqs_coupling_atom(x * masks[0], out, masks[0], log0=0, **opts)
        t0 = time.time()
        one_layer()                                  # warm-up (thread pool, allocator, oneDNN primitive cache)
        warm = time.time() - t0
        times = []
        # >= 3 repetitions, more while the budget (shared by the two legs) allows
        while len(times) < 3 or (len(times) < 5 and time.time() - t_start + 2 * warm < budget_s * (0.5 if name == "fp32" else 1.0)):
            t0 = time.time()
            one_layer()
            times.append(time.time() - t0)
        med = statistics.median(times)
        legs[name] = {"configs_per_s": 1.0 / (med * n_layers), "seconds_per_layer_median": med, "repetitions": len(times),
                      "warmup_seconds": warm}
    return dict(value=legs["fp32"]["configs_per_s"], unit="configs/s", cores=nthreads, kind="port", dtype="fp32",
                value_fp64=legs["fp64"]["configs_per_s"], legs=legs, cpu_model=cpu_model_string(),
                sample=f"one of the {n_layers} identical-cost coupling layers (conv stack 1-8-8-{3 * m - 2} + RQ spline + log-det) of "
                       f"the bench network at batch 1 on the full {'x'.join(map(str, lattice))} lattice, torch CPU ops on "
                       f"{nthreads} threads, 1 warm-up + {legs['fp32']['repetitions']} (fp32) / {legs['fp64']['repetitions']} (fp64) "
                       f"timed repetitions, median, extrapolated linearly to {n_layers} layers; {time.time() - t_start:.1f} s of CPU work")


# ------------------------------------------------------------------------------------------ the other BASELINE configurations
def other_configs(dev):
    """Short timings of BASELINE.json's other GPU configurations (parity-test cases, NOT the bench metric: reported as extra
    fields so that the driver's own run shows them): config 2 (16x16, 4 affine layers, batch 512), config 3 (16^3, 8 RQ-spline
    layers m=16, batch 1024) -- both on the small-lattice fused kernel, one launch per layer --, and config 5's lattice and
    precision (48^4, 8 affine + 8 spline layers, fp16 parameters and field, fp32 log-det) at batch 8.  Forward + log|det J|,
    no_grad, synthetic inputs, one net per block, each: 1 warm-up + `reps` timed passes between synchronisations."""
    import torch
    from normflow__amd.nn import ConvAct, RQSplineCoupling_, AffineCoupling_, ModuleList_
    from normflow__amd.mask import EvenOddMask

    def build(shape, kinds, dtype):
        d = len(shape)
        mask = EvenOddMask(shape=shape)
        lim = dict(xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
        blocks = []
        for kind in kinds:
            net = ConvAct(1, 46 if kind == 'rqs' else 2, 3, conv_dim=d, hidden_sizes=[8, 8], acts=['tanh', 'tanh', None])
            with torch.no_grad():
                for p in list(net.parameters())[-2:]:
                    p.mul_(0.3)
            blocks.append(RQSplineCoupling_([net], mask=mask, **lim) if kind == 'rqs' else AffineCoupling_([net], mask=mask))
        net_ = ModuleList_(blocks)
        net_.to(device=dev, dtype=dtype)
        return net_

    out = {}
    for name, shape, kinds, B, dtype, reps in (
            ("config2_16x16_4affine_b512", (16, 16), ['affine'] * 4, 512, torch.float32, 20),
            ("config3_16x16x16_8rqs_b1024", (16, 16, 16), ['rqs'] * 8, 1024, torch.float32, 10),
            ("config5_48x48x48x48_8affine8rqs_fp16storage_b8", (48,) * 4, ['affine', 'rqs'] * 8, 8, torch.float16, 2)):
        torch.manual_seed(0)
        net_ = build(shape, kinds, dtype)
        x = torch.randn((B,) + shape, device=dev, dtype=torch.float32).to(dtype)
        with torch.no_grad():
            y, lj = net_(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                y, lj = net_(x)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        assert bool(torch.isfinite(lj).all()) and bool(torch.isfinite(y.float()).all())
        out[name] = {"configs_per_s": B / dt, "ms_per_step": 1e3 * dt, "batch": B, "layers": len(kinds)}
        del net_, x, y, lj
    return out


# ------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        sys.exit(self_launch(a))           # the parent never touches the GPU
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        print(f"bench.py: --gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr)
        sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the hot path)"
    # NF_BENCH_REHEARSAL=1: dry run of the multi-rank code path on a box with fewer GPUs than ranks
    # (all ranks share cuda:0, gloo instead of RCCL).  Never used for reported numbers.
    rehearsal = os.environ.get("NF_BENCH_REHEARSAL", "0") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local)
    torch.cuda.set_device(dev)
    if a.scaling in ("strong", "both") and world > 1 and a.batch % world:
        # refuse before any rank touches the GPU: a strong-scaling leg cuts ONE global batch into equal shards
        print(f"bench.py: the strong-scaling leg cuts the global batch {a.batch} into {world} equal shards, and {a.batch} is not "
              f"a multiple of {world}; pass --batch as a multiple of --gpus or --scaling weak", file=sys.stderr)
        sys.exit(2)
    ranks_info = None
    if world > 1:
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)   # "nccl" = RCCL
        assert dist.get_world_size() == a.gpus
        mine = {"rank": rank, "device": f"cuda:{dev.index}", "name": torch.cuda.get_device_name(dev)}
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        ranks_info = {"rccl_world_size": dist.get_world_size(), "backend": dist.get_backend(), "ranks": gathered}
    import normflow__amd  # noqa: F401
    from normflow__amd import _hip
    lattice = tuple(int(s) for s in a.lattice.split(","))
    net_, cpl = build_net(lattice, a.layers, a.knots, dev, seed=2024)       # same weights on all ranks
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((a.batch,) + lattice, device=dev, dtype=torch.float32, generator=g)

    def timed(xb, steps, warmup):
        """W untimed + exactly K timed passes over xb, barrier + synchronize on both sides, MAX over ranks."""
        y = logJ = None
        with torch.no_grad():
            for _ in range(warmup):
                y, logJ = net_(xb)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                y, logJ = net_(xb)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        assert bool(torch.isfinite(logJ).all()) and bool(torch.isfinite(y).all())
        return elapsed

    # ---- before timing: the timed arithmetic against exact fp32 products on a slab of the timed input
    selfcheck = None
    if not a.no_selfcheck:
        nb = min(4, a.batch)
        with torch.no_grad():
            y1, l1 = net_(x[:nb])
            split_ran = _hip.load().nf_conv_last_path() == 3
            with _hip.options(split16=False):
                y0, l0 = net_(x[:nb])
        ey = float((y1 - y0).abs().max() / y0.abs().max().clamp_min(1.0))
        el = float(((l1 - l0).abs() / l0.abs().clamp_min(1.0)).max())
        selfcheck = {"samples": nb, "split16_kernels_ran": bool(split_ran), "max_rel_y_vs_fp32_products": ey,
                     "max_rel_logJ_vs_fp32_products": el, "nonzero_fraction_y": float((y1 != 0).float().mean())}
        # a did-the-work check, not the parity bound (tests/test_gpu_parity.py::test_headline_network_* hold that): two fp32-level
        # arithmetics drift apart through 8 stacked layers by the layers' own sensitivity (the reference's fp32
        # run of a whole 4-D net is 6e-5 from its fp64 run on y)
        assert ey < 2e-3 and el < 1e-4 and selfcheck["nonzero_fraction_y"] > 0.99, selfcheck
        del y1, y0

    # ---- allocator priming (not warm-up steps, not timed): the caching allocator reaches its steady state only in the second full
    # pass -- the outputs of pass 1 are alive during pass 2 and one of its 4 GiB requests splits a cached 32 GiB block, so pass 2
    # hipMallocs a third 32 GiB segment, ~1 s in the first process on a freshly booted box (tools/first_pass_diag.py).  With
    # --warmup 1 that second was landing inside the timed region.
    with torch.no_grad():
        keep = None
        for _ in range(2):
            keep = net_(x)
        torch.cuda.synchronize()
        del keep

    legs = {}
    if a.scaling in ("weak", "both") or world == 1:
        el = timed(x, a.steps, a.warmup)
        legs["weak"] = {"value": a.batch * world * a.steps / el, "ms_per_step": 1e3 * el / a.steps, "global_batch": a.batch * world,
                        "batch_per_gpu": a.batch}
    if a.scaling in ("strong", "both"):
        if world == 1 and "weak" in legs:
            legs["strong"] = dict(legs["weak"])
        else:
            per = a.batch // world           # divisibility was checked before the process group came up
            el = timed(x[:per], a.steps, a.warmup)
            legs["strong"] = {"value": a.batch * a.steps / el, "ms_per_step": 1e3 * el / a.steps, "global_batch": a.batch,
                              "batch_per_gpu": per}
    main_leg = "strong" if a.scaling == "strong" else "weak"
    inloop = inloop_kernel_ms(net_, x if main_leg == "weak" else x[:a.batch // world]) if rank == 0 else None
    fp32_leg = None
    if not a.no_fp32_products:
        k2 = max(1, min(a.steps, 2))
        xb = x if main_leg == "weak" else x[:a.batch // world]
        with _hip.options(split16=False):
            el = timed(xb, k2, 1)
        fp32_leg = {"value": xb.shape[0] * world * k2 / el, "ms_per_step": 1e3 * el / k2, "steps": k2, "warmup": 1}

    if rank == 0:
        # Which kernels the timed pipeline runs: with this package's ConvAct on a plain even-odd
        # mask the last conv layer and the spline are ONE kernel (nf_conv_rqs); otherwise the
        # stand-alone coupling kernel consumes a materialised logit tensor.
        with torch.no_grad():
            probe = torch.zeros((1,) + lattice, device=dev, dtype=torch.float32)
            fused = cpl._fused_atom(False, probe, probe, 0, cpl.nets[0], 0) is not None
            pipeline_pair = cpl._params(cpl.nets[0], probe, parity=0)[1] == 1
        kt = time_rqs_kernel(cpl, lattice, a.knots, dev, a.kernel_reps, pipeline_pair)
        traffic, traffic_src = profiled_traffic("rqs_kernel", kt["slab"], lattice, a.knots)
        hbm_obj = {"kernel": "nf::rqs_kernel<float,16,fwd>" + ("<pair>" if pipeline_pair else "<full>") +
                             " (stand-alone RQ-spline coupling kernel)",
                   "bound": "hbm", "achieved": kt["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": kt["gbs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                   "launch_ms": 1e3 * kt["seconds"], "slab_batch": kt["slab"],
                   "algorithmic_bytes_per_launch": kt["algo_bytes"],
                   "in_timed_pipeline": not fused}
        others = []
        if fused:
            reps = max(2, a.kernel_reps // 3)
            ft = time_fused_last_layer(cpl, lattice, a.knots, dev, reps, a.batch)
            split16 = ft["path"] == 3
            if split16:
                # K5h: every fp32 product = three fp16 MFMA products (fp32 accumulate).  `achieved` stays the ALGORITHMIC
                # (fp32-equivalent) rate; the peak that bounds it is the fp16 matrix peak / 3.  Executed fp16 flops per
                # algorithmic flop: 3 products x 84/81 (K slices padded to 4 kernel rows) x 48/46 (columns padded to 3 tiles).
                peak = MFMA_F16_PEAK_TFLOPS / 3.0
                h_traffic, h_src = profiled_traffic("conv_h_kernel", ft["slab"], lattice, a.knots)
                il = inloop.get("nf_conv_rqs") if inloop else None
                if il is not None:                       # the duration inside the timed loop; the burst figure stays beside it
                    ft = dict(ft, burst_seconds=ft["seconds"], seconds=1e-3 * il["ms"], tflops=ft["flops"] / (1e-3 * il["ms"]) / 1e12,
                              launches=il["launches"])
                executed = ft["tflops"] * 3.0 * (84.0 / 81.0) * (48.0 / 46.0)
                roof = {"kernel": "nf::conv_h_kernel<fwd> (last conv layer 8->46 at the active sites + RQ-spline coupling epilogue, "
                                  "fp32 products as 3 x v_mfma_f32_16x16x32_f16; dominant kernel of the timed region)",
                        "bound": "mfma", "achieved": ft["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": ft["tflops"] / peak, "traffic": h_traffic, "traffic_source": h_src,
                        "launch_ms": 1e3 * ft["seconds"], "slab_batch": ft["slab"],
                        "algorithmic_flops_per_launch": ft["flops"],
                        "peak_note": "dense fp16 MFMA peak 2500 TFLOP/s / 3 fp16 products per fp32 product",
                        "executed_fp16_tflops": executed, "executed_frac_of_fp16_peak": executed / MFMA_F16_PEAK_TFLOPS,
                        "vs_fp32_mfma_peak": ft["tflops"] / MFMA_F32_PEAK_TFLOPS}
                if "burst_seconds" in ft:
                    roof["launch_ms_source"] = (f"HIP events around each of the {ft['launches']} launches of one more pass of the timed loop, "
                                                "right after the timed steps")
                    roof["launch_ms_burst"] = 1e3 * ft["burst_seconds"]
            else:
                p_traffic, p_src = profiled_traffic("conv_pipe_kernel_fused", ft["slab"], lattice, a.knots)
                roof = {"kernel": "nf::conv_pipe_kernel<2,3,3,compact,fused-rqs-fwd,wide,unrolled> (last conv layer 8->46 at the active sites "
                                  "+ RQ-spline coupling epilogue; dominant kernel of the timed region)",
                        "bound": "mfma", "achieved": ft["tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ft["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": p_traffic, "traffic_source": p_src,
                        "launch_ms": 1e3 * ft["seconds"], "slab_batch": ft["slab"],
                        "algorithmic_flops_per_launch": ft["flops"]}
            others = time_hidden_layers(cpl, lattice, dev, reps, a.batch)
            for o, key, entry in zip(others, ("conv_c2_kernel", "conv_g2_kernel"), ("nf_conv_first_split16", "nf_conv_fwd_split16")):
                o["traffic"], o["traffic_source"] = profiled_traffic(key, o["slab_batch"], lattice, a.knots)
                il = inloop.get(entry) if (inloop and split16) else None
                if il is not None:
                    o["launch_ms_burst"] = o["launch_ms"]
                    o["achieved"] *= o["launch_ms"] / il["ms"]
                    o["frac"] = o["achieved"] / o["peak"]
                    o["launch_ms"] = il["ms"]
                    o["launch_ms_source"] = f"HIP events around each of the {il['launches']} launches of one more pass of the timed loop"
        else:
            roof = hbm_obj
        lead = legs[main_leg]
        line = {
            "metric": "lattice configs/sec (forward+logdet)", "value": lead["value"], "unit": "configs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": lead["ms_per_step"],
            "higher_is_better": True, "scaling": main_leg, "vs_baseline": None, "dtype": "f32",
            "arithmetic": "fp32 in/out; conv products as three fp16 matrix-core products with fp32 accumulation where the inputs are tanh outputs (1e-5 rel. vs the fp64 oracle), fp32 MFMA otherwise",
            "data": "synthetic",
            "config": {"workload": f"{'x'.join(map(str, lattice))} phi^4 lattice, {a.layers} RQ-spline coupling "
                                   f"layers (knots_len {a.knots}, ConvAct 1-8-8-{3*a.knots-2}, k=3, tanh), "
                                   f"batch {lead['batch_per_gpu']} per GPU, forward + log|det J|, no_grad",
                       "global_batch": lead["global_batch"], "parallelism": f"dp{world}"},
            "roofline": roof,
            "roofline_kernels": others,
            "roofline_hbm_kernel": hbm_obj,
            "kernel_src_sha": kernel_src_sha(),
        }
        if ranks_info is not None:
            line.update(ranks_info)
        for name, leg in legs.items():
            if name != main_leg:
                line[name] = dict(leg, scaling=name)
        if fp32_leg is not None:
            line["value_fp32_products"] = fp32_leg["value"]
            line["fp32_products"] = dict(fp32_leg, arithmetic="exact fp32 MFMA products everywhere (nf_set_option(NF_OPT_SPLIT16, 0)); same network, same input")
        if selfcheck is not None:
            line["selfcheck"] = selfcheck
        if not a.no_other_configs and world == 1:
            line["other_configs"] = other_configs(dev)
        if not a.no_cpu_baseline and world == 1:     # CPU baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(cpl, lattice, a.knots, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
