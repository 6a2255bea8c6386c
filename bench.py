#!/usr/bin/env python3
"""bench.py -- lattice configs/sec of forward + log|det J| (Posterior.sample_ minus the prior
draw) on BASELINE.json's headline workload: 32^4 phi^4 lattice, 8 RQ-spline coupling layers
(knots_len 16, ConvAct hidden [8,8], kernel 3, tanh), batch 1024 per GPU, fp32, synthetic
inputs already resident in HBM.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU; the batch shards over ranks with no data-path collective (forward /
sampling needs none), so scaling is weak: every rank runs the full per-GPU batch.  Rank 0
prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     : the fused RQ-spline coupling kernel (HBM-bound), HIP-event timed in here
  cpu_baseline : the CPU oracle (oracle/nf_oracle.py, kind "port") on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("NORMFLOW_AMD_KEEP_TORCH_DEFAULTS", "1")   # bench manages dtype/device itself

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense f32-input MFMA peak (= vector f32 peak)
MFMA_F16_PEAK_TFLOPS = 2500.0  # dense fp16 / bf16 MFMA peak (task statement: ~2.5 PFLOP/s; 16x the f32-input rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1024, help="batch per GPU")
    ap.add_argument("--lattice", type=str, default="32,32,32,32")
    ap.add_argument("--layers", type=int, default=8)
    ap.add_argument("--knots", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=25.0, help="budget of the CPU baseline sample")
    ap.add_argument("--kernel-reps", type=int, default=10)
    return ap.parse_args()


def build_net(lattice, layers, m, dev, seed):
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


def time_rqs_kernel(cpl, lattice, m, dev, reps, layout_pair):
    """Average duration of one nf_rqs_fwd launch on the slab shape the pipeline uses; HIP events
    on the launch stream (the kernels are launched on torch's current stream)."""
    from normflow__amd import _hip
    from normflow__amd.nn.scalar import couplings_ as cp
    V = 1
    for n in lattice:
        V *= n
    C = 3 * m - 2
    Vp = V // 2 if layout_pair else V
    slab = max(1, min(64, cp.PARAM_SLAB_BYTES // (C * V * 4)))
    act = cpl.mask.activity(0).reshape(-1).to(dev)
    g = torch.Generator(device=dev).manual_seed(99)
    x = torch.randn(slab, V, device=dev, dtype=torch.float32, generator=g) * act.float()
    params = 0.5 * torch.randn(slab, C, Vp, device=dev, dtype=torch.float32, generator=g)
    opts = _hip.make_rqs_opts(m, (-5.0, 5.0), (-5.0, 5.0), {'left': 'linear', 'right': 'linear'},
                              _hip.LAYOUT_PAIR if layout_pair else _hip.LAYOUT_FULL)
    for _ in range(2):
        _hip.RQSCouplingFn.apply(x, params, None, act, opts, False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        _hip.RQSCouplingFn.apply(x, params, None, act, opts, False)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    algo_bytes = slab * (V // 2) * (C + 2) * 4          # B*(V/2)*(C+2)*sizeof
    return dict(seconds=sec, slab=slab, algo_bytes=algo_bytes, gbs=algo_bytes / sec / 1e9)


def time_fused_last_layer(cpl, lattice, m, dev, reps, batch):
    """Average duration of one nf_conv_rqs launch (last conv layer 8 -> 3m-2 at the active sites
    + fused RQ-spline epilogue) on the slab shape the pipeline uses; HIP events on the launch
    stream.  Algorithmic flops: 2 * 3^d * cin * cout per ACTIVE site (last layer)."""
    from normflow__amd import _hip
    V = 1
    for n in lattice:
        V *= n
    net = cpl.nets[0]
    hidden = max(net.conv_kwargs['hidden_sizes'])
    slab = max(1, min(batch, cpl.HIDDEN_SLAB_BYTES // (hidden * V * 4)))   # the pipeline's own slab
    last = [mod for mod in net if any(True for _ in mod.parameters())][-1]
    g = torch.Generator(device=dev).manual_seed(98)
    h = torch.tanh(torch.randn((slab, hidden) + tuple(lattice), device=dev, dtype=torch.float32, generator=g))
    x = torch.randn(slab, V, device=dev, dtype=torch.float32, generator=g)
    opts = _hip.make_rqs_opts(m, (-5.0, 5.0), (-5.0, 5.0), {'left': 'linear', 'right': 'linear'}, _hip.LAYOUT_PAIR)
    w, b = last.weight.detach(), last.bias.detach()
    f = lambda: _hip.conv_rqs(h, w, b, x, None, 0, opts, False, unit_input=True)    # h = tanh(...): |h| <= 1
    got = net.hidden_and_last(torch.zeros((1, 1) + tuple(lattice), device=dev, dtype=torch.float32))
    if got is not None and got[3]:      # the pipeline hands the kernel fp16 (hi, lo) pairs, channel-last: time that form
        hp = h.reshape(slab, hidden, V).permute(0, 2, 1).contiguous()
        hi = hp.half()
        h16 = torch.cat((hi, (hp - hi.float()).half()), dim=2).contiguous()
        del hp, hi, h
        f = lambda: _hip.conv_rqs(h16, w, b, x, None, 0, opts, False, unit_input=True, lattice=tuple(lattice))
    for _ in range(2):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / reps
    flops = 2.0 * (3 ** len(lattice)) * hidden * (3 * m - 2) * (V // 2) * slab
    return dict(seconds=sec, slab=slab, flops=flops, tflops=flops / sec / 1e12)


def cpu_baseline(cpl, lattice, m, budget_s):
    """The CPU oracle restatement (plain PyTorch CPU ops, all host threads, fp32) on a bounded
    sample of the same workload: whole coupling layers of the bench network at batch 1."""
    from oracle import nf_oracle as O
    # the GPU box gives one GPU's job a 16-CPU share (task statement); more threads than that
    # only oversubscribes (256 threads ran 10x slower than 8 in the build container)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))
    torch.set_num_threads(nthreads)
    nets = []
    for net in cpl.nets:
        convs = [mod for mod in net if any(True for _ in mod.parameters())]
        layers = []
        for c in convs:
            w = c.weight.detach().float().cpu().contiguous()
            layers.append((w, c.bias.detach().float().cpu()))
        nets.append(lambda t, layers=layers: O.conv_act(t, layers, ['tanh', 'tanh', None]))
    g = torch.Generator(device='cpu').manual_seed(4321)
    x = torch.randn((1,) + tuple(lattice), generator=g, dtype=torch.float32, device='cpu')
    opts = dict(xlim=(-5.0, 5.0), ylim=(-5.0, 5.0), extrap={'left': 'linear', 'right': 'linear'})
    done, t0 = 0, time.time()
    with torch.no_grad():
        masks = [O.channel_mask(lattice, c, dtype=torch.float32) for c in (0, 1)]
        parts = [x * masks[0], x * masks[1]]
        log0 = 0
        for k, net in enumerate(nets):
            p = k % 2
            out = net(parts[1 - p].unsqueeze(1))
            parts[p], log0 = O.rqs_coupling_atom(parts[p], out, masks[p], log0=log0, **opts)
            done += 1
            spent = time.time() - t0
            if spent + spent / done > budget_s:     # the next layer would overrun the budget
                break
    dt = time.time() - t0
    per_cfg = dt / done * len(nets)
    return dict(value=1.0 / per_cfg, unit="configs/s", cores=nthreads, kind="port",
                sample=f"{done} of {len(nets)} coupling layers (conv stack + RQ spline + log-det) of the bench "
                       f"network at batch 1, fp32, torch CPU ops on {nthreads} threads, {dt:.1f} s; "
                       f"extrapolated linearly to {len(nets)} layers")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the hot path)"
    # NF_BENCH_REHEARSAL=1: dry run of the multi-rank code path on a box with fewer GPUs than ranks
    # (all ranks share cuda:0, gloo instead of RCCL).  Never used for reported numbers.
    rehearsal = os.environ.get("NF_BENCH_REHEARSAL", "0") == "1"
    dev = torch.device("cuda", 0 if rehearsal else local)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group(backend="gloo" if rehearsal else "nccl", rank=rank, world_size=world)   # "nccl" = RCCL
    import normflow__amd  # noqa: F401
    lattice = tuple(int(s) for s in a.lattice.split(","))
    net_, cpl = build_net(lattice, a.layers, a.knots, dev, seed=2024)       # same weights on all ranks
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn((a.batch,) + lattice, device=dev, dtype=torch.float32, generator=g)

    def step():
        with torch.no_grad():
            return net_(x)

    for _ in range(a.warmup):
        y, logJ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y, logJ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device="cpu" if rehearsal else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    assert bool(torch.isfinite(logJ).all()) and bool(torch.isfinite(y).all())

    if rank == 0:
        # Which kernels the timed pipeline runs: with this package's ConvAct on a plain even-odd
        # mask the last conv layer and the spline are ONE kernel (nf_conv_rqs); otherwise the
        # stand-alone coupling kernel consumes a materialised logit tensor.
        with torch.no_grad():
            probe = torch.zeros((1,) + lattice, device=dev, dtype=torch.float32)
            fused = cpl._fused_atom(False, probe, probe, 0, cpl.nets[0], 0) is not None
            pipeline_pair = cpl._params(cpl.nets[0], probe, parity=0)[1] == 1
        kt = time_rqs_kernel(cpl, lattice, a.knots, dev, a.kernel_reps, pipeline_pair)
        # HBM traffic of one launch of the stand-alone coupling kernel from the committed PMC
        # profile (rocprofv3 cannot collect counters from inside the bench); only quoted when the
        # launch shape is the profiled one
        traffic, traffic_src = None, None
        try:
            prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_rqs.json")))
            if pipeline_pair and prof["slab_batch"] == kt["slab"] and prof["algorithmic_bytes_per_launch"] == kt["algo_bytes"]:
                traffic, traffic_src = prof["hbm_bytes_per_launch"], "profiles/r01_pmc_rqs.json"
        except (OSError, KeyError, ValueError):
            pass
        hbm_obj = {"kernel": "nf::rqs_kernel<float,16,fwd>" + ("<pair>" if pipeline_pair else "<full>") +
                             " (stand-alone RQ-spline coupling kernel)",
                   "bound": "hbm", "achieved": kt["gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                   "frac": kt["gbs"] / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                   "launch_ms": 1e3 * kt["seconds"], "slab_batch": kt["slab"],
                   "algorithmic_bytes_per_launch": kt["algo_bytes"],
                   "in_timed_pipeline": not fused}
        if fused:
            ft = time_fused_last_layer(cpl, lattice, a.knots, dev, max(2, a.kernel_reps // 3), a.batch)
            # fabric traffic of one launch from the committed PMC profile (separate --pmc passes), quoted only for
            # the profiled lattice and layer and scaled to this launch's slab
            ftraffic, ftraffic_src = None, None
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_conv.json")))
                k = prof["kernels"]["void nf::conv_pipe_kernel<2, 3, 3, true, 1, true, true>(nf::ConvArgs)"]
                if lattice == (32, 32, 32, 32) and a.knots == 16:
                    ftraffic = k["traffic_bytes_per_launch"] * ft["slab"] / prof["slab_batch"]
                    ftraffic_src = "profiles/r01_pmc_conv.json"
            except (OSError, KeyError, ValueError):
                pass
            from normflow__amd import _hip as _h
            split16 = _h.load().nf_conv_last_path() == 3
            if split16:
                # K5h: every fp32 product = three fp16 MFMA products (fp32 accumulate).  `achieved` stays the ALGORITHMIC
                # (fp32-equivalent) rate; the peak that bounds it is the fp16 matrix peak / 3.  Executed fp16 flops per
                # algorithmic flop: 3 products x 84/81 (K slices padded to 4 kernel rows) x 48/46 (columns padded to 3 tiles).
                peak = MFMA_F16_PEAK_TFLOPS / 3.0
                h_traffic, h_traffic_src = None, None
                try:
                    prof = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_conv.json")))["split16_kernels"]
                    if lattice == (32, 32, 32, 32) and a.knots == 16:
                        h_traffic = (prof["kernels"]["void nf::conv_h_kernel<1>(nf::ConvArgs)"]["traffic_bytes_per_launch"]
                                     * ft["slab"] / prof["slab_batch"])
                        h_traffic_src = "profiles/r01_pmc_conv.json (split16_kernels)"
                except (OSError, KeyError, ValueError):
                    pass
                executed = ft["tflops"] * 3.0 * (84.0 / 81.0) * (48.0 / 46.0)
                roof = {"kernel": "nf::conv_h_kernel<fwd> (last conv layer 8->46 at the active sites + RQ-spline coupling epilogue, "
                                  "fp32 products as 3 x v_mfma_f32_16x16x32_f16; dominant kernel of the timed region)",
                        "bound": "mfma", "achieved": ft["tflops"], "peak": peak, "unit": "TFLOP/s",
                        "frac": ft["tflops"] / peak, "traffic": h_traffic, "traffic_source": h_traffic_src,
                        "launch_ms": 1e3 * ft["seconds"], "slab_batch": ft["slab"],
                        "algorithmic_flops_per_launch": ft["flops"],
                        "peak_note": "dense fp16 MFMA peak 2500 TFLOP/s / 3 fp16 products per fp32 product",
                        "executed_fp16_tflops": executed, "executed_frac_of_fp16_peak": executed / MFMA_F16_PEAK_TFLOPS,
                        "vs_fp32_mfma_peak": ft["tflops"] / MFMA_F32_PEAK_TFLOPS}
            else:
                roof = {"kernel": "nf::conv_pipe_kernel<2,3,3,compact,fused-rqs-fwd,wide,unrolled> (last conv layer 8->46 at the active sites "
                                  "+ RQ-spline coupling epilogue; dominant kernel of the timed region)",
                        "bound": "mfma", "achieved": ft["tflops"], "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": ft["tflops"] / MFMA_F32_PEAK_TFLOPS, "traffic": ftraffic, "traffic_source": ftraffic_src,
                        "launch_ms": 1e3 * ft["seconds"], "slab_batch": ft["slab"],
                        "algorithmic_flops_per_launch": ft["flops"]}
        else:
            roof = hbm_obj
        cfgs = a.batch * world * a.steps
        line = {
            "metric": "lattice configs/sec (forward+logdet)", "value": cfgs / elapsed, "unit": "configs/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "arithmetic": "fp32 in/out; conv products as three fp16 matrix-core products with fp32 accumulation where the inputs are tanh outputs (1e-5 rel. vs the fp64 oracle), fp32 MFMA otherwise",
            "data": "synthetic",
            "config": {"workload": f"{'x'.join(map(str, lattice))} phi^4 lattice, {a.layers} RQ-spline coupling "
                                   f"layers (knots_len {a.knots}, ConvAct 1-8-8-{3*a.knots-2}, k=3, tanh), "
                                   f"batch {a.batch} per GPU, forward + log|det J|, no_grad",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}"},
            "roofline": roof,
            "roofline_hbm_kernel": hbm_obj,
        }
        if not a.no_cpu_baseline and world == 1:     # CPU baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(cpl, lattice, a.knots, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
