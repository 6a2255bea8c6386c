"""hipGraph replay for launch-bound inference (small lattices).

On a 16x16 lattice a coupling block is a dozen kernels of a few microseconds each; the step time is the
host's launch path, not the GPU.  `GraphedFlow` captures one no_grad pass of a flow (forward or backward)
into a HIP graph -- every kernel of this package launches on torch's current stream and allocates nothing
itself, so the pass is capturable as is -- and replays it with one host call per batch.

    fast = GraphedFlow(model.net_, example_x)          # example_x fixes batch size, shape, dtype
    y, logJ = fast(x)                                   # same results as model.net_(x), bitwise

The reference has no counterpart (it runs eagerly); this is an MI355X-side convenience, not part of the
normflow API surface.
"""
import torch


class GraphedFlow:
    def __init__(self, net_, example_x, inverse=False, log0=None, warmup=2):
        if not example_x.is_cuda:
            raise ValueError("GraphedFlow needs a CUDA/HIP tensor")
        self._x = example_x.detach().clone()
        self._log0 = None if log0 is None else log0.detach().clone()
        fn = net_.backward if inverse else net_.forward
        call = (lambda: fn(self._x)) if self._log0 is None else (lambda: fn(self._x, self._log0))
        side = torch.cuda.Stream(device=example_x.device)
        side.wait_stream(torch.cuda.current_stream(example_x.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(warmup):                   # one-time initialisation (function attributes, workspaces) outside the capture
                call()
        torch.cuda.current_stream(example_x.device).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._y, self._logj = call()

    def __call__(self, x, log0=None, clone=True):
        if x.shape != self._x.shape or x.dtype != self._x.dtype:
            raise ValueError(f"GraphedFlow was captured for {tuple(self._x.shape)} {self._x.dtype}, got {tuple(x.shape)} {x.dtype}")
        self._x.copy_(x)
        if self._log0 is not None:
            if log0 is None:
                raise ValueError("this graph was captured with a log0 input")
            self._log0.copy_(log0)
        self._graph.replay()
        if clone:
            return self._y.clone(), self._logj.clone()
        return self._y, self._logj
