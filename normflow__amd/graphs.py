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
        self._net, self._inverse, self._warmup = net_, inverse, warmup
        self._capture()

    def _param_state(self):
        return tuple((p.data_ptr(), p._version) for p in self._net.parameters())

    def _capture(self):
        fn = self._net.backward if self._inverse else self._net.forward
        call = (lambda: fn(self._x)) if self._log0 is None else (lambda: fn(self._x, self._log0))
        dev = self._x.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(self._warmup):             # one-time initialisation (function attributes, the host-side range
                call()                                # checks of the split-fp16 kernels) outside the capture
        torch.cuda.current_stream(dev).wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self._graph):
            self._y, self._logj = call()
        self._state = self._param_state()

    def __call__(self, x, log0=None, clone=True):
        if x.shape != self._x.shape or x.dtype != self._x.dtype:
            raise ValueError(f"GraphedFlow was captured for {tuple(self._x.shape)} {self._x.dtype}, got {tuple(x.shape)} {x.dtype}")
        if self._param_state() != self._state:
            # the parameters changed since the capture (optimizer step, load_state_dict): the kernel choice frozen into
            # the graph (split-fp16 products need weights inside the fp16 range) was validated for the OLD values
            self._capture()
        self._x.copy_(x)
        if self._log0 is not None:
            if log0 is None:
                raise ValueError("this graph was captured with a log0 input")
            self._log0.copy_(log0)
        self._graph.replay()
        if clone:
            return self._y.clone(), self._logj.clone()
        return self._y, self._logj
