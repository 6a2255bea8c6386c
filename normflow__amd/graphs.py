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


class GraphedTrainStep:
    """Forward pass, loss and backward pass of one reverse-KL step (Fitter.step; reference src/_normflowcore.py:275-294)
    as ONE HIP graph.  On the small lattices flows are usually trained on (16^2, 16^3) a step is several hundred kernels of
    a few microseconds each: the host's launch path, not the GPU, sets the step time.

        step = GraphedTrainStep(model, loss_fn, batch_size)
        x, logr = model.prior.sample_(batch_size)        # eager: the generator's state is host-side
        loss, logqp = step(x, logr)                      # .grad of every parameter now holds this step's gradient
        optimizer.step()

    * the random draw, the gradient all-reduce and the optimiser stay outside the graph;
    * weight repacking is captured with the pass (`_hip.pack_inside_capture`), so a replay always uses the current values;
    * the kernel choice frozen into the graph (split-fp16 products need |w| < 29) is guarded: the graph also computes
      max |parameter|, read back with the loss; outside the validated range the graph is captured again on the exact
      fp32 kernels (valid for any weights) and the step repeated;
    * the parameters' .grad tensors live in the graph's memory pool and are overwritten by every replay: do not call
      `optimizer.zero_grad()` between replay and `optimizer.step()` (Fitter does not, in this mode).
    Results are those of the eager step, bit for bit (same kernels, same order)."""

    def __init__(self, model, loss_fn, batch_size, warmup=2):
        self._model, self._loss_fn, self._B, self._warmup = model, loss_fn, int(batch_size), warmup
        self._params = [p for p in model.net_.parameters() if p.requires_grad]
        if not self._params or not self._params[0].is_cuda:
            raise ValueError("GraphedTrainStep needs a model with trainable parameters on a CUDA/HIP device")
        self._guard = True                            # the captured kernel choice holds for |w| below the limit only
        self._capture(split16=True)

    def _body(self):
        m = self._model
        y, logj = m.net_(self._x)
        logq, logp = self._logr - logj, -m.action(y)
        return self._loss_fn(logq, logp), logq - logp

    def _capture(self, split16):
        from . import _hip
        if split16:
            return self._capture_now()
        with _hip.options(split16=False):             # exact fp32 products everywhere: valid for any weights
            return self._capture_now()

    def _capture_now(self):
        from . import _hip
        m = self._model
        dev = self._params[0].device
        rng = torch.cuda.get_rng_state(dev)           # (the example draw must not shift the training run's random stream)
        with torch.no_grad():
            x, logr = m.prior.sample_(self._B)
        torch.cuda.set_rng_state(rng, dev)
        self._x, self._logr = x.detach().clone(), logr.detach().clone()
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(self._warmup):             # one-time initialisation and the host-side weight checks, eagerly
                for p in self._params:
                    p.grad = None
                loss, _ = self._body()
                loss.backward()
        torch.cuda.current_stream(dev).wait_stream(side)
        for p in self._params:
            p.grad = None
        self._graph = torch.cuda.CUDAGraph()
        with _hip.pack_inside_capture(), torch.cuda.graph(self._graph):
            loss, d = self._body()
            loss.backward()
            wmax = torch.stack(torch._foreach_norm([p.detach() for p in self._params], float('inf'))).max()
        self._loss, self._d, self._wmax = loss.detach(), d.detach(), wmax
        self._grads = [p.grad for p in self._params]
        self._limit = 2.9e4 / _hip.SPLIT16_WEIGHT_SCALE

    def __call__(self, x, logr):
        if tuple(x.shape) != tuple(self._x.shape) or x.dtype != self._x.dtype:
            raise ValueError(f"GraphedTrainStep was captured for {tuple(self._x.shape)} {self._x.dtype}, "
                             f"got {tuple(x.shape)} {x.dtype}")
        for attempt in range(2):
            self._x.copy_(x)
            self._logr.copy_(logr)
            for p, g in zip(self._params, self._grads):
                p.grad = g
            self._graph.replay()
            if not self._guard:
                break
            wmax = float(self._wmax)                  # (the caller reads the loss next: this is the step's one sync)
            if wmax < self._limit:
                break
            # NaN / inf / large weights: not the range the capture validated.  Capture again on the fp32 kernels, which take
            # any weights (so the guard retires), and repeat the step
            self._capture(split16=False)
            self._guard = False
        return self._loss.clone(), self._d.clone()
