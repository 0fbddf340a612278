"""hipGraph replay of a whole training step (forward + backward + optimizer update).

The loss kernels take microseconds; an eager PyTorch step around them costs 100+ us of Python,
autograd-engine and launch overhead.  For fixed batch shapes the step can be captured once into a
hipGraph and replayed: ``GraphedStep`` owns static input buffers, copies each new batch into them
and replays the captured work -- kernels of this package are capture-safe (a fixed number of
launches per call, no allocation, no host synchronisation).

    step = GraphedStep(model, optimizer, lambda xs, ys, n: loss_fn(model(xs), ys, n).mean(),
                       example_batch=(xs, ys, n))
    for xs, ys, n in batches:            # same shapes as the example
        loss = step(xs, ys, n)           # a 0-dim tensor that is overwritten by the next call

The reference has nothing comparable (it runs eagerly on the CPU); this is the "HIP graphs
instead of a tracing compiler" part of the MI355X design.
"""
from typing import Callable, Sequence

import torch


class GraphedStep:
    """Captures ``loss = loss_closure(*batch); loss.backward(); optimizer.step()`` into a hipGraph.

    Args:
        params_owner: the ``torch.nn.Module`` (or an iterable of parameters) being trained.
        optimizer: a torch optimizer; stateful ones that keep a step counter (Adam, ...) must be built with
            ``capturable=True`` so that their step counters live on the device.
        loss_closure: maps the static batch tensors to a scalar loss.
        example_batch: tensors with the shapes / dtypes / device of every later batch.
        warmup: eager iterations run on a side stream before the capture (allocator, lazy inits,
            one-time kernel attribute calls).  They DO update the parameters.
    """

    def __init__(self, params_owner, optimizer: torch.optim.Optimizer,
                 loss_closure: Callable[..., torch.Tensor], example_batch: Sequence[torch.Tensor],
                 warmup: int = 3):
        self._params = list(params_owner.parameters() if hasattr(params_owner, "parameters")
                            else params_owner)
        self._optimizer = optimizer
        self._closure = loss_closure
        self._static = [t.clone() for t in example_batch]
        if not self._static or not self._static[0].is_cuda:
            raise ValueError("GraphedStep needs device tensors")
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
            self._loss = self._eager_step()
        torch.cuda.synchronize()

    def _eager_step(self):
        self._optimizer.zero_grad(set_to_none=True)
        loss = self._closure(*self._static)
        loss.backward()
        self._optimizer.step()
        return loss.detach()

    def __call__(self, *batch: torch.Tensor) -> torch.Tensor:
        if len(batch) != len(self._static):
            raise ValueError("expected %d tensors, got %d" % (len(self._static), len(batch)))
        for dst, src in zip(self._static, batch):
            if dst.shape != src.shape:
                raise ValueError("batch shape %s differs from the captured shape %s (pad batches "
                                 "to a fixed list size)" % (tuple(src.shape), tuple(dst.shape)))
            dst.copy_(src, non_blocking=True)
        self._graph.replay()
        return self._loss
