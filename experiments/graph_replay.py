"""HIP-graph replay of a feature module's forward for a fixed input shape.

A forward of the octave transforms is a chain of small launches (CQT2010v2 / VQT, 96 bins: seven
decimations, the grouped octave contractions, their pre-passes); back-to-back on a stream they are
separated by a few microseconds each.  ``Graphed`` captures one forward into a HIP graph (through
``torch.cuda.CUDAGraph``: the library launches on torch's current stream, which is the capturing
stream, and its scratch tensors come from the graph's private pool) and replays it per call::

    fast = nnaudio_amd.graph.Graphed(module, example_batch)
    y = fast(x)            # x: same shape/dtype/device as example_batch; y is a static buffer

The graph freezes everything that is decided on the host: shapes, the basis buffers' addresses and
their derived tensors (supports, split planes, scales).  Re-capture after ``load_state_dict`` /
``.to()`` / in-place edits of the buffers; inference only (no autograd graph is recorded)."""
import torch


class Graphed:
    def __init__(self, module, example, warmup=3, **forward_kwargs):
        if not example.is_cuda:
            raise RuntimeError("Graphed needs a GPU tensor (there is no CPU path)")
        self.module = module
        self.kwargs = dict(forward_kwargs)
        self.static_in = example.detach().clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(max(1, warmup)):  # one-time work (kernel attributes, caches) out of the graph
                module(self.static_in, **self.kwargs)
        torch.cuda.current_stream(example.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = module(self.static_in, **self.kwargs)

    def __call__(self, x):
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype or x.device != self.static_in.device:
            raise RuntimeError("Graphed was captured for %s %s on %s, got %s %s on %s"
                               % (tuple(self.static_in.shape), self.static_in.dtype, self.static_in.device,
                                  tuple(x.shape), x.dtype, x.device))
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
