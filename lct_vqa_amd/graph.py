"""HIP-graph capture of a launch-bound step.

The PC-DARTS path issues ~9 000 small kernels per forward+backward; at small batch the Python / autograd / ctypes work per
launch (~10 us) exceeds the kernels' own time.  Every entry point of the C ABI only launches on the
stream it is given (no allocation, no synchronisation), so a whole step can be captured once and replayed:

    step = lambda: torch.autograd.grad(net(x).sum(), params)      # static shapes, inputs read from fixed tensors
    g = Graphed(step)                                             # warms up on a side stream, then captures
    g.replay(); grads = g.outputs                                 # static output tensors, overwritten by every replay

Limits (those of torch.cuda.graph): shapes and control flow are frozen, inputs must be copied into the captured tensors,
host-side values are baked in -- in particular the (seed, offset) of dropout masks drawn inside the step, so a captured
step with live dropout repeats its masks; capture steps without dropout (the encoder) or draw the masks outside.
"""
import torch


class Graphed:

    def __init__(self, fn, warmup=2):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs
