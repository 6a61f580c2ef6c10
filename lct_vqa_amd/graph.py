"""HIP-graph capture of a launch-bound step.

The PC-DARTS path issues thousands of small kernels per forward+backward; at small batch the Python / autograd / ctypes work
per launch (~10 us) exceeds the kernels' own time.  Every entry point of the C ABI only launches on
the stream it is given (no allocation, no synchronisation), so a whole step can be captured once and replayed:

    step = lambda: torch.autograd.grad(net(x).sum(), params)      # static shapes, inputs read from fixed tensors
    g = Graphed(step)                                             # warms up on a side stream, then captures ON THAT STREAM
    g.replay(); grads = g.outputs                                 # static output tensors, overwritten by every replay

Limits (those of torch.cuda.graph): shapes and control flow are frozen, inputs must be copied into the captured tensors,
host-side values are baked in -- in particular the (seed, offset) of dropout masks drawn inside the step, so a captured
step with live dropout repeats its masks; capture steps without dropout (the encoder) or draw the masks outside.

The one way to crash a capture that round 1 met (gpurun_out/graph_fault.log: segmentation fault inside
``hipStreamEndCapture``, preceded by torch's "AccumulateGrad node's stream does not match" warning): the step ran
``loss.backward()`` into ``.grad`` while an autograd graph of an EARLIER, eager iteration was still alive (its loss was
kept), so the parameters' AccumulateGrad nodes were the old ones, bound to the default stream.  During capture the engine
then runs the accumulation -- and its synchronisation with the capture stream -- on the legacy default stream, i.e. work
that does not belong to the capturing stream is enqueued mid-capture and ending the capture dereferences a graph the
runtime has already invalidated.  ``Graphed`` therefore (1) warms up and captures on the SAME side stream, so nodes made
during warm-up match the capture, and (2) watches the warm-up for that stream mismatch and refuses (``GraphCaptureError``)
BEFORE any capture starts.  Use ``torch.autograd.grad`` inside captured steps, or drop every reference to earlier graphs
(losses, outputs) and set ``.grad`` to None before constructing ``Graphed``.
"""
import warnings

import torch

_MISMATCH = "AccumulateGrad node's stream does not match"


class GraphCaptureError(RuntimeError):
    pass


class Graphed:

    def __init__(self, fn, warmup=2):
        if warmup < 1:
            raise ValueError('Graphed needs at least one warm-up run (it is also the capture-safety check)')
        self.stream = torch.cuda.Stream()
        self.stream.wait_stream(torch.cuda.current_stream())
        warn_always = torch.is_warn_always_enabled()
        try:
            torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(True)
        except AttributeError:                                      # pragma: no cover - older torch
            pass
        with warnings.catch_warnings(record=True) as seen:
            warnings.simplefilter('always')
            torch.set_warn_always(True)                             # the mismatch warning is a warn-once otherwise
            try:
                with torch.cuda.stream(self.stream):
                    for _ in range(warmup):
                        fn()
            finally:
                torch.set_warn_always(warn_always)
        torch.cuda.current_stream().wait_stream(self.stream)
        torch.cuda.synchronize()
        for w in seen:
            if _MISMATCH in str(w.message):
                raise GraphCaptureError(
                    'this step accumulates into .grad through AccumulateGrad nodes that belong to another stream (an autograd '
                    'graph from an earlier iteration is still alive): capturing it would enqueue work on the default stream '
                    'mid-capture and crash hipStreamEndCapture.  Use torch.autograd.grad inside the step, or delete the '
                    'earlier losses/outputs and set .grad = None before Graphed(...).')
        for w in seen:
            warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, stream=self.stream):
            self.outputs = fn()

    def replay(self):
        self.graph.replay()
        return self.outputs
