"""hipGraph replay of an inference forward (BASELINE configs[4]: "hipGraph-captured encoder+attention+MANO").

The C ABI is stream-ordered and allocation-free, and the host modules allocate only through torch's caching allocator, so a
whole eval forward (~1000 launches for the fp32 network, ~76 + decoder for the fp16 backbone) can be captured once and
replayed with one launch call; tools/infer_bench.py measures the difference.  `GraphedInference` packages the usual
static-buffer protocol:

    g = GraphedInference(model.eval(), example_img)         # warm-up on a side stream, then capture
    out = g(img)                                            # copies img into the static input, replays, returns the
                                                            # static outputs (overwritten by the next call)
"""
import torch


class GraphedInference:
    def __init__(self, fn, *example_inputs, warmup=2):
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError('GraphedInference needs GPU tensors (HIP graphs only)')
        self.fn = fn
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                      # lazy tables, weight packing, allocator pools
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.static_out = fn(*self.static_in)

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError('GraphedInference was captured for %s %s, got %s %s'
                                 % (tuple(dst.shape), dst.dtype, tuple(src.shape), src.dtype))
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        return self.static_out
