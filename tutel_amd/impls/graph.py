"""HIP-graph capture of the MoE forward (inference).

With capacity_factor > 0 the forward path never synchronises with the host (the dropless mode's
one `int(capacity)` is the only sync the API implies), and every HIP kernel is enqueued on the
caller's current stream through the C ABI -- so the whole layer, routing to combine, can be captured
once and replayed with zero per-step host work:

    graphed = GraphedForward(layer, example_input)     # captures on a side stream
    y = graphed(x)                                     # copies x into the static input, replays

The static input/output buffers are owned by the wrapper; `y` is valid until the next call."""
import torch


class GraphedForward:
    def __init__(self, layer, example, warmup=3, **forward_kwargs):
        assert example.is_cuda, "HIP-graph capture needs a device tensor"
        if forward_kwargs.get("capacity_factor", getattr(layer.gates[0], "capacity_factor", 1.0)) <= 0:
            raise ValueError("dropless routing (capacity_factor <= 0) reads the capacity back to the host and cannot be captured")
        self.layer, self.kwargs = layer, forward_kwargs
        self._ep = getattr(layer, "world_size", 1) > 1
        self.static_in = example.clone()
        self.stream = torch.cuda.Stream(device=example.device)
        self.stream.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(warmup):
                layer(self.static_in, **forward_kwargs)
            torch.cuda.synchronize()
            # W > 1: only the IPC transport (plain kernels + events, epochs counted on the device) replays safely.  Captured RCCL
            # collectives replay ~200-400 times and then never complete (RCCL 2.26.6 in torch 2.10, profiles/r03_ep_streams.txt):
            # refuse instead of handing out a graph that hangs (TUTEL_AMD_GRAPH_RCCL=1 overrides, for probing)
            if getattr(layer, "world_size", 1) > 1:
                import os
                from . import ep_native
                comm = ep_native.communicator(layer.group, example.device) if ep_native.group_ok(layer.group) else None
                if (comm is None or not comm.ipc) and os.environ.get("TUTEL_AMD_GRAPH_RCCL", "0") != "1":
                    raise RuntimeError("GraphedForward: an expert-parallel layer can only be captured with the IPC transport "
                                       "(peer stores; TUTEL_AMD_EP_TRANSPORT=auto|ipc): replaying captured RCCL collectives hangs")
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, stream=self.stream):
                self.static_out = layer(self.static_in, **forward_kwargs)
        # the captured kernels hold raw pointers into the layer's cached pipeline workspaces (impls/ep_native.py keeps an LRU of
        # them): pin the ones alive now for as long as this graph lives, so that eviction cannot free memory a replay writes
        self._pinned = list(layer.__dict__.get("_ep_workspaces", {}).values())
        self.l_aux = getattr(self.static_out, "l_aux", None)
        torch.cuda.current_stream().wait_stream(self.stream)

    def __call__(self, x):
        # a replay makes no library call, so nothing would ever report an exchange of the IPC transport that gave up (a peer that
        # never arrived, rows behind their flag): ask before every replay -- one read of a pinned host word.  The
        # replay that gave up has poisoned its own output with NaN on the device (csrc/ep.hip::ep_poison_kernel).
        if self._ep:
            from . import ep_native
            ep_native.ipc_status(self.layer.group, self.static_in.device)
        if x.data_ptr() != self.static_in.data_ptr():
            self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out
