"""HIP-graph capture of the eval-mode `Generator.forward` ("HIP-graph the step").

One frame of the inference driver is ~90 small launches around the two MLP kernels; at low resolution the host cannot
issue them as fast as the GPU retires them.  `GraphedForward` records the whole forward once into a hipGraph
(`torch.cuda.CUDAGraph`; our ctypes launches go to torch's current stream, which is the capturing stream, so they are
recorded like any other kernel) and replays it with one `hipGraphLaunch` per frame.  Inputs (pose, latent, background
colour) are copied into fixed device buffers before each replay; outputs are views of the graph's own memory pool and
are overwritten by the next replay (clone what must survive).

Restrictions: eval mode only (no per-ray jitter: its RNG offset would be frozen), fixed batch size / resolution /
iteration counter (`cos_anneal_ratio` is a launch argument and therefore baked in), and parameters must not be
re-packed between capture and replay (call `recapture()` after an optimiser step or `load_state_dict`).
"""
import torch


class GraphedForward:
    def __init__(self, generator, bs=1, it=0, return_raw=True, keys=None):
        if generator.training:
            raise ValueError("GraphedForward captures the eval-mode forward (generator.eval())")
        self.gen, self.bs, self.it, self.return_raw, self.keys = generator, bs, int(it), return_raw, keys
        dev = generator.it.device
        if dev.type != "cuda":
            raise ValueError("GraphedForward needs the generator on a GPU")
        self.b2w = torch.eye(4, device=dev).repeat(bs, 1, 1)
        self.b2w[:, 2, 3] = 0.0
        self.z = torch.zeros(bs, generator.z_dim, device=dev)
        self.bg = torch.zeros(bs, 3, device=dev)
        self.graph, self.out = None, None

    def _run(self):
        with torch.no_grad():
            blob = self.gen(bs=self.bs, it=self.it, data={"b2w": self.b2w, "z": self.z, "bg_color": self.bg},
                            return_raw=self.return_raw)["box"]
        out = blob["render_out"]
        return out if self.keys is None else {k: out[k] for k in self.keys}

    def recapture(self):
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # warm-up off the capture: packs the weights, sets kernel attributes
            for _ in range(2):
                self._run()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._run()
        return self

    def __call__(self, b2w, z, bg_color=None):
        if self.graph is None:
            self.recapture()
        self.b2w.copy_(b2w, non_blocking=True)
        self.z.copy_(z, non_blocking=True)
        if bg_color is not None:
            self.bg.copy_(bg_color, non_blocking=True)
        self.graph.replay()
        return self.out
