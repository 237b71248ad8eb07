"""Would the trunk forward gain from one stream per modality?  (GPU box)
  python scripts/modality_streams_probe.py
A chain of bottleneck blocks (conv1 1x1 -> BN+ReLU -> conv2 3x3 -> BN+ReLU -> conv3 1x1 -> BN + residual + ReLU; BatchNorm
statistics through the fixed-point totals) at the layer1 / layer2 / layer3 / layer4 shapes of the B = 64 step, run (a) as the
executor does -- every launch covers the 3 modalities, one stream -- and (b) as three independent single-modality chains on
three streams.  Same kernels, same bytes; (b) has 3x the launches at a third of the workgroups each."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ieee_amd import _lib as L, _ops

lib = L.require_gpu()
dt = torch.bfloat16
N = 64


class Unit:
    def __init__(self, G, H, W, Ci, Co, R, gen):
        self.G, self.H, self.W, self.Ci, self.Co, self.R = G, H, W, Ci, Co, R
        self.M = N * H * W
        w = (torch.randn(G, Co, Ci, R, R, generator=gen) * 0.05).cuda()
        self.wp = _ops.pack_conv_weight(w, dt, 0)
        self.rb = lib.ieee_conv2d_fwd_stats_rblocks(N, H, W)
        self.part = torch.zeros(G, 2, Co, self.rb, device="cuda")
        self.rep = 1 if (self.M + 127) // 128 <= 64 else 4
        self.tot = torch.zeros(self.rep, G, 2, Co, dtype=torch.int64, device="cuda")
        self.y = torch.empty(G, N, H, W, Co, device="cuda", dtype=dt)
        self.a = torch.empty_like(self.y)
        self.gam, self.bet = torch.ones(G, Co).cuda(), torch.zeros(G, Co).cuda()
        self.stats = torch.zeros(G, 4, Co, device="cuda")
        self.rm, self.rv = torch.zeros(G, Co, device="cuda"), torch.ones(G, Co, device="cuda")

    def run(self, x, res=None):
        G = self.G
        self.tot.zero_()
        L.check(lib.ieee_conv_next_bn_totals(L.ptr(self.tot), 2 * self.Co, self.rep, None))
        L.check(lib.ieee_conv2d_fwd(L.ptr(x), L.ptr(self.wp), L.ptr(self.y), L.IEEE_BF16, G, N, self.H, self.W, self.Ci, self.Co, self.R, self.R,
                                    1, self.R // 2, x[0].numel(), self.wp.stride(0), self.y[0].numel(), L.ptr(self.part), L.stream()))
        L.check(lib.ieee_bn2d_fwd_totals(L.ptr(self.y), L.ptr(res) if res is not None else None, L.ptr(self.a), L.IEEE_BF16, G, self.M, self.Co,
                                         self.M * self.Co, L.ptr(self.gam), L.ptr(self.bet), self.Co, L.ptr(self.rm), L.ptr(self.rv), self.Co,
                                         L.ptr(self.stats), L.ptr(self.tot), self.rep, 0.1, 1e-5, 1, None, None, L.stream()))
        return self.a


class Chain:
    def __init__(self, G, H, W, p, blocks, gen):
        self.x = torch.randn(G, N, H, W, 4 * p, generator=gen).cuda().to(dt)
        self.blocks = [(Unit(G, H, W, 4 * p, p, 1, gen), Unit(G, H, W, p, p, 3, gen), Unit(G, H, W, p, 4 * p, 1, gen)) for _ in range(blocks)]

    def run(self):
        x = self.x
        for c1, c2, c3 in self.blocks:
            x = c3.run(c2.run(c1.run(x)), res=x)
        return x


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


streams = [torch.cuda.Stream() for _ in range(3)]
for name, H, W, p, blocks in (("layer1", 64, 32, 64, 3), ("layer2", 32, 16, 128, 4), ("layer3", 16, 8, 256, 6), ("layer4", 16, 8, 512, 3)):
    gen = torch.Generator().manual_seed(3)
    grouped = Chain(3, H, W, p, blocks, gen)
    single = [Chain(1, H, W, p, blocks, gen) for _ in range(3)]

    def three_streams():
        main = torch.cuda.current_stream()
        fork = torch.cuda.Event()
        fork.record(main)
        for s, c in zip(streams, single):
            s.wait_event(fork)
            with torch.cuda.stream(s):
                c.run()
            e = torch.cuda.Event()
            e.record(s)
            main.wait_event(e)

    def one_stream_singles():
        for c in single:
            c.run()

    if os.environ.get("PROBE_GRAPH", "1") != "0":
        # the Python host loop needs 5-8 us per call: with 216 launches the three-stream form of layer3 is host-bound.  Replayed
        # from captured graphs neither form waits for the host.
        def captured(fn):
            fn(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            return g.replay
        g_g, g_3 = captured(grouped.run), captured(three_streams)
        print("%s as graphs: grouped %.0f us | three streams %.0f us" % (name, min(timed(g_g), timed(g_g)), min(timed(g_3), timed(g_3))), flush=True)
    t_g, t_3, t_1 = timed(grouped.run), timed(three_streams), timed(one_stream_singles)
    t_g, t_3, t_1 = min(t_g, timed(grouped.run)), min(t_3, timed(three_streams)), min(t_1, timed(one_stream_singles))   # best of two rounds
    print("%s (%d blocks, p = %d, %dx%d): grouped launches on one stream %.0f us | three single-modality chains on three streams %.0f us | "
          "the same three chains on one stream %.0f us" % (name, blocks, p, H, W, t_g, t_3, t_1), flush=True)
