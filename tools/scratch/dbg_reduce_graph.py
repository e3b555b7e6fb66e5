"""torch reductions inside a captured hipGraph: replay with new data vs eager."""
import torch
dev = torch.device('cuda', 0)
CL = torch.channels_last
cases = [
    ('sum(2,3) [3,64,32,32] CL', lambda: torch.randn(3, 64, 32, 32, device=dev).contiguous(memory_format=CL), lambda x: x.sum(dim=(2, 3))),
    ('sum(2,3) [3,64,32,32] NCHW', lambda: torch.randn(3, 64, 32, 32, device=dev), lambda x: x.sum(dim=(2, 3))),
    ('sum(2,3) [12,512,8,8] CL', lambda: torch.randn(12, 512, 8, 8, device=dev).contiguous(memory_format=CL), lambda x: x.sum(dim=(2, 3))),
    ('sum(2,3) [6,128,128,128] CL', lambda: torch.randn(6, 128, 128, 128, device=dev).contiguous(memory_format=CL), lambda x: x.sum(dim=(2, 3))),
    ('sum(1,2,3) [6,3,128,128]', lambda: torch.randn(6, 3, 128, 128, device=dev), lambda x: (x * x).sum(dim=[1, 2, 3])),
    ('sum() [3,3,128,128]', lambda: torch.randn(3, 3, 128, 128, device=dev), lambda x: torch.sum(x * 2.0)),
    ('sum() [3,3,32,32]', lambda: torch.randn(3, 3, 32, 32, device=dev), lambda x: torch.sum(x * 2.0)),
    ('mean() [6]', lambda: torch.randn(6, device=dev), lambda x: x.mean()),
    ('sum(0) [4096,512]', lambda: torch.randn(4096, 512, device=dev), lambda x: x.sum(dim=0)),
    ('mean(0) [24,512]', lambda: torch.randn(24, 512, device=dev), lambda x: x.mean(dim=0)),
    ('sum(2) mean(1) [3,8,512]', lambda: torch.randn(3, 8, 512, device=dev), lambda x: torch.sqrt(torch.mean(torch.sum(x * x, dim=2), dim=1))),
    ('isfinite.all [1M]', lambda: torch.randn(1 << 20, device=dev), lambda x: torch.isfinite(x).all().float()),
    ('norm [8M]', lambda: torch.randn(8 << 20, device=dev), lambda x: x.norm()),
]
for name, make, fn in cases:
    x = make()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = fn(x)
    res = []
    for it in range(4):
        x.copy_(make())
        g.replay()
        torch.cuda.synchronize()
        ref = fn(x)
        res.append(float((y - ref).abs().max() / (ref.abs().max() + 1e-30)))
    print('%-34s replay-vs-eager rel err per replay: %s' % (name, ' '.join('%.1e' % r for r in res)), flush=True)
