#!/usr/bin/env python3
"""Does RCCL's all-reduce survive hipGraph capture on this stack?  One rank (the boxes have one GPU): the collective degenerates
to a copy but goes through the same capture path (communicator stream, work objects, watchdog).  Prints what
dnnlib.tflib.optimizer.collectives_capturable() would decide on a multi-rank group, then runs a captured GradientExchange-style
sequence (async all-reduce of bucket chunks issued from inside the captured region, waited at its end) for several replays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29541')
import torch  # noqa: E402


def main():
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    torch.distributed.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    t = torch.ones(1 << 20, device=dev)
    torch.distributed.all_reduce(t)
    torch.cuda.synchronize()
    ok = True
    try:
        g = torch.cuda.CUDAGraph()
        chunks = [t[i * (1 << 18):(i + 1) * (1 << 18)] for i in range(4)]
        side = torch.zeros(1 << 18, device=dev)
        with torch.cuda.graph(g, capture_error_mode='thread_local'):
            works = []
            for c in chunks:
                c.mul_(2.0)                                     # "backward" producing a chunk
                works.append(torch.distributed.all_reduce(c, async_op=True))
                side.add_(1.0)                                  # more compute while the collective is in flight
            for w in works:
                w.wait()
            t.add_(1.0)                                         # "optimizer" after the exchange
        for _ in range(5):
            g.replay()
        torch.cuda.synchronize()
        # each replay: x -> 2x + 1, five times from 1: 63
        ok = bool((t == 63.0).all()) and bool((side == 20.0).all())
        print('captured all-reduce: replays correct =', ok)
    except Exception as e:   # noqa: BLE001
        ok = False
        print('captured all-reduce failed:', type(e).__name__, str(e)[:300])
    x = torch.ones(8, device=dev)
    torch.distributed.all_reduce(x)                              # the communicator is still usable afterwards
    torch.cuda.synchronize()
    print('eager all-reduce after the capture: ok', float(x.sum()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
    print('RESULT', 'capturable' if ok else 'not capturable (exchange stays outside the graphs)')


if __name__ == '__main__':
    main()
