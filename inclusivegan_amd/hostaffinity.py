"""Host-side placement of a rank: one process per GPU with a SMALL host thread pool (and, optionally, on the GPU's own NUMA node).

What was measured (round 4, MI355X node = 2 sockets x 64 cores x 2 threads, container with a 16-CPU cgroup quota;
`python bench.py --data-size 1152 --steps 60`, same box, same code; profiles/r04_host_threads.txt):

    default (PyTorch sizes its intra-op pool to the affinity mask: 128 threads)      240 ... 299 img/s from run to run
    the same with the process confined to ONE socket (either one; 128 CPUs)            96 ... 105
    confined to 4 / 8 / 16 CPUs of either socket                                     278
    OMP_NUM_THREADS = 1 / 4 / 16, no confinement                                     283.7 / 283.7 / 283.7
    one socket + OMP_NUM_THREADS = 4                                                 284.6

With the HIP runtime's graph packet capture off (inclusivegan_amd/__init__.py) a graph replay is the host thread handing kernel nodes to
the device, so anything that stalls that thread shows as device idle time.  The stall was PyTorch's own CPU thread pool: every small CPU
tensor op of the loop (the copies into the pinned staging buffers) opens a parallel region over 128 OpenMP threads which then spin, the
container's CPU quota runs out and the whole process is throttled.  The socket the thread runs on made no measurable difference.
So: `limit_host_threads()` caps the pool (the engine's host work is a few KB of NumPy per iteration), the training loop fills its staging
buffers through NumPy views (no torch CPU op left in the iteration), and `pin_to_device_node()` stays available but off by default.
"""
import os


def limit_host_threads(n=4):
    """Cap PyTorch's intra-op CPU pool at `n` threads unless the user set OMP_NUM_THREADS / IGAN_HOST_THREADS.  -> the pool size in effect."""
    import torch
    want = os.environ.get('IGAN_HOST_THREADS')
    if want is not None:
        n = max(1, int(want))
    elif os.environ.get('OMP_NUM_THREADS'):
        return torch.get_num_threads()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if not part:
            continue
        lo, _, hi = part.partition('-')
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def device_local_cpus(device_index):
    """CPU ids local to the GPU `device_index` of this process (sysfs local_cpulist of its PCI function; None if unknown)."""
    import torch
    p = torch.cuda.get_device_properties(device_index)
    bus = getattr(p, 'pci_bus_id', None)
    if bus is None:
        return None
    bdf = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), bus, getattr(p, 'pci_device_id', 0))
    try:
        with open('/sys/bus/pci/devices/%s/local_cpulist' % bdf) as f:
            cpus = _parse_cpulist(f.read())
    except OSError:
        return None
    return cpus or None


def pin_to_device_node(device_index):
    """IGAN_PIN_NUMA=1 only (no gain measured, see the module docstring): confine every thread of the process -- and, by inheritance, every
    later one -- to the cores local to that GPU.  Needs an initialised device (PCI address).  -> sorted CPU list, or None when nothing changed."""
    if os.environ.get('IGAN_PIN_NUMA', '0') != '1' or not hasattr(os, 'sched_setaffinity'):
        return None
    cpus = device_local_cpus(device_index)
    if not cpus:
        return None
    allowed = os.sched_getaffinity(0) & cpus        # never widen what the launcher allowed
    if not allowed:
        return None
    try:
        tids = [int(t) for t in os.listdir('/proc/self/task')]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            os.sched_setaffinity(tid, allowed)
        except OSError:
            pass        # a thread that ended meanwhile
    return sorted(allowed)
