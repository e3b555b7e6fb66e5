"""Parity at the FULL sizes of the bench configuration (CelebA 128x128, config-e, minibatch_gpu 6), where the fp64 oracle
would take minutes: size-independent properties instead of element-wise comparison with the oracle.

  * adjointness  <conv(x; w), dy> == <x, dgrad(dy; w)> == <w, wgrad(x, dy)>   (the three kernels are independent
    implementations of one bilinear form -- forward, data-gradient and weight-gradient kernels, sliced tails, thin and
    dense special cases all have to agree on it);
  * linearity in each argument;
  * fused == composite (a different kernel path through the same library);
  * upfirdn: <upfirdn(x), y> == <x, upfirdn_grad(y)>;
  * 1-NN against torch.cdist on a slice of the queries; idempotence of the running minimum.
Inner products are accumulated in fp64; tolerance 2e-4 relative to the magnitude sum |a||b| (fp32 kernels, K up to 4608)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dot(a, b):
    return float((a.detach().double() * b.detach().double()).sum())


def _close(u, v, scale, tol=2e-4):
    assert abs(u - v) <= tol * scale, (u, v, scale)


# (name, N, Cin, H, Cout, K, stride, up, pad, out, scales) -- the layers of config-e @128 (tools/conv_bench.py)
LAYERS = [
    ('G 128 Conv1', 6, 128, 128, 128, 3, 1, 1, 1, 128, True),
    ('G 128 Conv0_up', 6, 256, 64, 128, 3, 1, 2, 2, 129, True),
    ('G 32 Conv1 (4 calls)', 24, 512, 32, 512, 3, 1, 1, 1, 32, True),
    ('G 4x4 Conv', 24, 512, 4, 512, 3, 1, 1, 1, 4, True),
    ('G 128 ToRGB', 24, 128, 128, 3, 1, 1, 1, 0, 128, 'in'),
    ('D 128 Conv0', 12, 128, 128, 128, 3, 1, 1, 1, 128, False),
    ('D 128 Conv1_down', 12, 128, 131, 256, 3, 2, 1, 0, 65, False),
    ('D 128 Skip', 12, 128, 129, 256, 1, 2, 1, 0, 65, False),
    ('D 128 FromRGB', 12, 3, 128, 128, 1, 1, 1, 0, 128, False),
    ('VGG conv1_1', 18, 3, 128, 64, 3, 1, 1, 1, 128, False),
    ('VGG conv3_2', 18, 256, 32, 256, 3, 1, 1, 1, 32, False),
    ('mapping dense', 24, 512, 1, 512, 1, 1, 1, 0, 1, False),
    ('D dense 8192', 12, 8192, 1, 512, 1, 1, 1, 0, 1, False),
]


@pytest.mark.parametrize('layer', LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_adjointness_and_linearity_full_size(layer, cuda_device):
    from inclusivegan_amd import hip_ops
    name, N, Cin, H, Cout, K, stride, up, pad, out, scales = layer
    g = torch.Generator(device='cpu').manual_seed(len(name) * 977 + N)
    dev = cuda_device
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    x = cl(torch.randn(N, Cin, H, H, generator=g)); x2 = cl(torch.randn(N, Cin, H, H, generator=g))
    w = (torch.randn(K, K, Cin, Cout, generator=g) / (K * K * Cin) ** 0.5).to(dev)
    dy = cl(torch.randn(N, Cout, out, out, generator=g))
    s = (torch.rand(N, Cin, generator=g) + 0.5).to(dev) if scales else None
    d = (torch.rand(N, Cout, generator=g) + 0.5).to(dev) if scales is True else None
    geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad, 0.61)
    y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d)
    assert tuple(y.shape) == (N, Cout, out, out) and bool(torch.isfinite(y).all())
    # the adjoint kernels of the same bilinear form  y = alpha * d * conv(x * s, w)
    dyd = dy * d[:, :, None, None] if d is not None else dy
    dxs = hip_ops.conv2d_raw(dyd, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True)
    dx = dxs * s[:, :, None, None] if s is not None else dxs
    dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d)
    lhs = _dot(y, dy)
    scale = float(y.double().norm() * dy.double().norm())
    _close(lhs, _dot(x, dx), scale)
    _close(lhs, _dot(w, dw), scale)
    # linearity in x (same weights, scales): conv(2x - 3x2) = 2 conv(x) - 3 conv(x2)
    y2 = hip_ops.conv2d_raw(x2, w, geom, (out, out), Cout, in_scale=s, out_scale=d)
    y3 = hip_ops.conv2d_raw(2.0 * x - 3.0 * x2, w, geom, (out, out), Cout, in_scale=s, out_scale=d)
    err = float((y3 - (2.0 * y - 3.0 * y2)).abs().max() / (y3.abs().max() + 1e-30))
    assert err < 5e-5, err
    # fused scales == scales applied outside the kernel (a different path through the same kernels)
    if s is not None:
        yc = hip_ops.conv2d_raw(x * s[:, :, None, None], w, geom, (out, out), Cout)
        if d is not None:
            yc = yc * d[:, :, None, None]
        assert float((y - yc).abs().max() / (yc.abs().max() + 1e-30)) < 5e-5


@pytest.mark.parametrize('layer', LAYERS, ids=[l[0] for l in LAYERS])
def test_conv_family_against_fp64_oracle_samples_full_size(layer, cuda_device):
    """An INDEPENDENT check at the bench sizes: 256 randomly chosen elements of the forward output, of the data gradient and
    of the weight gradient of every layer shape of config-e @128, each computed in fp64 by direct gather on the CPU
    (oracle/conv_sample.py: the TensorFlow definitions, no tiling) and compared with the HIP kernels' values.
    Tolerance 3e-5 of the largest sampled magnitude (exact-fp32 MFMA accumulation over K up to 8192)."""
    from inclusivegan_amd import hip_ops
    from oracle import conv_sample as CS
    name, N, Cin, H, Cout, K, stride, up, pad, out, scales = layer
    g = torch.Generator(device='cpu').manual_seed(len(name) * 131 + N)
    dev = cuda_device
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    x_c = torch.randn(N, Cin, H, H, generator=g)
    w_c = torch.randn(K, K, Cin, Cout, generator=g) / (K * K * Cin) ** 0.5
    dy_c = torch.randn(N, Cout, out, out, generator=g)
    s_c = (torch.rand(N, Cin, generator=g) + 0.5) if scales else None
    d_c = (torch.rand(N, Cout, generator=g) + 0.5) if scales is True else None
    alpha = 0.83
    x, w, dy = cl(x_c), w_c.to(dev), cl(dy_c)
    s = s_c.to(dev) if s_c is not None else None
    d = d_c.to(dev) if d_c is not None else None
    geom = hip_ops.ConvGeom(K, K, stride, up, pad, pad, alpha)
    y = hip_ops.conv2d_raw(x, w, geom, (out, out), Cout, in_scale=s, out_scale=d)
    dyd = dy * d[:, :, None, None] if d is not None else dy
    dxs = hip_ops.conv2d_raw(dyd, w, hip_ops.dgrad_geom(geom), (H, H), Cin, w_transposed=True)
    dx = dxs * s[:, :, None, None] if s is not None else dxs
    dw = hip_ops.conv2d_wgrad_raw(x, dy, geom, in_scale=s, out_scale=d)
    rng = np.random.RandomState(len(name))
    m = 256
    pick = lambda *dims: np.stack([rng.randint(n, size=m) for n in dims], 1)
    xn, wn, dyn = x_c.numpy(), w_c.numpy(), dy_c.numpy()
    sn = s_c.numpy() if s_c is not None else None
    dn = d_c.numpy() if d_c is not None else None

    def check(got, want, what):
        got = got.astype(np.float64)
        err = np.abs(got - want).max() / (np.abs(want).max() + 1e-30)
        assert err < 3e-5, (name, what, err)

    i = pick(N, Cout, out, out)
    check(y.cpu().numpy()[tuple(i.T)], CS.forward_samples(xn, wn, i, stride, up, pad, sn, dn, alpha), 'forward')
    i = pick(N, Cin, H, H)
    check(dx.cpu().numpy()[tuple(i.T)], CS.dgrad_samples(dyn, wn, i, (H, H), stride, up, pad, sn, dn, alpha), 'data gradient')
    i = pick(K, K, Cin, Cout)
    iw = i[:64]     # a weight-gradient element sums N*OH*OW products: 64 of them keep the CPU side in seconds
    check(dw.cpu().numpy()[tuple(iw.T)], CS.wgrad_samples(xn, dyn, iw, stride, up, pad, sn, dn, alpha), 'weight gradient')


def test_torgb_and_dense_against_torch_einsum_full_size(cuda_device):
    """The thin-channel and small-batch dense kernels against torch's own matmul on the device (fp32, same inputs)."""
    from inclusivegan_amd import hip_ops
    dev = cuda_device
    g = torch.Generator(device='cpu').manual_seed(3)
    x = torch.randn(24, 128, 128, 128, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(1, 1, 128, 3, generator=g) / 128 ** 0.5).to(dev)
    s = (torch.rand(24, 128, generator=g) + 0.5).to(dev)
    y = hip_ops.conv2d_raw(x, w, hip_ops.ConvGeom(1, 1, 1, 1, 0, 0), (128, 128), 3, in_scale=s)
    ref = torch.einsum('nchw,co->nohw', x.double() * s.double()[:, :, None, None], w.double()[0, 0])
    assert float((y.double() - ref).abs().max() / ref.abs().max()) < 1e-5
    z = torch.randn(24, 512, generator=g).to(dev); a = (torch.randn(512, 512, generator=g) / 512 ** 0.5).to(dev)
    assert float((hip_ops.matmul(z, a, alpha=0.5).double() - 0.5 * z.double() @ a.double()).abs().max()) < 1e-5


@pytest.mark.parametrize('case', [(6, 129, 128, 1, 1, 1, 1), (12, 128, 128, 1, 1, 2, 2), (6, 64, 3, 2, 1, 2, 1), (6, 128, 3, 1, 2, 1, 2)])
def test_upfirdn_adjoint_full_size(case, cuda_device):
    """<upfirdn(x), y> == <x, d upfirdn(x) / dx applied to y> at the 128x128 call sites (FIR after the up-conv, before the
    strided convs, RGB up / down sampling)."""
    from inclusivegan_amd.dnnlib.tflib.ops.upfirdn_2d import _setup_kernel, _simple_upfirdn_2d
    B, H, C, up, down, p0, p1 = case
    dev = cuda_device
    g = torch.Generator(device='cpu').manual_seed(H + C)
    x = torch.randn(B, C, H, H, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    k = _setup_kernel([1, 3, 3, 1]) * (up ** 2)
    y = _simple_upfirdn_2d(x, k, up=up, down=down, pad0=p0, pad1=p1, data_format='NCHW')
    dy = torch.randn(y.shape, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    (dx,) = torch.autograd.grad(y, x, dy)
    _close(_dot(y, dy), _dot(x.detach(), dx), float(y.double().norm() * dy.double().norm()), 1e-5)


def test_nn1_against_cdist_and_idempotence(cuda_device):
    """CelebA-sized rows (49 152 dims): the streaming 1-NN equals torch.cdist's argmin on a slice of the queries, and
    folding the same candidates twice leaves the running minimum unchanged."""
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dci_code.dci import unpack_best
    dev = cuda_device
    g = torch.Generator(device='cpu').manual_seed(11)
    dim, nq, nc = 49152, 1024, 512
    q = torch.rand(nq, dim, generator=g).to(dev) * 2 - 1
    c = torch.rand(nc, dim, generator=g).to(dev) * 2 - 1
    qn = hip_ops.row_sqnorm_raw(q); cn = hip_ops.row_sqnorm_raw(c)
    bd, bi = hip_ops.nn1_state(nq, dev)
    hip_ops.nn1_update_raw(q, qn, c, cn, bd, bi, 0)
    once = (bd.clone(), bi.clone())
    hip_ops.nn1_update_raw(q, qn, c, cn, bd, bi, 0)
    assert torch.equal(bd, once[0]) and torch.equal(bi, once[1])
    idx, dist = unpack_best(bd, bi)
    ref = torch.cdist(q[:128].double(), c.double())
    rd, ri = ref.min(dim=1)
    assert torch.equal(idx[:128].cpu(), ri.cpu())
    assert float((dist[:128].cpu() - rd.cpu()).abs().max() / rd.max()) < 1e-9      # winners are measured in fp64


def test_nn1_near_ties_decided_in_fp64(cuda_device):
    """Planted near-ties at dim 49 152 (the reference decides on fp64 distances, dci_code/src/util.c:62-69): per query
    eight candidates lie at distances 0.1 * (1 + k * 2^-12) -- gaps of 2.4e-5 relative in d^2 = 5e-7 absolute, four orders
    of magnitude below the cancellation error of |q|^2 + |c|^2 - 2 q.c in fp32 (~1e-6 * 3e4) -- scattered over the
    candidate batches among far random rows.  The arg-min and its distance must equal the fp64 brute-force search of
    oracle/nn.py exactly, whatever the batch order."""
    from inclusivegan_amd import hip_ops
    from inclusivegan_amd.dci_code.dci import unpack_best
    from oracle import nn as ONN
    dev = cuda_device
    rng = np.random.RandomState(5)
    dim, nq, nc, nb = 49152, 24, 96, 4
    q = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(nb * nc, dim)).astype(np.float32)
    want = np.empty(nq, dtype=np.int64)
    for i in range(nq):
        slots = rng.choice(nb * nc, size=8, replace=False)
        order = rng.permutation(8)                      # which planted rank sits in which slot
        for k, slot in zip(order, slots):
            row = q[i].copy()
            j = rng.randint(dim)
            row[j] = np.float32(row[j] + 0.1 * (1.0 + k * 2.0 ** -12) * (1 if row[j] < 0 else -1))
            c[slot] = row
    oidx, odist = ONN.nearest_neighbour(c, q)
    qd = torch.from_numpy(q).to(dev); cd = torch.from_numpy(c).to(dev)
    qn = hip_ops.row_sqnorm_raw(qd)
    for batches in (range(nb), reversed(range(nb))):
        bd, bi = hip_ops.nn1_state(nq, dev)
        for b in batches:
            cb = cd[b * nc:(b + 1) * nc]
            hip_ops.nn1_update_raw(qd, qn, cb, hip_ops.row_sqnorm_raw(cb), bd, bi, b * nc)
        idx, dist = unpack_best(bd, bi)
        assert np.array_equal(idx.cpu().numpy(), oidx.astype(np.int64))
        assert np.abs(dist.cpu().numpy() - odist).max() <= 1e-12 * odist.max()
    # a non-finite candidate can never win, and does not poison the state
    cd2 = cd[:nc].clone(); cd2[3, 7] = float('nan'); cd2[5, 9] = float('inf')
    bd, bi = hip_ops.nn1_state(nq, dev)
    hip_ops.nn1_update_raw(qd, qn, cd2, hip_ops.row_sqnorm_raw(cd2), bd, bi, 0)
    assert bool(torch.isfinite(bd).all()) and not bool(((bi == 3) | (bi == 5)).any())


def test_knn_short_list_is_widened_when_many_candidates_nearly_coincide(cuda_device):
    """k > 1 (the exclusive assignment): 40 candidates per query within 2^-12 relative of one another at dim 49 152 -- far more
    than the k + 8 short list of the fp32 screening, and indistinguishable to it.  query_device_k must notice that the list
    cannot be proven sufficient, widen it, and return the fp64 brute-force k nearest (distance, then index)."""
    from inclusivegan_amd.dci_code.dci import DCI
    dev = cuda_device
    rng = np.random.RandomState(9)
    dim, nq, n, k = 49152, 6, 400, 5
    q = rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32)
    c = rng.uniform(-1, 1, size=(n, dim)).astype(np.float32)
    for i in range(nq):
        slots = rng.choice(n // nq, size=40, replace=False) * nq + i         # disjoint slot sets per query
        for r, slot in enumerate(rng.permutation(slots)):
            row = q[i].copy()
            j = rng.randint(dim)
            row[j] = np.float32(row[j] + 0.1 * (1.0 + r * 2.0 ** -12) * (1 if row[j] < 0 else -1))
            c[slot] = row
    d = np.sqrt(((q[:, None, :].astype(np.float64) - c[None, :, :].astype(np.float64)) ** 2).sum(-1))
    want = np.lexsort((np.broadcast_to(np.arange(n), d.shape), d), axis=1)[:, :k]
    db = DCI(dim, device=dev)
    db.add(torch.from_numpy(c).to(dev))
    idx, dist = db.query_device_k(torch.from_numpy(q).to(dev), k)
    assert max(db.last_margins) > 8, 'the planted near-ties did not force a wider short list: the test no longer probes the check'
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.abs(dist.cpu().numpy() - np.take_along_axis(d, want, 1)).max() <= 1e-12 * d.max()


def test_knn_collapsed_candidates_terminate_inside_the_memory_budget(cuda_device):
    """a collapsed candidate set (every candidate within rounding of every other -- a mode-collapsed generator) makes the
    sufficiency check fail however wide the short list is.  The search must still terminate, must not gather blocks beyond
    `rerank_bytes`, must grow from margin 0, and must return the fp64 brute-force neighbours: past `max_keep` it selects by threshold
    against the exact k-th distance found so far instead of keeping every candidate."""
    from inclusivegan_amd.dci_code.dci import DCI
    dev = cuda_device
    rng = np.random.RandomState(11)
    dim, nq, n, k = 3072, 5, 700, 4
    base = rng.uniform(-1, 1, size=(1, dim)).astype(np.float32)
    c = np.repeat(base, n, axis=0)
    for i in range(n):                                            # differences of a few ulps: far inside the screening error
        c[i, rng.randint(dim)] += np.float32(2.0 ** -20 * (1 + i % 7))
    q = (base + rng.uniform(-1, 1, size=(nq, dim)).astype(np.float32) * np.float32(0.05)).astype(np.float32)
    d = np.sqrt(((q[:, None, :].astype(np.float64) - c[None, :, :].astype(np.float64)) ** 2).sum(-1))
    want = np.lexsort((np.broadcast_to(np.arange(n), d.shape), d), axis=1)[:, :k]
    db = DCI(dim, device=dev)
    db.rerank_bytes = 64 * dim * 8                                # 64 candidate rows per fp64 block
    db.add(torch.from_numpy(c).to(dev))
    torch.cuda.reset_peak_memory_stats(dev)
    before = torch.cuda.memory_allocated(dev)
    idx, dist = db.query_device_k(torch.from_numpy(q).to(dev), k, margin=0, max_keep=64)
    assert db.last_margins[:3] == [0, 8, 32] and db.last_threshold_queries > 0
    assert torch.cuda.max_memory_allocated(dev) - before < 64 << 20
    assert np.array_equal(idx.cpu().numpy(), want)
    assert np.abs(dist.cpu().numpy() - np.take_along_axis(d, want, 1)).max() <= 1e-12 * d.max()


@pytest.mark.parametrize('case', [('G 128 Conv1', 6, 128, 128, 128, True), ('G 32 Conv1 (4 calls)', 24, 512, 32, 512, True), ('G 8 Conv1', 24, 512, 8, 512, False)],
                         ids=lambda c: c[0])
def test_fused_synthesis_layer_against_fp64_samples_full_size(case, cuda_device):
    """The one-kernel synthesis layer (modulated 3x3 conv + noise + bias + lrelu, hip_ops.ModConvBanFn) at the bench sizes against
    fp64 values computed from the definition (networks_stylegan2.py:99-127,349-357) for sampled elements: the forward output, and
    -- for a few (sample, channel) pairs -- the demodulation gradient dd[n,c] = sum_pixels dx[n,c,p] * z[n,c,p], which the backward
    kernel obtains by inverting the activation instead of keeping z."""
    from inclusivegan_amd import hip_ops
    from oracle import conv_sample as CS
    name, N, C, H, Cout, per_sample = case
    g = torch.Generator(device='cpu').manual_seed(len(name) * 71 + N)
    dev = cuda_device
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    x_c = torch.randn(N, C, H, H, generator=g)
    w_c = torch.randn(3, 3, C, Cout, generator=g) / (9 * C) ** 0.5
    s_c = torch.rand(N, C, generator=g) + 0.5
    d_c = torch.rand(N, Cout, generator=g) + 0.5
    b_c = torch.randn(Cout, generator=g) * 0.2
    nz_c = torch.randn(N if per_sample else 1, 1, H, H, generator=g)
    dy_c = torch.randn(N, Cout, H, H, generator=g)
    alpha, strength, gain, slope = 0.83, 0.35, float(np.sqrt(2)), 0.2
    x = cl(x_c).requires_grad_(True)
    d = d_c.to(dev).requires_grad_(True)
    st = torch.tensor(strength, device=dev)
    geom = hip_ops.ConvGeom(3, 3, 1, 1, 1, 1, alpha)
    y = hip_ops.ModConvBanFn.apply(x, w_c.to(dev), s_c.to(dev), d, b_c.to(dev), nz_c.to(dev), st, geom, (H, H), 3, slope, gain)
    (gd,) = torch.autograd.grad(y, [d], cl(dy_c))
    rng = np.random.RandomState(len(name))
    m = 256
    i = np.stack([rng.randint(n, size=m) for n in (N, Cout, H, H)], 1)
    xn, wn, sn, dn = x_c.numpy(), w_c.numpy(), s_c.numpy(), d_c.numpy()
    conv = CS.forward_samples(xn, wn, i, 1, 1, 1, sn, dn, alpha)                                   # alpha * d * conv(x * s, w), fp64
    nz = nz_c.numpy().astype(np.float64)
    pre = conv + nz[i[:, 0] if per_sample else 0, 0, i[:, 2], i[:, 3]] * strength + b_c.numpy().astype(np.float64)[i[:, 1]]
    want = np.where(pre > 0, pre, pre * slope) * gain
    got = y.detach().cpu().numpy()[tuple(i.T)].astype(np.float64)
    assert np.abs(got - want).max() / np.abs(want).max() < 3e-5
    # dd for three (n, c) pairs: every pixel of that sample and channel in fp64
    for n, c in ((0, 1), (N - 1, Cout - 1), (N // 2, 7)):
        hw = np.stack(np.meshgrid(np.arange(H), np.arange(H), indexing='ij'), -1).reshape(-1, 2)
        idx = np.concatenate([np.full((H * H, 1), n), np.full((H * H, 1), c), hw], 1)
        z = CS.forward_samples(xn, wn, idx, 1, 1, 1, sn, None, alpha)                               # un-demodulated output
        pre = z * float(dn[n, c]) + nz[n if per_sample else 0, 0].reshape(-1) * strength + float(b_c[c])
        dpre = dy_c.numpy()[n, c].reshape(-1).astype(np.float64) * np.where(pre > 0, 1.0, slope) * gain
        want_dd = float((dpre * z).sum())
        scale = float(np.abs(dpre * z).sum())
        assert abs(float(gd[n, c]) - want_dd) < 2e-5 * scale, (name, n, c, float(gd[n, c]), want_dd)
