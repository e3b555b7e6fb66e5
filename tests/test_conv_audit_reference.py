"""CPU: the vectorised fp64 sample evaluators of tools/conv_audit.py (used on the device to audit every convolution call of a second-order step) against
oracle/conv_sample.py, the pinned single-element definitions (tests/test_oracle_ops.py holds those to torch's fp64 convolutions): forward, the
w_transposed (data-gradient) form of include/igan_hip.h, up-sampling and stride-2 geometries, modulation scales, weight gradient."""
import numpy as np
import pytest
import torch

from inclusivegan_amd.hip_ops import ConvGeom, dgrad_geom
from oracle import conv_sample as CS
from tools import conv_audit as CA


@pytest.mark.parametrize('stride,up,pad,H,OH', [(1, 1, 1, 9, 9), (2, 1, 0, 11, 5), (1, 2, 2, 5, 11)])
def test_sample_references_match_the_oracle_definitions(stride, up, pad, H, OH):
    rng = np.random.RandomState(3)
    N, Cin, Cout, K = 2, 8, 12, 3
    x = torch.from_numpy(rng.randn(N, Cin, H, H).astype(np.float32)).contiguous(memory_format=torch.channels_last)
    w = torch.from_numpy(rng.randn(K, K, Cin, Cout).astype(np.float32))
    dy = torch.from_numpy(rng.randn(N, Cout, OH, OH).astype(np.float32)).contiguous(memory_format=torch.channels_last)
    s = torch.from_numpy(rng.rand(N, Cin).astype(np.float32) + 0.5)
    d = torch.from_numpy(rng.rand(N, Cout).astype(np.float32) + 0.5)
    geom = ConvGeom(K, K, stride, up, pad, pad, 0.37)
    m = 64
    idx = np.stack([rng.randint(0, N, m), rng.randint(0, Cout, m), rng.randint(0, OH, m), rng.randint(0, OH, m)], 1)
    want = CS.forward_samples(x.numpy(), w.numpy(), idx, stride, up, pad, s=s.numpy(), d=d.numpy(), alpha=0.37)
    got, mag = CA.conv_reference(x, w, geom, tuple(torch.from_numpy(idx[:, i]) for i in range(4)), False, s, d)
    xs = (x * s[:, :, None, None]).double().numpy()        # the audit rounds x * s to fp32 once, as the kernels do
    want32 = CS.forward_samples((x * s[:, :, None, None]).numpy(), w.numpy(), idx, stride, up, pad, d=d.numpy(), alpha=0.37)
    np.testing.assert_allclose(got.numpy(), want32, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-5, atol=1e-6)
    assert bool((mag.numpy() >= np.abs(got.numpy()) - 1e-12).all())
    # data gradient = the same entry point with the mirrored geometry and the forward weight read transposed + flipped (include/igan_hip.h)
    gd = dgrad_geom(geom)
    idx = np.stack([rng.randint(0, N, m), rng.randint(0, Cin, m), rng.randint(0, H, m), rng.randint(0, H, m)], 1)
    want = CS.dgrad_samples((dy * d[:, :, None, None]).numpy(), w.numpy(), idx, (H, H), stride, up, pad, s=s.numpy(), alpha=0.37)
    got, _ = CA.conv_reference(dy, w, gd, tuple(torch.from_numpy(idx[:, i]) for i in range(4)), True, d, s)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)
    # weight gradient
    idx = np.stack([rng.randint(0, K, m), rng.randint(0, K, m), rng.randint(0, Cin, m), rng.randint(0, Cout, m)], 1)
    want = CS.wgrad_samples((x * s[:, :, None, None]).numpy(), (dy * d[:, :, None, None]).numpy(), idx, stride, up, pad, alpha=0.37)
    got, mag = CA.wgrad_reference(x, dy, geom, tuple(torch.from_numpy(idx[:, i]) for i in range(4)), s, d)
    np.testing.assert_allclose(got.numpy(), want, rtol=1e-12, atol=1e-12)
    assert bool((mag.numpy() >= np.abs(got.numpy()) - 1e-12).all())
