"""GPU: the IMLE nearest-neighbour assignment of one refresh (reference training/training_loop.py:357-406) against a NumPy fp64
brute force over the same candidate images: plain, with the random projection (:205-213,365,380) and with the exclusive
assignment (:382-396).  The generator is replaced by a fixed map latent -> image so that both sides see the same candidates."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class _Rec:
    """A `training_set_rec` stand-in: uint8 images served in data-set order, 2 * minibatch per pull."""

    def __init__(self, images):
        self.images = images
        self.shape = list(images.shape[1:])
        self.dynamic_range = [0, 255]
        self.cur = 0

    def get_minibatch_np(self, n):
        out = self.images[self.cur:self.cur + n]
        self.cur = (self.cur + n) % self.images.shape[0]
        return out, np.zeros((n, 0), np.float32)


class _G:
    """get_output_for(latents, labels, is_validation=True) -> [n, 3, R, R] images in [-1, 1]: tanh of a fixed linear map."""

    def __init__(self, res, device, seed):
        g = torch.Generator().manual_seed(seed)
        self.w = (torch.randn(16, 3 * res * res, generator=g) * 0.5).to(device)
        self.res = res

    def get_output_for(self, z, lab, is_validation=False):
        return torch.tanh(z @ self.w).reshape(z.shape[0], 3, self.res, self.res).contiguous(memory_format=torch.channels_last)


def _brute(reals, cands, k):
    d = np.sqrt(((reals[:, None, :].astype(np.float64) - cands[None, :, :].astype(np.float64)) ** 2).sum(-1))
    o = np.argsort(d, axis=1, kind='stable')[:, :k]
    return o, np.take_along_axis(d, o, 1)


@pytest.mark.parametrize('mode', ['plain', 'projected', 'exclusive'])
def test_refresh_assignment_matches_brute_force(cuda_device, mode):
    from inclusivegan_amd.training import training_loop as TL
    from inclusivegan_amd.training import misc
    res, data_size, factor, mb = 8, 24, 5, 3
    rng = np.random.RandomState(11)
    images = rng.randint(0, 256, size=(data_size, 3, res, res)).astype(np.uint8)
    rec = _Rec(images)
    G = _G(res, cuda_device, 5)
    lat = rng.randn(data_size * factor, 16).astype(np.float32)
    labels = np.zeros((data_size * factor, 0), np.float32)
    proj = None
    if mode == 'projected':
        proj = rng.normal(0.0, 1.0 / 12, size=(3 * res * res, 12))
    projector = None if proj is None else torch.from_numpy(proj.astype(np.float32)).to(cuda_device)
    k = 3 if mode == 'exclusive' else 0
    idx, dist = TL.imle_refresh(G, rec, lat, labels, data_size, mb, 32, [-1, 1], cuda_device, projector=projector, exclusive_k=k)
    assert rec.cur == 0                                             # one full pass over the data set (:374-403)
    # the same candidates, on the host
    with torch.no_grad():
        cands = G.get_output_for(torch.from_numpy(lat).to(cuda_device), None).contiguous().reshape(lat.shape[0], -1).cpu().numpy()
    reals = misc.adjust_dynamic_range(images.astype(np.float32), [0, 255], [-1, 1]).reshape(data_size, -1)
    if proj is not None:
        cands = cands.astype(np.float64) @ proj.astype(np.float32).astype(np.float64)
        reals = reals.astype(np.float64) @ proj.astype(np.float32).astype(np.float64)
    if mode == 'exclusive':
        oi, od = _brute(reals, cands, k)
        want_i, want_d = TL.exclusive_assignment(oi, od)          # pinned to the reference's statements in tests/test_imle_host.py
    else:
        oi, od = _brute(reals, cands, 1)
        want_i, want_d = oi[:, 0], od[:, 0]
    assert idx.shape == (data_size,) and dist.dtype == np.float64
    if proj is None:
        assert np.array_equal(idx, want_i)
        assert np.allclose(dist, want_d, rtol=1e-6, atol=0)
    else:
        # the projection itself runs in fp32 on the device (the reference multiplies in fp64): allow a different pick only where
        # the two best candidates are closer than that rounding
        full = np.sqrt(((reals[:, None, :] - cands[None, :, :]) ** 2).sum(-1))
        got_d = full[np.arange(data_size), idx]
        assert np.all(got_d <= want_d * (1 + 1e-5))
        assert np.allclose(dist, got_d, rtol=1e-4)
    if mode == 'exclusive':
        assert len(set(idx.tolist())) >= data_size - 2               # picks are (nearly) all distinct
