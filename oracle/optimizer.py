"""ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's own `SimpleAdam` arithmetic (dnnlib/tflib/optimizer.py:318-332: "behaves
identically" to tf.train.AdamOptimizer under tflib.Optimizer), the non-finite-gradient skip
(:237-239) and the EMA of Network.setup_as_moving_average_of (dnnlib/tflib/network.py:341-351),
restated in NumPy float32 on flat arrays.
PINNED (round 4): tests/golden/ref_train_golden.npz holds the weights the reference's own Optimizer.apply_updates + SimpleAdam produce step
by step (one / two devices, a non-finite step, the accumulation branch) and the results of setup_as_moving_average_of, executed under
tests/golden/np_tf.py; tests/test_ref_train_golden.py requires this file to reproduce them.
"""
import numpy as np


class SimpleAdam:
    def __init__(self, n, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = np.float32(learning_rate), np.float32(beta1), np.float32(beta2), np.float32(epsilon)
        self.b1pow = np.float32(1)
        self.b2pow = np.float32(1)
        self.m = np.zeros(n, np.float32)
        self.v = np.zeros(n, np.float32)

    def apply(self, w, g, lr=None):
        """In place on w.  Skips (optimizer.py:237) when any gradient is non-finite."""
        if not np.all(np.isfinite(g)):
            return False
        lr = self.lr if lr is None else np.float32(lr)
        self.b1pow = np.float32(self.b1pow * self.b1)
        self.b2pow = np.float32(self.b2pow * self.b2)
        lr_new = np.float32(lr * np.sqrt(np.float32(1) - self.b2pow) / (np.float32(1) - self.b1pow))
        self.m = (self.b1 * self.m + (np.float32(1) - self.b1) * g).astype(np.float32)
        self.v = (self.b2 * self.v + (np.float32(1) - self.b2) * np.square(g)).astype(np.float32)
        w -= (lr_new * self.m / (np.sqrt(self.v) + self.eps)).astype(np.float32)
        return True


def ema(dst, src, beta):
    """lerp(src, dst, beta) = src + (dst - src) * beta  (tfutil.py:62-65, network.py:348)."""
    return (src + (dst - src) * np.float32(beta)).astype(np.float32)
