"""RCCL process group next to hipGraph capture (world size 1: the boxes have one GPU; the N>1 arithmetic is covered by the
gloo world-2 test in test_host_logic.py): init nccl, capture a GraphedStep while the watchdog thread is alive,
interleave replays with all-reduces."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_group_and_graph_capture_coexist(cuda_device):
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'dist_smoke.py')], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert 'ok ' in r.stdout and 'True' in r.stdout
