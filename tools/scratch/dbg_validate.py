import os, sys
sys.path.insert(0, os.getcwd())
import tests.test_gpu_loop_parity as T
from inclusivegan_amd.training import training_loop as TL
from inclusivegan_amd.dnnlib.tflib import tfutil
seen = []
hooks = dict(on_graphs=seen.append, on_iteration=lambda i: True)
if os.environ.get('TAP') == '1': hooks['random_source'] = tfutil.TapRandom()
TL.training_loop(hooks=hooks, **T.loop_kwargs(int(os.environ.get('FMAP', '8192')), 6, data_size=48))
print('GRAPHS', seen[0])
