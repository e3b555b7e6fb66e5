"""CPU: checkpoint interchange.  Snapshots are pickles of (G, D, Gs) in the layout of the
reference's Network.__getstate__ (dnnlib/tflib/network.py:255-265): class path dnnlib.tflib.network.Network, state
{version 4, name, static_kwargs, components, build_module_src, build_func_name, variables [(local name, ndarray)]},
components pickled recursively with their own variables.  Parity unpinned at the byte level (the reference's class needs
TensorFlow to import); the layout is checked field by field against the cited lines."""
import os
import pickle
import pickletools

import numpy as np
import torch


def _nets():
    from inclusivegan_amd.dnnlib import tflib
    kw = dict(num_channels=3, resolution=16, label_size=0, fmap_base=128, device='cpu')
    G = tflib.Network('G', func_name='inclusivegan_amd.training.networks_stylegan2.G_main', architecture='skip', seed=1, init_mul=1.0, **kw)
    D = tflib.Network('D', func_name='inclusivegan_amd.training.networks_stylegan2.D_stylegan2_feature', architecture='resnet', seed=2, **kw)
    return G, D


def test_state_layout_matches_reference_getstate():
    G, D = _nets()
    st = G.state_v4(build_module_src='SRC')
    assert list(st) == ['version', 'name', 'static_kwargs', 'components', 'build_module_src', 'build_func_name', 'variables']
    assert st['version'] == 4 and st['name'] == 'G' and st['build_func_name'] == 'G_main' and st['build_module_src'] == 'SRC'
    assert type(st['static_kwargs']) is dict and st['static_kwargs']['fmap_base'] == 128 and 'func_name' not in st['static_kwargs']
    assert sorted(st['components']) == ['mapping', 'synthesis']                      # networks_stylegan2.py:187-190
    assert [n for n, _ in st['variables']] == ['lod', 'dlatent_avg']                 # G's own variables (:194-195)
    syn = st['components']['synthesis'].state_v4()
    assert syn['name'] == 'G_synthesis' and syn['build_func_name'] == 'G_synthesis_stylegan2'
    names = [n for n, _ in syn['variables']]
    assert '4x4/Const/const' in names and '16x16/Conv0_up/mod_weight' in names and 'noise0' in names
    assert all(isinstance(v, np.ndarray) and v.dtype == np.float32 for _, v in syn['variables'])
    w = dict(syn['variables'])['16x16/Conv1/weight']
    assert w.ndim == 4 and w.shape[:2] == (3, 3) and w.shape[2] == w.shape[3]        # HWIO, like get_weight (:23)
    assert [n for n, _ in D.state_v4()['variables']][:2] == ['16x16/FromRGB/weight', '16x16/FromRGB/bias'] and not D.state_v4()['components']


def test_reference_layout_round_trip(tmp_path):
    from inclusivegan_amd.training import misc
    G, D = _nets()
    Gs = G.clone('Gs')
    with torch.no_grad():
        Gs.vars['dlatent_avg'].fill_(0.25)
    f = str(tmp_path / 'network-snapshot-000012.pkl')
    misc.save_pkl((G, D, Gs), f, reference_layout=True, build_module_src='# reference module text goes here')
    globals_ = {a for op, a, _ in pickletools.genops(open(f, 'rb').read()) if op.name in ('SHORT_BINUNICODE', 'BINUNICODE', 'GLOBAL') and isinstance(a, str)}
    assert 'dnnlib.tflib.network' in globals_ and 'Network' in globals_
    assert not any('inclusivegan_amd' in g for g in globals_)                       # nothing of this package's class paths inside
    import sys
    assert 'dnnlib.tflib.network' not in sys.modules                                 # the temporary stand-in modules are gone
    loaded = misc.load_pkl(f)
    assert isinstance(loaded, tuple) and len(loaded) == 3
    G2, D2, Gs2 = misc.as_networks(loaded, device='cpu')
    for a, b in ((G, G2), (D, D2), (Gs, Gs2)):
        assert a.name == b.name and list(a.vars) == list(b.vars) and dict(a.static_kwargs) == dict(b.static_kwargs)
        assert all(torch.equal(a.vars[n], b.vars[n]) for n in a.vars)
    assert float(Gs2.vars['dlatent_avg'][0]) == 0.25
    # older state versions of the reference (network.py:277) are accepted too
    st = G.state_v4(); st['version'] = 2
    from inclusivegan_amd.dnnlib.tflib.network import network_from_state
    assert torch.equal(network_from_state(st, device='cpu').vars['G_mapping/Dense0/weight'], G.vars['G_mapping/Dense0/weight'])


def test_plain_pickle_round_trip_and_missing_variable(tmp_path):
    from inclusivegan_amd.training import misc
    import pytest
    G, D = _nets()
    f = str(tmp_path / 'own.pkl')
    misc.save_pkl((G, D), f)
    G2, D2 = misc.load_pkl(f)
    assert all(torch.equal(G.vars[n], G2.vars[n]) for n in G.vars) and all(torch.equal(D.vars[n], D2.vars[n]) for n in D.vars)
    st = D.state_v4()
    st['variables'] = st['variables'][1:]
    with pytest.raises(KeyError):
        D.load_state_v4(st)


def test_resume_bookkeeping_from_log(tmp_path):
    """misc.py:147-187: the snapshot's kimg comes from its file name, the elapsed time from the matching tick line of log.txt."""
    from inclusivegan_amd.training import misc
    from inclusivegan_amd.dnnlib.util import format_time
    line = 'tick %-5d kimg %-8.1f lod %-5.2f minibatch %-4d time %-12s sec/tick %-7.1f sec/kimg %-7.2f maintenance %-6.1f gpumem %.1f' % (
        7, 150.1, 0.0, 12, format_time(3 * 3600 + 25 * 60 + 7), 100.0, 5.0, 1.0, 3.2)
    (tmp_path / 'log.txt').write_text('something else\n' + line + '\n')
    kimg, secs = misc.resume_kimg_time(str(tmp_path / 'network-snapshot-000150.pkl'))
    assert kimg == 150.1 and secs == 3 * 3600 + 25 * 60 + 7
    assert misc.time_to_seconds('1d 02h 03m') == ((24 + 2) * 60 + 3) * 60 and misc.time_to_seconds('04m 05s') == 245.0
