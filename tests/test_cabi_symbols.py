"""The C-ABI library loads on a CPU-only host and exports every symbol include/sgnn_hip.h declares
(no compute calls here).  Also pins the ctypes prototype table to the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'sgnn_hip.h')


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'\b(sgnn_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ['sgnn_hash_build', 'sgnn_rulebook_subm3', 'sgnn_rulebook_down2', 'sgnn_conv_fwd',
                 'sgnn_conv_bwd_weight', 'sgnn_bn_fwd', 'sgnn_bn_bwd', 'sgnn_compact_sigmoid', 'sgnn_expand8_coords',
                 'sgnn_hash_lookup', 'sgnn_concat_rows', 'sgnn_sparse_to_dense', 'sgnn_linear_fwd']:
        assert must in names


def test_library_exports_every_declared_symbol():
    from sgnn_amd import _lib
    if not os.path.isfile(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, 'declared in sgnn_hip.h but not exported: %s' % missing
    assert _lib.load().sgnn_arch() == b'gfx950'
    assert _lib.load().sgnn_version() >= 100


def test_ctypes_prototypes_cover_the_header():
    from sgnn_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_functions()
    # argument counts must match the header (every int-returning entry point ends with the stream)
    src = re.sub(r'/\*.*?\*/', '', open(HEADER).read(), flags=re.S)
    for name, (res, args) in _lib.PROTOTYPES.items():
        m = re.search(r'\b%s\s*\(([^)]*)\)' % name, src)
        params = [p for p in m.group(1).split(',') if p.strip() and p.strip() != 'void']
        assert len(params) == len(args), (name, len(params), len(args))


def test_operators_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from sgnn_amd import _lib
    import sgnn_amd.scn as scn
    with pytest.raises(_lib.SgnnError):
        _lib.require_gpu()
    layer = scn.InputLayer(3, [8, 8, 8], mode=0)
    with pytest.raises(RuntimeError):
        layer([torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, 1)])
