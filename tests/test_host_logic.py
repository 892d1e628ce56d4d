"""CPU-side checks: the C-ABI library loads and exports every symbol the header declares,
and the host logic (interpolation tables, registry, parameter groups, lr schedule)."""
import os
import re

import pytest
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, 'include', 'vitpose_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'^\s*(?:int|const char\*)\s+(vp_\w+)\s*\(', src, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from vitpose_amd import _lib
    L = _lib.lib()   # raises when the library is absent: there is no fallback
    declared = header_functions()
    assert len(declared) > 25
    for name in declared:
        assert hasattr(L, name), f'{name} declared in include/vitpose_hip.h but not exported'
    assert sorted(_lib.exported_symbols()) == declared
    assert L.vp_abi_version() == _lib.ABI_VERSION == 4 and L.vp_build_arch() == b'gfx950'


@pytest.mark.parametrize('n_in,scale,align', [(16, 4, False), (12, 4, False), (12, 2, True), (5, 3, False), (7, 1, False)])
def test_bilinear_tables_match_torch(n_in, scale, align):
    from vitpose_amd.head_engine import bilinear_tables
    n_out = n_in * scale
    idx, w, tidx, tw = bilinear_tables(n_in, n_out, scale if scale > 1 else None, align)
    x = torch.randn(1, 1, n_in, 1, dtype=torch.float64)
    if scale > 1:
        ref = F.interpolate(x, scale_factor=(scale, 1), mode='bilinear', align_corners=align)[0, 0, :, 0]
    else:
        ref = x[0, 0, :, 0]
    v = x[0, 0, :, 0]
    got = v[idx[:, 0].long()] * w[:, 0].double() + v[idx[:, 1].long()] * w[:, 1].double()
    torch.testing.assert_close(got, ref, rtol=1e-6, atol=1e-6)
    # the transposed lists hold exactly the same coefficients
    dense = torch.zeros(n_out, n_in, dtype=torch.float64)
    for u in range(n_out):
        for a in range(2):
            dense[u, idx[u, a]] += float(w[u, a])
    dense_t = torch.zeros(n_out, n_in, dtype=torch.float64)
    for j in range(n_in):
        for a in range(tidx.shape[1]):
            if tidx[j, a] >= 0:
                dense_t[tidx[j, a], j] += float(tw[j, a])
    torch.testing.assert_close(dense_t, dense)
