"""No-GPU checks of the C-ABI library: it loads, exports every symbol include/pandepth_amd.h
declares, and refuses to create a context without a gfx950 device (no CPU fallback)."""
import os
import re

import pytest

import pandepth_amd as pda
from pandepth_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = set()
    for h in ("pandepth_amd.h", "pandepth_amd_dev.h"):          # the drop-in boundary + the development / measurement entry points
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pd_[a-z0-9_]+)\s*\(", text))
    return sorted(names)


def test_the_boundary_header_holds_no_development_entry_points():
    text = open(os.path.join(ROOT, "include", "pandepth_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name in ("pd_set_param", "pd_push_bgzf_units", "pd_x_bgzf_inflate"):
        assert not re.search(r"\b%s\s*\(" % name, text), name


def test_library_exports_every_declared_symbol():
    L = pda.load()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), n
    assert sorted(capi.EXPORTS) == names
    assert L.pd_abi_version() == 3


def test_no_cpu_fallback():
    from conftest import HAS_GPU
    if HAS_GPU:
        pytest.skip("GPU present")
    with pytest.raises(pda.PdError) as ei:
        pda.Engine([1000, 2000])
    assert ei.value.code == -2 and "no CPU fallback" in str(ei.value)
