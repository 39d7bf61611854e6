"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol that
include/nerf_atlas_amd.h declares (no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

from conftest import REPO


def header_symbols():
    src = open(os.path.join(REPO, "include", "nerf_atlas_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(na_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_binding_table():
    from nerf_atlas_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_builds_and_exports_every_symbol():
    from nerf_atlas_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert lib.na_version() == 100


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nerf_atlas_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NativeLibraryMissing):
        _lib.load()


def test_ops_refuse_cpu_tensors():
    import torch
    from nerf_atlas_amd import ops
    with pytest.raises(ValueError):
        ops.view_elaz(torch.zeros(4, 3))


def test_desc_struct_layout():
    import ctypes
    from nerf_atlas_amd import _lib
    assert ctypes.sizeof(_lib.NaMlpDesc) == 40
