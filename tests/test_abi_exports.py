"""The C-ABI library loads on a CPU-only host and exports every symbol include/mvmaxsim.h declares
(no compute calls without a GPU), and fails loudly -- not silently -- when no device exists."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    import morphik_core_amd as m

    if not os.path.exists(m.library_path()):
        m.build_library()
    return m.library_path()


def test_exports_match_header(libpath):
    hdr = open(os.path.join(ROOT, "include", "mvmaxsim.h")).read()
    declared = set(re.findall(r"MV_API\s+[\w\s\*]+?\b(mv_\w+)\s*\(", hdr))
    assert len(declared) >= 25
    lib = ctypes.CDLL(libpath)
    missing = sorted(s for s in declared if not hasattr(lib, s))
    assert not missing, missing
    from morphik_core_amd import _lib

    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_loads_and_reports_version_and_errors(libpath):
    from morphik_core_amd import MvError, _lib

    L = _lib.lib()
    assert L.mv_version().decode().startswith("mvmaxsim")
    import torch

    if not torch.cuda.is_available():
        from morphik_core_amd.index import MvIndex

        with pytest.raises(MvError) as e:
            MvIndex(capacity_pages=4, stride_rows=16)  # no GPU here: loud failure, no CPU fallback
        assert "HIP" in str(e.value) or "device" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "morphik-core_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "libmvoracle" not in src and "mv_oracle.c\"" not in src, f
