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


def _integration_md_binding():
    """The ctypes stub INTEGRATION.md section 2 tells a morphik-core maintainer to add, extracted from the document and executed
    as it stands (only the library NAME is replaced by the built file's path).  -> its namespace."""
    import os
    import re

    import morphik_core_amd as m

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```python\n(.*?)```", text, re.S) if "mv_abi_version" in b]
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one binding stub"
    code = blocks[0]
    assert 'C.CDLL("libmvmaxsim.so")' in code
    ns = {}
    exec(compile(code.replace('"libmvmaxsim.so"', repr(m.library_path())), "INTEGRATION.md#binding", "exec"), ns)
    return ns, code


def test_integration_md_binding_loads_and_fails_loudly_without_a_gpu():
    """the documented binding asserted ABI 3 against a header at 4 -- nothing executed the document.  Now the
    stub is run: it must load the built library, pass ITS OWN abi assertion against the shipped header, and (on a box without a
    GPU) `create` must raise instead of falling back to anything."""
    import re

    import pytest
    import torch

    from morphik_core_amd import _lib

    ns, code = _integration_md_binding()  # the import itself runs `assert L.mv_abi_version() == N`
    lit = [int(x) for x in re.findall(r"mv_abi_version\(\) == (\d+)", code)]
    assert lit == [_lib.MV_ABI_VERSION], f"INTEGRATION.md asserts ABI {lit}, the header / binding are at {_lib.MV_ABI_VERSION}"
    # no other stale literal anywhere in the documents
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for doc in ("INTEGRATION.md", "DESIGN.md", "README.md"):
        for x in re.findall(r"abi_version\(\) == (\d+)", open(os.path.join(root, doc)).read()):
            assert int(x) == _lib.MV_ABI_VERSION, (doc, x)
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no HIP device|MI355X"):
            ns["create"](16)


def test_host_pin_budget_is_readable_without_a_gpu_and_honours_the_environment_cap(monkeypatch):
    """mv_host_pin_budget_bytes(): what mv_index_create checks a pinned-host exact tier against BEFORE pinning (a container past its memory
    cgroup limit is killed, not told).  No GPU needed: it reads the cgroup files and /proc/meminfo."""
    from morphik_core_amd import _lib

    L = _lib.lib()
    b = int(L.mv_host_pin_budget_bytes())
    assert b > 0
    avail = [int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable:")][0]
    assert b <= avail  # never more than the machine has free; less when a cgroup limit is tighter
    monkeypatch.setenv("MV_HOST_EXACT_MAX_BYTES", "123456789")
    assert int(L.mv_host_pin_budget_bytes()) == min(123456789, b) or int(L.mv_host_pin_budget_bytes()) <= 123456789
    monkeypatch.setenv("MV_HOST_EXACT_MAX_BYTES", "0")
    assert int(L.mv_host_pin_budget_bytes()) == 0
