"""CPU checks of the drop-in boundary: the shared library loads without a GPU and exports
every symbol include/icp_hip.h declares; the ctypes table matches the header; the product
path refuses to run without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "icp_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"^(?:int|const char \*)\s*(\w+)\(", text, flags=re.M)))


def test_header_symbols_exported():
    from icp_amd import _lib
    names = declared_functions()
    assert len(names) >= 25 and "icp_closest" in names and "nicp_solve" in names
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libicp_hip.so does not export %s" % n
    assert sorted(_lib.exported_symbols()) == names          # ctypes table covers the whole header
    _lib.load_library()
    assert lib.icp_version() >= 100


def test_flag_values_match_the_header():
    """The Python host's flag constants are the header's enumerators (ICP_COHERENT joined them in round 3)."""
    from icp_amd import _lib
    text = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    vals = {k: int(v) for k, v in re.findall(r"\b(ICP_[A-Z_]+)\s*=\s*(\d+)u", text)}
    for name in ("ICP_COMPARE_SQRT", "ICP_FORCE_BRUTE", "ICP_FORCE_GRID", "ICP_HINT_IDX", "ICP_NO_FUSE", "ICP_COHERENT"):
        assert vals[name] == getattr(_lib, name), name
    assert len({vals[n] for n in vals if n in dir(_lib)}) >= 6          # distinct bits


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from icp_amd import icp_3d, _lib
    with pytest.raises(_lib.IcpHipError):
        icp_3d.Match(np.zeros((4, 3)), np.zeros((4, 3)))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under icp_amd/ may reference it."""
    for dp, _, files in os.walk(os.path.join(ROOT, "icp_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "icp_oracle" not in text and "c_oracle" not in text and "oracle/" not in text, f


def test_error_reporting_without_gpu():
    from icp_amd import _lib
    lib = _lib.load_library()
    assert lib.icp_closest(None, None, 1, None, 1, 3, 0, None, None) == -1      # ICP_ERR_ARG, no launch
    assert b"null pointer" in lib.icp_last_error()


def test_header_is_plain_c_and_the_c_example_compiles(tmp_path):
    """include/icp_hip.h is a C header (the drop-in boundary is a C-ABI): strict C99, no warnings; the plain-C host of
    examples/ compiles against it (linking needs the library, which the GPU test of that program covers)."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "icp_hip.h")
    probe = tmp_path / "probe.c"
    probe.write_text('#include "icp_hip.h"\nint main(void) { return (int)sizeof(icp_comm) + (int)sizeof(nicp_cg_profile) > 0 ? 0 : 1; }\n')
    for src in (str(probe), os.path.join(ROOT, "examples", "rigid_c_abi.c")):
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.dirname(hdr), "-c", src,
                            "-o", str(tmp_path / "o.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
