"""CPU checks of the boundary: the library loads, exports every symbol include/*.h declares,
and refuses to run without a device (no silent CPU fallback)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "revrand_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from revrand_amd import _hip
    lib = _hip.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "missing export: %s" % name
    # and the ctypes table covers the header exactly
    assert sorted(_hip.SIGNATURES) == declared
    assert lib.rr_abi_version() == 1


def test_no_cpu_fallback_without_device():
    from revrand_amd import _hip
    from revrand_amd.basis_functions import RandomRBF
    if _hip.device_available():
        pytest.skip("a GPU is present")
    b = RandomRBF(nbases=4, Xdim=2, random_state=0)
    with pytest.raises(_hip.HipError):
        b.transform(np.zeros((3, 2)))
    with pytest.raises(_hip.HipError):
        b.gram(np.zeros((3, 2)), np.zeros(3))


def test_product_does_not_import_oracle():
    """Tier rule 3: nothing under revrand_amd/ may import or execute the oracle."""
    pkg = os.path.join(ROOT, "revrand_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "revrand_oracle" not in text and "oracle/" not in text, f
                assert "/root/reference" not in text, f


def test_single_hip_runtime_with_torch():
    """Loading the library BEFORE torch must not leave two HIP/HSA runtimes in the process (torch would then
    see no GPU): the loader shares torch's bundled runtime when torch is installed."""
    import subprocess
    import sys
    pytest.importorskip("torch")
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from revrand_amd import _hip\n"
        "_hip.load_library()\n"
        "import torch\n"
        "maps = open('/proc/self/maps').read().splitlines()\n"
        "for key in ('libamdhip64', 'libhsa-runtime64'):\n"
        "    paths = sorted({l.split()[-1] for l in maps if key in l})\n"
        "    assert len(paths) == 1, paths\n"
        "print('ok')\n" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
