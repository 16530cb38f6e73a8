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


def _runtime_probe(env_mode, torch_first):
    import subprocess
    import sys
    code = (
        "import sys, os; sys.path.insert(0, %r)\n"
        "%s"
        "from revrand_amd import _hip\n"
        "_hip.load_library()\n"
        "first = _hip.hip_runtime_path()\n"
        "%s"
        "maps = open('/proc/self/maps').read().splitlines()\n"
        "for key in ('libamdhip64', 'libhsa-runtime64'):\n"
        "    paths = sorted({l.split()[-1] for l in maps if key in l})\n"
        "    assert len(paths) == 1, paths\n"
        "print('RUNTIME', first, _hip.rccl_library_path())\n"
        % (ROOT, "import torch\n" if torch_first else "", "" if torch_first or env_mode != "torch" else "import torch\n"))
    env = dict(os.environ)
    env.pop("RR_HIP_RUNTIME", None)
    env.pop("RR_RCCL_LIB", None)
    if env_mode:
        env["RR_HIP_RUNTIME"] = env_mode
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "RUNTIME" in r.stdout, r.stderr[-2000:]
    _, runtime, rccl = r.stdout.strip().splitlines()[-1].split()
    return runtime, rccl


def test_hip_runtime_selection():
    """by default the library runs on the HIP runtime it was built against (/opt/rocm), whether or
    not a torch wheel is installed, and RCCL is then the system's; torch's bundled runtime (and its RCCL) only when torch
    is already in the process or RR_HIP_RUNTIME=torch asks for it -- one runtime per process in every case."""
    runtime, rccl = _runtime_probe(None, False)
    assert os.path.realpath(runtime).startswith(os.path.realpath("/opt/rocm")), runtime
    # the copy next to THAT runtime, by path (round 6: a bare dlopen("librccl.so.1") returns whichever copy is already in the
    # process -- the torch wheel's, once anything imported torch after the library was loaded; found by the full GPU suite)
    assert rccl == "None" or os.path.realpath(rccl).startswith(os.path.realpath("/opt/rocm")), rccl
    pytest.importorskip("torch")
    for mode, torch_first in (("torch", False), (None, True)):
        runtime, rccl = _runtime_probe(mode, torch_first)
        assert "/torch/lib/" in runtime and "/torch/lib/" in rccl, (mode, torch_first, runtime, rccl)
