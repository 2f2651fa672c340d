"""CPU: the C-ABI library loads and exports exactly what include/cpf.h declares; no compute calls."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest


def test_header_and_binding_agree():
    from cudaparticlesfoam_amd import _lib
    assert _lib.header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    from cudaparticlesfoam_amd import _lib
    lib = _lib.load()                                     # resolves every symbol or raises
    assert lib.cpf_abi_version() == 1
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = {l.split()[-1] for l in out.stdout.splitlines() if " T " in l}
    assert set(_lib.header_symbols()) <= exported
    # nothing of the oracle is linked into the product
    assert not any(s.startswith(("orc_", "cw_", "ref_")) for s in exported)


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "cudaparticlesfoam_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".H")):
                text = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "liboracle" not in text, f


def test_fails_loudly_without_a_gpu():
    """No silent CPU fallback: on a box without a HIP device creating a context raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from cudaparticlesfoam_amd import _lib
    from cudaparticlesfoam_amd.api import Context
    with pytest.raises(_lib.CpfError) as e:
        Context(0)
    assert e.value.status == _lib.CPF_ERR_HIP and "no HIP device" in str(e.value)


def test_null_arguments_are_rejected():
    from cudaparticlesfoam_amd import _lib
    lib = _lib.load()
    assert lib.cpf_create(0, None) == _lib.CPF_ERR_ARG
    assert lib.cpf_step(None, 1e-4, 0.0, 1, 0) == _lib.CPF_ERR_ARG
    assert lib.cpf_destroy(None) == _lib.CPF_OK
    assert lib.cpf_last_error(None) is not None
