"""-m gpu: every C-ABI kernel against a plain PyTorch fp32 reference of the same op (tests/kernel_checks.py)."""
import pytest
import torch

from tests import kernel_checks as kc

pytestmark = pytest.mark.gpu
_CHECKS = kc.all_checks() if torch.cuda.is_available() else []


@pytest.mark.parametrize("name,fn,tol", _CHECKS, ids=[c[0] for c in _CHECKS])
def test_kernel(name, fn, tol):
    err = fn()
    torch.cuda.synchronize()
    assert err <= tol, f"{name}: max-rel error {err:.3e} > {tol:.1e}"


def test_gpu_available():
    """Fails (not skips) on a GPU box without a visible device, so a silent CPU-only run cannot look green."""
    assert torch.cuda.is_available()
    from idm_vton_amd import ffi
    ffi.lib()
