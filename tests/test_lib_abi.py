"""CPU: the C-ABI shared library builds, loads and exports every declared symbol."""
import ctypes


def test_library_exports_every_declared_symbol(hip_lib):
    from edgedict_amd import _lib
    names = _lib.declared_symbols()
    assert "edgedict_rnnt_loss_forward" in names
    for n in names:
        assert hasattr(hip_lib, n), n


def test_abi_version_and_workspace_query(hip_lib):
    assert hip_lib.edgedict_abi_version() == 1
    n = hip_lib.edgedict_rnnt_workspace_bytes(64, 201, 65)
    assert n >= 7 * 64 * 201 * 65 * 4
    assert hip_lib.edgedict_rnnt_workspace_bytes(0, 1, 1) == 0


def test_invalid_arguments_return_status_not_abort(hip_lib):
    # no GPU needed: argument validation happens before any launch
    rc = hip_lib.edgedict_rnnt_loss_forward(None, 0, None, None, None, 1, 1, 2000, 8, 0,
                                            None, None, ctypes.c_float(1.0), None, None)
    assert rc == -1
    assert b"1024" in hip_lib.edgedict_last_error()


def test_product_path_refuses_cpu_tensors(hip_lib):
    import pytest
    import torch
    from edgedict_amd.loss import RNNTLoss
    acts = torch.zeros(1, 2, 3, 5)
    labels = torch.zeros(1, 2, dtype=torch.int32)
    lens = torch.tensor([2], dtype=torch.int32)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        RNNTLoss(check_lengths=False)(acts, labels, lens, lens)
