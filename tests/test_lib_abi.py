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


def test_encoder_stack_struct_mirrors_match_the_header(hip_lib):
    """The ctypes mirrors in edgedict_amd/encoder_stack.py must have the C structs' sizes."""
    from edgedict_amd import encoder_stack as es
    assert ctypes.sizeof(es.StackLayer) == hip_lib.edgedict_stack_struct_bytes(0)
    assert ctypes.sizeof(es.StackDesc) == hip_lib.edgedict_stack_struct_bytes(1)


def test_encoder_stack_validates_before_launching(hip_lib):
    from edgedict_amd import encoder_stack as es
    layers = (es.StackLayer * 1)()
    d = es.StackDesc()
    d.B, d.H, d.L, d.chunk, d.T0, d.I0 = 4, 48, 1, 8, 10, 16     # H % 32 != 0
    d.layers = ctypes.cast(layers, ctypes.POINTER(es.StackLayer))
    assert hip_lib.edgedict_stack_forward(ctypes.byref(d), None) == -1
    assert b"H % 32" in hip_lib.edgedict_last_error()
    assert hip_lib.edgedict_stack_workspace_bytes(ctypes.byref(d)) > 0
    rc = hip_lib.edgedict_stack_pack_weights(None, None, None, None, 64, 16, None, None, None,
                                             None, None, None)
    assert rc == -1


def test_streams_busy_probe_is_safe_without_a_device_and_rejects_null(hip_lib):
    """edgedict_streams_busy (the isolation fixture's probe): a null mask is an argument error, and on a box without a
    device it returns a status (nothing has been created, so nothing can be busy) instead of touching HIP state."""
    assert hip_lib.edgedict_streams_busy(None) == -1
    mask = ctypes.c_uint(123)
    rc = hip_lib.edgedict_streams_busy(ctypes.byref(mask))
    import torch
    if torch.cuda.is_available():
        assert rc == 0
    else:
        assert mask.value == 0        # cleared before anything else happens; rc may report the missing device
