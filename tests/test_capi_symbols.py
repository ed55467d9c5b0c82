"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, and exports every symbol that
include/daspeech_dag.h declares; the Python operator surface has the reference's eight names.  No compute calls."""
import ctypes
import inspect
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    out = []
    for fn in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not fn.endswith(".h"):
            continue
        text = open(os.path.join(ROOT, "include", fn)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out += re.findall(r"\b(dsp_[a-z0-9_]+)\s*\(", text)
    return sorted(set(out))


@pytest.fixture(scope="module")
def lib_path():
    from daspeech_amd import build
    return build.build()


def test_header_symbols_are_exported(lib_path):
    lib = ctypes.CDLL(lib_path)
    names = declared_symbols()
    assert len(names) >= 8
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/*.h but not exported by {lib_path}"


def test_binding_covers_header(lib_path):
    from daspeech_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    lib = _lib.load()
    assert lib.dsp_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.dsp_last_error(), bytes)
    assert lib.dsp_dag_workspace_bytes(1, 2, 3, 4) >= 0


def test_operator_surface_matches_reference_names():
    # DASpeech/custom_ops/__init__.py:1
    from daspeech_amd import custom_ops
    names = ["dag_loss", "dag_loss_with_alpha_beta", "dag_best_alignment", "dag_logsoftmax_gather_inplace",
             "torch_dag_loss", "torch_dag_best_alignment", "torch_dag_logsoftmax_gather_inplace", "logsumexp_keepdim"]
    for n in names:
        assert callable(getattr(custom_ops, n))
    import sys
    mod = sys.modules["daspeech_amd.custom_ops.dag_loss"]      # the package attribute is the function, as in the reference
    for cls in ("DagLossFunc", "DagLossWithAlphaBetaFunc", "DagBestAlignmentFunc", "DagLogsoftmaxGatherFunc"):
        assert inspect.isclass(getattr(mod, cls))
    assert mod.DagLossFunc.config == 1 and mod.DagLossFunc.config1 == 2 and mod.DagLossFunc.config2 == 2


def test_hip_ops_refuse_cpu_tensors():
    import torch
    from daspeech_amd import custom_ops
    m = torch.zeros(1, 2, 3); k = torch.zeros(1, 3, 2)
    ol = torch.tensor([3]); tl = torch.tensor([2])
    with pytest.raises(RuntimeError):
        custom_ops.dag_loss(m, k, ol, tl)
    with pytest.raises(RuntimeError):
        custom_ops.dag_best_alignment(m, k, ol, tl)
    with pytest.raises(RuntimeError):
        custom_ops.dag_logsoftmax_gather_inplace(torch.zeros(1, 3, 5), torch.zeros(1, 3, 2, dtype=torch.long))


def test_dispatch_questions_follow_the_window_and_the_kernel_pin(lib_path):
    """Host-side logic of the C ABI (no device call): which windows take PITCHED rows (dsp_dag_pitch_supported: op 0 = dag_loss forward / backward,
    op 1 = dag_best_alignment), which need no trace tensor (dsp_dag_alignment_trace_optional), and how the workspace questions grow — under the
    default dispatch and under the per-thread `dp_path` pins the tests use.  r06: every strip family up to a window of 128 takes row pitches."""
    from daspeech_amd import _lib
    lib = _lib.load()
    try:
        _lib.set_option("dp_path", 0)
        for TR in (1, 8, 32, 33, 64, 65, 128):
            assert lib.dsp_dag_pitch_supported(0, 1030, TR) == 1, TR
            assert lib.dsp_dag_pitch_supported(1, 1030, TR) == 1, TR
            assert lib.dsp_dag_alignment_trace_optional(1032, TR) == 1, TR
        assert lib.dsp_dag_pitch_supported(0, 1030, 129) == 0 and lib.dsp_dag_pitch_supported(1, 1030, 1029) == 0      # dense window: dense tensors
        assert lib.dsp_dag_pitch_supported(0, 0, 32) == 0
        assert lib.dsp_dag_pitch_supported(1, 9000, 32) == 0                        # the narrow back-trace keeps the path image in LDS: L <= 8192
        assert lib.dsp_dag_alignment_trace_optional(4096, 4095) == 1                # blocked max-plus kernels: no trace either
        for pin, narrow, wide in ((5, 1, 0), (8, 0, 1), (2, 0, 0), (1, 0, 0), (9, 0, 0)):
            _lib.set_option("dp_path", pin)
            assert lib.dsp_dag_pitch_supported(0, 1030, 32) == narrow, pin
            assert lib.dsp_dag_pitch_supported(0, 1030, 64) == wide and lib.dsp_dag_pitch_supported(0, 1030, 100) == wide, pin
        _lib.set_option("dp_path", 7)
        assert lib.dsp_dag_pitch_supported(1, 1030, 32) == 1 and lib.dsp_dag_pitch_supported(1, 1030, 100) == 1
        _lib.set_option("dp_path", 2)
        assert lib.dsp_dag_pitch_supported(1, 1030, 64) == 0 and lib.dsp_dag_alignment_trace_optional(1032, 64) == 0      # log-space strips + trace walk
        _lib.set_option("dp_path", 0)
        # workspaces: monotone in the batch, and a pitched graph (L rounded up to 4) never asks for less than the dense one
        for TR in (32, 64, 128, 1029):
            a = lib.dsp_dag_workspace_bytes(4, 50, 1030, TR); b = lib.dsp_dag_workspace_bytes(8, 50, 1030, TR); c = lib.dsp_dag_workspace_bytes(4, 50, 1032, TR)
            assert 0 <= a <= b and a <= c, (TR, a, b, c)
            a = lib.dsp_dag_alignment_workspace_bytes(4, 50, 1030, TR); b = lib.dsp_dag_alignment_workspace_bytes(8, 50, 1030, TR)
            assert 0 <= a <= b, (TR, a, b)
    finally:
        _lib.set_option("dp_path", 0)
