"""INTEGRATION.md Option B, executed: the `_Kernel` ctypes shim a reference maintainer would paste over `get_dag_kernel()`
(DASpeech/custom_ops/dag_loss.py:37-64) is cut out of INTEGRATION.md as it stands, exec'd, and its four pybind-shaped methods
(dag_loss.cpp:24-29) are called the way the reference's autograd Functions call them, against the reference-generated goldens."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _shim_source():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = md[md.index("## Option B"):]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "class _Kernel" in code and "def get_dag_kernel" in code
    return code


def test_shim_block_names_only_exported_symbols():
    """(CPU) every dsp_* symbol the shim calls is declared in include/daspeech_dag.h and bound in _lib.SIGNATURES."""
    from daspeech_amd import _lib
    code = _shim_source()
    used = set(re.findall(r"_lib\.(dsp_[a-z_0-9]+)", code))
    header = open(os.path.join(ROOT, "include", "daspeech_dag.h")).read()
    assert used and all(u in _lib.SIGNATURES and u in header for u in used), used


@pytest.mark.gpu
def test_option_b_kernel_shim_runs_verbatim_against_goldens():
    from daspeech_amd import _lib
    _lib.load()                                                 # (torch's HIP runtime first, as in the product)
    code = _shim_source().replace("/path/to/libdaspeech_hip.so", _lib.SO_PATH)
    ns = {}
    exec(compile(code, "INTEGRATION.md:OptionB", "exec"), ns)
    kern = ns["get_dag_kernel"]()
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    for name in ("dag_banded", "dag_full", "dag_ragged", "dag_ties", "dag_forceemit"):
        g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
        m, k, ol, tl = t(g["match"]), t(g["links"]), t(g["out_len"]), t(g["tgt_len"])
        B = m.shape[0]
        ar = torch.arange(B, device=dev)
        fin = torch.from_numpy(g["finite"]).to(dev)
        # dag_loss.py:105-110  alpha, beta = kernel.dag_loss(..., require_gradient, config); loss = alpha[b, T_b-1, L_b-1] (beta[b,0,0] with grad)
        alpha, beta = kern.dag_loss(m, k, ol, tl, True, 1)
        ref = torch.from_numpy(g["loss"]).to(dev)
        la = alpha[ar, (tl - 1).clamp(min=0), (ol - 1).clamp(min=0)]
        torch.testing.assert_close(la[fin].double(), ref[fin], rtol=1e-5, atol=1e-5)
        torch.testing.assert_close(beta[ar, 0, 0][fin].double(), ref[fin], rtol=1e-5, atol=1e-5)
        alpha0, _ = kern.dag_loss(m, k, ol, tl, False, 1)
        torch.testing.assert_close(alpha0[ar, (tl - 1).clamp(min=0), (ol - 1).clamp(min=0)][fin].double(), ref[fin], rtol=1e-5, atol=1e-5)
        # dag_loss.py:157  grad_match, grad_links = kernel.dag_loss_backward(grad_output, alpha, beta, match_all, links, ol, tl, c1, c2)
        go = fin.to(torch.float32)
        gm, gl = kern.dag_loss_backward(go, alpha, beta, m, k, ol, tl, 2, 2)
        np.testing.assert_allclose(gm.cpu().numpy(), g["grad_match"], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(gl.cpu().numpy(), g["grad_links"], rtol=2e-4, atol=1e-6)
        # dag_loss.py:227  alpha, path = kernel.dag_best_alignment(match_all, links, ol, tl, config); path.to(torch.long)
        _, path = kern.dag_best_alignment(m, k, ol, tl, 1)
        ok = g["path_valid"]
        np.testing.assert_array_equal(path.to(torch.long).cpu().numpy()[ok], g["path"][ok])
    # dag_loss.py:270  selected = kernel.logsoftmax_gather(word_ins_out, select_idx, require_gradient)  (in place softmax when it is)
    g = dict(np.load(os.path.join(GOLDEN, "lsg_f32.npz")))
    x, tgt = t(g["logits"]), t(g["targets"])
    idx = tgt.unsqueeze(1).expand(-1, x.shape[1], -1)                       # the criteria's stride-0 expand (nat_dag_loss.py:127)
    sel = kern.logsoftmax_gather(x, idx, True)
    np.testing.assert_allclose(sel.cpu().numpy(), g["match"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(x.cpu().numpy(), g["softmax"], rtol=1e-5, atol=1e-6)      # the buffer now holds the softmax (logsoftmax_gather.cu:303-305)
