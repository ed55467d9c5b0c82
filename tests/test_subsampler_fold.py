"""Conv1dSubsampler (speech_to_text/modules/convolution.py:13-59): the stride-2 convolutions as stride-1 convolutions over frame
pairs — the rewriting the eval-mode fp32 path uses to run them on the matrix-core kernel (daspeech_amd/models/daspeech.py)."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from daspeech_amd.models.daspeech import Conv1dSubsampler


@pytest.mark.parametrize("C,Cout,K,T", [(80, 64, 5, 37), (512, 32, 5, 20), (16, 8, 3, 9), (16, 8, 7, 12), (16, 8, 5, 1), (16, 8, 5, 2)])
def test_stride2_fold_is_the_same_convolution(C, Cout, K, T):
    torch.manual_seed(C + K + T)
    conv = nn.Conv1d(C, Cout, K, stride=2, padding=K // 2).double()
    x = torch.randn(3, T, C, dtype=torch.float64)
    with torch.no_grad():
        y = conv(x.transpose(1, 2)).transpose(1, 2)
        wf, Cp = Conv1dSubsampler._fold_stride2(conv)
        assert Cp >= 2 * C and (Cp in (128, 256, 512) or Cp % 512 == 0)
        xx = F.pad(x, (0, 0, 0, 1)) if T % 2 else x
        x2 = F.pad(xx.reshape(3, (T + 1) // 2, 2 * C), (0, Cp - 2 * C))
        y2 = F.conv1d(x2.transpose(1, 2), wf.double(), conv.bias, padding=wf.shape[2] // 2).transpose(1, 2)
    assert y.shape == y2.shape
    assert float((y - y2).abs().max()) <= 1e-5          # (the folded weights are kept in fp32)


@pytest.mark.gpu
def test_subsampler_matrix_core_path_matches_fp64():
    """Released widths (80 -> 1024 -> GLU -> 512 -> 512 -> GLU -> 256), odd and even lengths: the split path against the fp64 convolution,
    within 2e-6 of the output range (fp32 accuracy: operands split into fp16 hi / lo, fp32 accumulation)."""
    from daspeech_amd import decode_ops
    torch.manual_seed(5)
    sub = Conv1dSubsampler(80, 1024, 256).cuda().eval()
    for T in (301, 420, 7):
        x = torch.randn(3, T, 80, device="cuda")
        lens = torch.tensor([T, T - 3, max(1, T - 100)], device="cuda")
        old = decode_ops.set_split_gemm(True)
        try:
            with torch.no_grad():
                y, ol = sub(x, lens)
        finally:
            decode_ops.set_split_gemm(old)
        assert getattr(sub, "_split", None) is not None          # the matrix-core path really ran
        ref = x.double().transpose(1, 2)
        with torch.no_grad():
            for conv in sub.conv_layers:
                ref = F.glu(F.conv1d(ref, conv.weight.double(), conv.bias.double(), stride=2, padding=conv.kernel_size[0] // 2), dim=1)
        ref = ref.transpose(1, 2)
        assert y.shape == ref.shape and torch.equal(ol, sub.out_lengths(lens))
        err = float((y.double() - ref).abs().max()); scale = float(ref.abs().max())
        assert err <= 2e-6 * scale, (T, err, scale)
