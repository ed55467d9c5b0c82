"""dsp_ffn_split (the Conformer feed-forward module in one launch) against an fp64 restatement of fairseq's FeedForwardModule
(modules/conformer_layer.py:140-146) as ConformerEncoderLayer calls it (x + 0.5 * ffn(x), :254-281), and against the two-GEMM path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,H,act,with_ln", [(32, 197, 2048, "silu", True), (2, 100, 2048, "silu", True), (40, 300, 2048, "relu", False),
                                              (3, 129, 1024, "gelu", True), (8, 64, 512, "silu", True)])
def test_ffn_fused_matches_fp64_and_the_two_gemm_path(B, T, H, act, with_ln):
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(B + T)
    C = 256
    ln = torch.nn.LayerNorm(C).to(dev).eval() if with_ln else None
    l1, l2 = torch.nn.Linear(C, H).to(dev).eval(), torch.nn.Linear(H, C).to(dev).eval()
    with torch.no_grad():
        if ln is not None:
            ln.weight.normal_(1.0, 0.2); ln.bias.normal_(0.0, 0.2)
        x = torch.randn(B, T, C, device=dev) * 1.7 + 0.2
        got = decode_ops.ffn_fused(x, ln, l1, l2, act, residual=x, alpha=0.5)
        assert got is not None and got.shape == x.shape
        xin = x if ln is None else decode_ops.layer_norm(x, ln)
        two = decode_ops.linear(decode_ops.linear(xin, l1, act=act), l2, residual=x, alpha=0.5)
        xd = x.double()
        h = xd if ln is None else torch.nn.functional.layer_norm(xd, (C,), ln.weight.double(), ln.bias.double(), ln.eps)
        h = torch.nn.functional.linear(h, l1.weight.double(), l1.bias.double())
        h = {"silu": torch.nn.functional.silu, "relu": torch.relu, "gelu": torch.nn.functional.gelu}[act](h)
        ref = xd + 0.5 * torch.nn.functional.linear(h, l2.weight.double(), l2.bias.double())
    scale = ref.abs().max().item()
    e_f = (got.double() - ref).abs().max().item() / scale
    e_t = (two.double() - ref).abs().max().item() / scale
    assert e_f < 2e-6 and e_f < 3 * max(e_t, 3e-7), (e_f, e_t)


def test_ffn_fused_without_residual_and_not_served_shapes():
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    l1, l2 = torch.nn.Linear(256, 1024).to(dev).eval(), torch.nn.Linear(1024, 256).to(dev).eval()
    x = torch.randn(4, 90, 256, device=dev)
    with torch.no_grad():
        got = decode_ops.ffn_fused(x, None, l1, l2, "relu")
        ref = l2(torch.relu(l1(x)))
        assert decode_ops.ffn_fused(torch.randn(4, 90, 512, device=dev), None, torch.nn.Linear(512, 1024).to(dev).eval(),
                                    torch.nn.Linear(1024, 512).to(dev).eval(), "relu") is None          # 512 channels: the two-GEMM path
    torch.testing.assert_close(got, ref, rtol=2e-5, atol=2e-5)
    assert decode_ops.ffn_fused(x.requires_grad_(), None, l1, l2, "relu") is None                      # under autograd: torch
