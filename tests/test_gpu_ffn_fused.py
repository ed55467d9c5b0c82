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


@pytest.mark.parametrize("B,T,outs", [(32, 197, (256, 256, 256)), (3, 100, (512,)), (2, 333, (256,))])
def test_linear_with_staged_layer_norm_matches_layer_norm_then_linear(B, T, outs):
    """dsp_linear_ln_split: LayerNorm applied while the row tile is staged, against LayerNorm + the (stacked) projections in fp64 and
    against the two-launch path."""
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(T)
    ln = torch.nn.LayerNorm(256).to(dev).eval()
    lins = [torch.nn.Linear(256, o).to(dev).eval() for o in outs]
    with torch.no_grad():
        ln.weight.normal_(1.0, 0.3); ln.bias.normal_(0.0, 0.3)
        x = torch.randn(B, T, 256, device=dev) * 2 + 0.5
        got = decode_ops.linear_ln(x, ln, lins)
        two = decode_ops.linear_fused(decode_ops.layer_norm(x, ln), lins) if len(lins) > 1 else (decode_ops.linear(decode_ops.layer_norm(x, ln), lins[0]),)
        xn = torch.nn.functional.layer_norm(x.double(), (256,), ln.weight.double(), ln.bias.double(), ln.eps)
        for g, t, l in zip(got, two, lins):
            ref = torch.nn.functional.linear(xn, l.weight.double(), l.bias.double())
            scale = ref.abs().max().item()
            e_g, e_t = (g.double() - ref).abs().max().item() / scale, (t.double() - ref).abs().max().item() / scale
            assert g.shape == ref.shape and e_g < 3e-6 and e_g < 3 * max(e_t, 3e-7), (e_g, e_t)


def test_ffn_fused_post_layer_norm_output():
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    ln, post = torch.nn.LayerNorm(256).to(dev).eval(), torch.nn.LayerNorm(256).to(dev).eval()
    l1, l2 = torch.nn.Linear(256, 2048).to(dev).eval(), torch.nn.Linear(2048, 256).to(dev).eval()
    with torch.no_grad():
        post.weight.normal_(1.0, 0.2); post.bias.normal_(0.0, 0.2)
        x = torch.randn(8, 150, 256, device=dev)
        plain = decode_ops.ffn_fused(x, ln, l1, l2, "silu", residual=x, alpha=0.5)
        out, out_ln = decode_ops.ffn_fused(x, ln, l1, l2, "silu", residual=x, alpha=0.5, post_ln=post)
        none, only_ln = decode_ops.ffn_fused(x, ln, l1, l2, "silu", residual=x, alpha=0.5, post_ln=post, need_out=False)
        assert none is None and torch.equal(out, plain) and torch.equal(only_ln, out_ln)
        ref = torch.nn.functional.layer_norm(plain.double(), (256,), post.weight.double(), post.bias.double(), post.eps)
    assert (out_ln.double() - ref).abs().max().item() < 3e-6 * (ref.abs().max().item() + 1)
