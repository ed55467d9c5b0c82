"""dsp_attention_split (fp32-accurate matrix-core attention) against an fp64 torch restatement of fairseq's eval-mode
MultiheadAttention core (modules/multihead_attention.py: softmax(q k^T * dk^-0.5 + key_padding_mask) v)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(q, k, v, pad, heads):
    B, N, C = q.shape
    M = k.shape[1]
    dk = C // heads
    qd, kd, vd = (t.double().view(B, -1, heads, dk).transpose(1, 2) for t in (q, k, v))
    s = qd @ kd.transpose(-1, -2) * dk ** -0.5
    if pad is not None:
        s = s.masked_fill(pad.view(B, 1, 1, M), float("-inf"))
    return (torch.softmax(s, -1) @ vd).transpose(1, 2).reshape(B, N, C)


def _lengths_mask(lens, M, dev):
    return torch.arange(M, device=dev)[None, :] >= torch.tensor(lens, device=dev)[:, None]


CASES = [
    # B, N, M, heads, dk, key lengths (None: no mask)
    (3, 77, 77, 4, 64, [77, 40, 5]),
    (2, 130, 45, 8, 64, [45, 31]),
    (2, 33, 200, 2, 128, [200, 129]),
    (4, 354, 354, 8, 64, [354, 300, 211, 97]),
    (2, 500, 500, 2, 128, None),
    (1, 1, 1, 1, 64, None),
    (2, 128, 32, 2, 64, [32, 1]),
]


@pytest.mark.parametrize("B,N,M,H,dk,lens", CASES)
def test_attention_split_matches_fp64(B, N, M, H, dk, lens):
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + N + M)
    C = H * dk
    q = (torch.randn(B, N, C, generator=g) * 1.5).to(dev)
    k = (torch.randn(B, M, C, generator=g) * 1.5).to(dev)
    v = (torch.randn(B, M, C, generator=g) * 2.0 + 0.3).to(dev)
    pad = None if lens is None else _lengths_mask(lens, M, dev)
    with torch.no_grad():
        out = decode_ops.attention(q, k, v, pad, H)
        assert out is not None and out.shape == (B, N, C) and out.is_contiguous()
        ref = _ref(q, k, v, pad, H)
        sd = torch.nn.functional.scaled_dot_product_attention(
            *(t.view(B, -1, H, dk).transpose(1, 2) for t in (q, k, v)),
            attn_mask=None if pad is None else torch.zeros(B, 1, 1, M, device=dev).masked_fill(pad.view(B, 1, 1, M), float("-inf")))
        sd = sd.transpose(1, 2).reshape(B, N, C)
    scale = ref.abs().max().item()
    err = (out.double() - ref).abs().max().item() / scale
    err_sd = (sd.double() - ref).abs().max().item() / scale
    assert err < 3e-6, (err, err_sd)          # fp32 level: torch's own fp32 attention sits at ~1e-6 on these inputs
    assert err < 4 * max(err_sd, 5e-7), (err, err_sd)


def test_attention_split_serves_slices_of_a_fused_projection():
    """q | k | v as column slices of one [B,N,3C] buffer (row stride 3C) give the same bits as contiguous copies."""
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(3)
    B, N, H, dk = 3, 150, 4, 64
    C = H * dk
    qkv = torch.randn(B, N, 3 * C, device=dev)
    pad = _lengths_mask([150, 90, 33], N, dev)
    with torch.no_grad():
        a = decode_ops.attention(qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:], pad, H)
        b = decode_ops.attention(qkv[..., :C].contiguous(), qkv[..., C:2 * C].contiguous(), qkv[..., 2 * C:].contiguous(), pad, H)
    assert a is not None and torch.equal(a, b)


def test_attention_split_all_keys_masked_gives_nan_rows_like_torch():
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(4)
    q = torch.randn(2, 40, 128, device=dev); k = torch.randn(2, 50, 128, device=dev); v = torch.randn(2, 50, 128, device=dev)
    pad = _lengths_mask([50, 0], 50, dev)
    with torch.no_grad():
        out = decode_ops.attention(q, k, v, pad, 2)
    assert torch.isfinite(out[0]).all() and torch.isnan(out[1]).all()


def test_attention_split_rejects_unsupported_head_width():
    from daspeech_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 4, 96, device="cuda:0")
    rc = lib.dsp_attention_split(_lib.ptr(x), 96, _lib.ptr(x), 96, _lib.ptr(x), 96, None, _lib.ptr(x), 1, 4, 4, 1, 96, 0.1, None, 0, None)
    assert rc != 0 and b"head width" in lib.dsp_last_error()


@pytest.mark.parametrize("B,T,H,lens", [(3, 77, 4, [77, 40, 5]), (2, 200, 4, [200, 131]), (1, 5, 2, None), (2, 256, 1, [256, 255]),
                                        (2, 300, 4, [300, 170]), (1, 513, 2, None), (32, 177, 4, None)])
def test_relpos_attention_split_matches_fp64(B, T, H, lens):
    """dsp_relpos_attention (the matrix-core kernel) against an fp64 restatement of espnet's RelPositionMultiHeadedAttention core
    (fairseq/modules/espnet_multihead_attention.py:172-254: matrix_ac + rel_shift(matrix_bd), masked soft-max, value product)."""
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(T)
    C = H * 64
    q, k, v = ((torch.randn(B, T, C, generator=g) * 1.2).to(dev) for _ in range(3))
    pos = torch.randn(2 * T - 1, C, generator=g).to(dev)
    bu, bv = (torch.randn(H, 64, generator=g) * 0.5).to(dev), (torch.randn(H, 64, generator=g) * 0.5).to(dev)
    pad = None if lens is None else _lengths_mask(lens, T, dev)
    with torch.no_grad():
        out = decode_ops.relpos_attention(q, k, v, pos.unsqueeze(0), bu, bv, pad, H)
    assert out is not None
    qd, kd, vd = (t.double().view(B, T, H, 64) for t in (q, k, v))
    pd = pos.double().view(2 * T - 1, H, 64)
    ac = torch.einsum("bihd,bjhd->bhij", qd + bu.double(), kd)
    bd_full = torch.einsum("bihd,rhd->bhir", qd + bv.double(), pd)                      # [B,H,T,2T-1]
    idx = (T - 1) - torch.arange(T, device=dev)[:, None] + torch.arange(T, device=dev)[None, :]
    bd = torch.gather(bd_full, 3, idx.expand(B, H, T, T))
    s = (ac + bd) / 8.0
    if pad is not None:
        s = s.masked_fill(pad.view(B, 1, 1, T), float("-inf"))
    ref = torch.einsum("bhij,bjhd->bihd", torch.softmax(s, -1), vd).reshape(B, T, C)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 3e-6, err


def test_ragged_arguments_skip_padding_tiles_and_keep_valid_rows_bit_identical(monkeypatch):
    """lens / q_lens (include/daspeech_decode.h: dsp_conv1d_split_ragged, dsp_attention_split): rows below lens[b] + slack carry exactly
    the bits of the dense call, skipped tiles come back as zeros."""
    from daspeech_amd import decode_ops
    monkeypatch.setattr(decode_ops.SplitConv1d, "KSPLIT", False)      # the dense call must take the same single-launch form as the ragged one
    dev = torch.device("cuda:0")
    torch.manual_seed(11)
    B, T, C, H = 4, 330, 256, 4
    lens_l = [330, 201, 64, 7]
    lens = torch.tensor(lens_l, dtype=torch.int32, device=dev)
    pad = _lengths_mask(lens_l, T, dev)
    assert torch.equal(decode_ops.valid_lengths(pad), lens)
    x = torch.randn(B, T, C, device=dev)
    lin = torch.nn.Linear(C, 3 * C).to(dev).eval()
    conv = torch.nn.Conv1d(C, 1024, 9, padding=4).to(dev).eval()
    sc = decode_ops.SplitConv1d(conv.weight, conv.bias)
    with torch.no_grad():
        for slack in (0, 8, 40):
            dense, rag = decode_ops.linear(x, lin), decode_ops.linear(x, lin, lens=lens, slack=slack)
            dc, rc = sc(x, relu=True), sc(x, relu=True, lens=lens, slack=slack)
            q, k, v = dense[..., :C], dense[..., C:2 * C], dense[..., 2 * C:]
            da, ra = decode_ops.attention(q, k, v, pad, H), decode_ops.attention(q, k, v, pad, H, q_lens=lens, q_slack=slack)
            for d, r, tile in ((dense, rag, 128), (dc, rc, 128), (da, ra, 32)):
                for b, n in enumerate(lens_l):
                    lim = min(T, n + slack)
                    assert torch.equal(d[b, :lim], r[b, :lim])
                    first_skipped = (lim + tile - 1) // tile * tile                 # the first tile that starts at or after the bound
                    assert (r[b, first_skipped:] == 0).all()
        assert (rag[3, 128:] == 0).all() and (ra[3, 64:] == 0).all()                # something was skipped at all


@pytest.mark.parametrize("B,T,Cin,Cout,K", [(32, 61, 1024, 256, 9), (3, 40, 2048, 512, 1), (8, 17, 512, 256, 9), (2, 130, 1024, 128, 3)])
def test_split_k_form_of_short_sequence_layers_matches_fp64(B, T, Cin, Cout, K, monkeypatch):
    """dsp_conv1d_split_ksplit (slices x tap groups over workgroups + fixed-order reduction) against fp64 and the single-launch form."""
    from daspeech_amd import decode_ops
    dev = torch.device("cuda:0")
    torch.manual_seed(K + T)
    conv = torch.nn.Conv1d(Cin, Cout, K, padding=(K - 1) // 2).to(dev)
    sc = decode_ops.SplitConv1d(conv.weight, conv.bias)
    assert sc._tap_groups(B, T) > 0
    x, res = torch.randn(B, T, Cin, device=dev), torch.randn(B, T, Cout, device=dev)
    with torch.no_grad():
        got = sc(x, act="relu", residual=res, alpha=0.5)
        monkeypatch.setattr(decode_ops.SplitConv1d, "KSPLIT", False)
        one = sc(x, act="relu", residual=res, alpha=0.5)
        ref = res.double() + 0.5 * torch.relu(torch.nn.functional.conv1d(x.double().transpose(1, 2), conv.weight.double(), conv.bias.double(),
                                                                          padding=(K - 1) // 2).transpose(1, 2))
    scale = ref.abs().max().item()
    e_k, e_1 = (got.double() - ref).abs().max().item() / scale, (one.double() - ref).abs().max().item() / scale
    assert e_k < 3e-6 and e_k < 3 * max(e_1, 3e-7), (e_k, e_1)


def test_ragged_acoustic_stage_equals_the_dense_one_on_every_valid_frame():
    """The whole acoustic stage (Conformer -> NAT decoder -> graph decode -> FastSpeech2) with the tile-skipping switches on and off:
    identical tokens, identical mel bits on every valid frame (padding rows only feed padding)."""
    from daspeech_amd.generator import S2SNATGenerator
    from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
    from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
    dev = torch.device("cuda:0")
    torch.manual_seed(1234)
    model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
    gen = S2SNATGenerator(None, torch.zeros(80, device=dev), torch.ones(80, device=dev))
    batch = make_s2st_batch(6, dev, seed=3, min_frames=120, max_frames=700)
    with torch.no_grad():
        model.decoder.ragged = model.tts.ragged = True
        a = gen._acoustic(model, batch)
        model.decoder.ragged = model.tts.ragged = False
        b = gen._acoustic(model, batch)
    assert torch.equal(a["tokens"], b["tokens"]) and torch.equal(a["out_lens"], b["out_lens"])
    assert int(a["out_lens"].min()) < int(a["out_lens"].max())                 # the batch IS ragged
    for i, n in enumerate(a["out_lens"].tolist()):
        assert torch.equal(a["mel"][i, :n], b["mel"][i, :n]), i
