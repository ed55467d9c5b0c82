"""TTS-side parity against vectors produced by the reference's own modules (tests/golden/make_golden_tts.py)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import dag_oracle as orc


def load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name + ".npz")))


def hifigan_from_golden(g, backend="torch"):
    from daspeech_amd.models import HiFiGANGenerator
    cfg = json.loads(bytes(g["cfg_json"]).decode())
    m = HiFiGANGenerator(cfg, conv_backend=backend)
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w:")}
    m.load_reference_state_dict(sd)
    return m.eval()


def test_oracle_length_regulator_vs_reference(golden_dir):
    g = load(golden_dir, "fastspeech2_pieces")
    out, lens = orc.length_regulate(g["lr_x"], g["lr_dur"])
    np.testing.assert_array_equal(lens, g["lr_lens"])
    np.testing.assert_array_equal(out, g["lr_out"])


def test_oracle_durations_and_bucketize_vs_reference(golden_dir):
    g = load(golden_dir, "fastspeech2_pieces")
    dur = orc.durations(g["va_log_dur"], g["va_pad"], 1.0)
    assert dur.sum(1).tolist() == g["va_out_lens"].tolist()
    idx = orc.bucketize(g["va_pitch"].reshape(-1), g["va_pitch_bins"])
    ref = torch.bucketize(torch.from_numpy(g["va_pitch"].reshape(-1)), torch.from_numpy(g["va_pitch_bins"])).numpy()
    np.testing.assert_array_equal(idx, ref)


def test_hifigan_torch_backend_matches_reference(golden_dir):
    g = load(golden_dir, "hifigan_small")
    m = hifigan_from_golden(g)
    with torch.no_grad():
        wav = m(torch.from_numpy(g["mel"]))
    assert tuple(wav.shape) == g["wav"].shape
    np.testing.assert_allclose(wav.numpy(), g["wav"], rtol=1e-4, atol=1e-6)


def test_hifigan_weight_norm_folding():
    from daspeech_amd.models import HiFiGANGenerator
    v = torch.randn(6, 4, 3); gg = torch.rand(6, 1, 1) + 0.5
    conv = torch.nn.utils.weight_norm(torch.nn.Conv1d(4, 6, 3))
    with torch.no_grad():
        conv.weight_v.copy_(v); conv.weight_g.copy_(gg)
    sd = HiFiGANGenerator.fold_weight_norm({"c.weight_g": gg, "c.weight_v": v, "c.bias": torch.zeros(6)})
    torch.nn.utils.remove_weight_norm(conv)
    torch.testing.assert_close(sd["c.weight"], conv.weight.detach())


def test_positions_from_padding_mask():
    from daspeech_amd.models.fastspeech2 import positions_from_padding_mask, sinusoidal_table
    pm = torch.tensor([[False, False, False, True], [False, True, True, True]])
    assert positions_from_padding_mask(pm).tolist() == [[2, 3, 4, 1], [2, 1, 1, 1]]
    tab = sinusoidal_table(8, 6)
    assert torch.all(tab[1] == 0) and tab.shape == (8, 6)
    assert abs(tab[2, 0].item() - np.sin(2.0)) < 1e-6 and abs(tab[2, 3].item() - np.cos(2.0)) < 1e-6


@pytest.mark.gpu
def test_variance_adaptor_matches_reference_on_gpu(golden_dir):
    """VarianceAdaptor at inference: conv predictors (torch) + HIP durations / bucketize+embed / length regulator."""
    from daspeech_amd.models import VarianceAdaptor
    g = load(golden_dir, "fastspeech2_pieces")
    va = VarianceAdaptor(16, 16, 3, 32, -2.0, 3.0, -1.5, 2.5)
    sd = {k[5:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("va_w:")}
    sd = {k.replace("length_regulator.", ""): v for k, v in sd.items()}
    va.load_state_dict(sd, strict=True)
    va = va.cuda().eval()
    with torch.no_grad():
        y, lens, log_dur, pitch, energy = va(torch.from_numpy(g["va_x"]).cuda(), torch.from_numpy(g["va_pad"]).cuda())
    assert lens.tolist() == g["va_out_lens"].tolist()
    np.testing.assert_allclose(log_dur.cpu().numpy(), g["va_log_dur"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(y.cpu().numpy(), g["va_out"], rtol=1e-4, atol=1e-5)        # north star: <= 1e-4 rel


@pytest.mark.gpu
def test_hifigan_gpu_torch_backend(golden_dir):
    g = load(golden_dir, "hifigan_small")
    m = hifigan_from_golden(g).cuda()
    with torch.no_grad():
        wav = m(torch.from_numpy(g["mel"]).cuda())
    np.testing.assert_allclose(wav.cpu().numpy(), g["wav"], rtol=1e-3, atol=2e-5)


@pytest.mark.gpu
def test_hifigan_hip_matches_torch_fp32():
    """Hand-written MFMA conv stack (fp16 storage, fp32 accumulate) vs the fp32 torch path of the same V1 generator."""
    from daspeech_amd.models import HiFiGANGenerator
    torch.manual_seed(3)
    g = HiFiGANGenerator().cuda().eval()                      # full V1 widths (512 initial channels)
    with torch.no_grad():
        for p in g.parameters():                              # fan-in scaled weights keep activations O(1) through ~50 layers
            if p.dim() > 1:
                fan = p[0].numel() if not isinstance(p, torch.nn.ConvTranspose1d) else p.shape[0] * p.shape[2]
                p.copy_(torch.randn_like(p) / (p.shape[1] * p.shape[2]) ** 0.5)
            else:
                p.copy_(torch.randn_like(p) * 0.05)
    mel = torch.randn(2, 80, 37, device="cuda")
    with torch.no_grad():
        ref = g(mel)
        g.conv_backend = "hip"
        out = g(mel)
    assert out.shape == ref.shape == (2, 1, 37 * 256)
    err = (out - ref).abs().max().item()
    assert err < 2e-2 and torch.isfinite(out).all(), err
    assert (out - ref).abs().mean().item() < 2e-3
    # odd lengths / tile edges
    mel2 = torch.randn(1, 80, 5, device="cuda")
    with torch.no_grad():
        out2 = g(mel2); g.conv_backend = "torch"; ref2 = g(mel2)
    assert (out2 - ref2).abs().max().item() < 2e-2


@pytest.mark.gpu
def test_hifigan_fused_resblock_unit_bit_identical_to_layer_chain():
    """dsp_hifigan_resunit (conv, conv, residual in one launch, intermediate in LDS) against the two dsp_hifigan_conv launches it
    replaces: same rounding points and MFMA step order, so the fp16 outputs must be equal bit for bit — whole generator and
    single units at tile edges (T not a multiple of the 240/496-column tiles, T smaller than a halo)."""
    from daspeech_amd import _lib
    from daspeech_amd.hifigan_ops import HiFiGANHipRunner
    from daspeech_amd.models import HiFiGANGenerator
    torch.manual_seed(11)
    g = HiFiGANGenerator().cuda().eval()
    with torch.no_grad():
        for p in g.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[1] * p.shape[2]) ** 0.5 if p.dim() > 1 else torch.randn_like(p) * 0.05)
    fused, chain = HiFiGANHipRunner(g, fuse_units=True), HiFiGANHipRunner(g, fuse_units=False)
    for B, T in ((2, 37), (1, 5), (3, 64)):
        mel = torch.randn(B, 80, T, device="cuda")
        a, b = fused(mel), chain(mel)
        assert torch.isfinite(a).all() and torch.equal(a, b), (B, T, (a - b).abs().max().item())
    lib = _lib.load()
    st = _lib.current_stream_handle()
    # T * B large enough selects the wide tiles (C=64: 496 columns, C=32: 1008), small T the narrow ones
    for C, K, dil, T in ((32, 11, 5, 1000), (32, 3, 1, 497), (64, 7, 3, 481), (64, 11, 5, 7), (128, 11, 5, 250), (128, 3, 1, 239),
                         (32, 7, 3, 258111), (64, 11, 5, 127003), (64, 3, 1, 126976), (256, 11, 5, 300), (256, 3, 1, 111), (256, 7, 3, 113),
                         (256, 7, 3, 22403)):        # C=256: 48-column tiles for few tiles, 112-column tiles from 384 workgroups on
        assert lib.dsp_hifigan_resunit_supported(C, K, dil)
        x = (torch.randn(2, T, C, device="cuda") * 1.5).half()
        from daspeech_amd.hifigan_ops import pack_weights
        w1 = pack_weights((torch.randn(K, C, C, device="cuda") / (C * K) ** 0.5).half()); w2 = pack_weights((torch.randn(K, C, C, device="cuda") / (C * K) ** 0.5).half())
        b1 = torch.randn(C, device="cuda") * 0.1; b2 = torch.randn(C, device="cuda") * 0.1
        import ctypes
        sh1 = (ctypes.c_int * K)(*[(k - (K - 1) // 2) * dil for k in range(K)]); sh2 = (ctypes.c_int * K)(*[k - (K - 1) // 2 for k in range(K)])
        for accumulate in (0, 1):
            base = (torch.randn(2, T, C, device="cuda")).half()
            o1, o2, h = base.clone(), base.clone(), torch.empty_like(x)
            _lib.check(lib.dsp_hifigan_resunit(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(o1), 2, T, C, K, dil,
                                               0.1, 1.0 / 3, accumulate, st), "resunit")
            _lib.check(lib.dsp_hifigan_conv(_lib.ptr(x), _lib.ptr(w1), _lib.ptr(b1), None, _lib.ptr(h), 2, T, C, C, K, sh1, 0.1, 1.0, 0, 1, 0, T, C, st), "conv")
            _lib.check(lib.dsp_hifigan_conv(_lib.ptr(h), _lib.ptr(w2), _lib.ptr(b2), _lib.ptr(x), _lib.ptr(o2), 2, T, C, C, K, sh2, 0.1, 1.0 / 3,
                                            1 if accumulate else 0, 1, 0, T, C, st), "conv")
            assert torch.equal(o1, o2), (C, K, dil, T, accumulate, (o1.float() - o2.float()).abs().max().item())
    assert not lib.dsp_hifigan_resunit_supported(512, 3, 1) and not lib.dsp_hifigan_resunit_supported(64, 4, 1)
