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


# ======================================================================================================================
# Full-width vectors: the reference modules at the released sizes on seeded weights (make_golden_tts.py full); the weights are
# rebuilt here from the seed (tests/util_inputs.seeded_weights), the fixture holds inputs and reference outputs only.
# ======================================================================================================================
def _noemb_from_seed(g, device):
    from daspeech_amd.models.fastspeech2 import FastSpeech2NoEmb
    from tests.util_inputs import seeded_weights
    m = FastSpeech2NoEmb()
    w = seeded_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"]))
    w["var_adaptor.duration_predictor.proj.bias"] = np.full((1,), float(g["dur_bias"]), np.float32)
    assert sorted(w) == sorted(str(x) for x in g["param_names"])          # same parameter names as the reference module
    m.load_state_dict({k: torch.from_numpy(v) for k, v in w.items()}, strict=True)
    return m.to(device).eval()


def _check_noemb(m, g, device, mel_rtol):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    with torch.no_grad():
        mel, _, out_lens, log_dur, pitch, energy = m(t(g["x"]), t(g["pad"]))
        mel2, _, out_lens2, log_dur2, pitch2, energy2 = m(t(g["x"]), t(g["pad"]), durations=t(g["tf_dur"]), pitches=t(g["tf_pitch_in"]),
                                                       energies=t(g["tf_energy_in"]))
        fft0 = m.encoder_fft_layers[0](t(g["x"]), t(g["pad"]))
    assert out_lens.tolist() == g["inf_out_lens"].tolist() and out_lens2.tolist() == g["tf_out_lens"].tolist()
    scale = float(np.abs(g["inf_mel"]).max())
    # north star: <= 1e-4 relative on mel-spectrogram frames (relative to the mel range)
    for got, want, lens in ((mel, g["inf_mel"], g["inf_out_lens"]), (mel2, g["tf_mel"], g["tf_out_lens"])):
        got = got.cpu().numpy()
        assert got.shape == want.shape
        for b, n in enumerate(lens):                      # frames of each utterance (the padded tail is not part of any output)
            err = np.abs(got[b, :n] - want[b, :n]).max() if n else 0.0
            assert err <= mel_rtol * scale, (b, err, scale)
    pad = g["pad"]
    np.testing.assert_allclose(log_dur.cpu().numpy()[~pad], g["inf_log_dur"][~pad], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pitch.cpu().numpy()[~pad], g["inf_pitch"][~pad], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(energy.cpu().numpy()[~pad], g["inf_energy"][~pad], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(energy2.cpu().numpy()[~pad], g["tf_energy"][~pad], rtol=1e-4, atol=1e-4)
    f = fft0.cpu().numpy()
    np.testing.assert_allclose(f[~pad], g["fft0_out"][~pad], rtol=1e-4, atol=1e-4 * float(np.abs(g["fft0_out"]).max()))


def test_fastspeech2_noemb_torch_path_vs_reference(golden_dir):
    """FastSpeech2NoEmb (FFT layers, variance adaptor, length regulator, output projection) through the torch ops on CPU against
    FastSpeech2EncoderNoEmb.forward of the reference at the released widths: inference and teacher-forced."""
    g = load(golden_dir, "fastspeech2_noemb_seeded")
    _check_noemb(_noemb_from_seed(g, "cpu"), g, "cpu", 1e-4)


def test_variance_adaptor_training_path_is_differentiable(golden_dir):
    """With gradients enabled the adaptor must be differentiable end to end (mel loss -> encoder FFT layers, both embedding tables,
    the energy predictor through x + pitch_emb), as the reference's F.pad/cat length regulator and `x + embed(...)` are."""
    g = load(golden_dir, "fastspeech2_noemb_seeded")
    m = _noemb_from_seed(g, "cpu").train()
    x = torch.from_numpy(g["x"]).requires_grad_()
    mel, _, out_lens, log_dur, pitch, energy = m(x, torch.from_numpy(g["pad"]), durations=torch.from_numpy(g["tf_dur"]),
                                              pitches=torch.from_numpy(g["tf_pitch_in"]), energies=torch.from_numpy(g["tf_energy_in"]))
    mel.abs().sum().backward()                      # the mel L1 term ALONE
    for p in (m.encoder_fft_layers[0].ffn.ffn[0].weight, m.var_adaptor.embed_pitch.weight, m.var_adaptor.embed_energy.weight,
              m.decoder_fft_layers[0].self_attn.q_proj.weight):
        assert p.grad is not None and p.grad.abs().sum() > 0
    assert x.grad is not None and x.grad.abs().sum() > 0


@pytest.mark.gpu
def test_fastspeech2_noemb_hip_path_vs_reference(golden_dir):
    """The same on the GPU in eval / no-grad mode: split-precision MFMA convolutions and GEMMs (dsp_conv1d_split), dsp_layer_norm,
    dsp_durations, dsp_bucketize_embed_add, dsp_length_regulator_* — against the reference module's output."""
    from daspeech_amd import decode_ops
    g = load(golden_dir, "fastspeech2_noemb_seeded")
    m = _noemb_from_seed(g, "cuda")
    old = decode_ops.set_split_gemm(True)
    try:
        _check_noemb(m, g, "cuda", 1e-4)
    finally:
        decode_ops.set_split_gemm(old)


def _hifigan_v1_from_seed(g, backend, device):
    from daspeech_amd.models import HiFiGANGenerator
    from tests.util_inputs import seeded_weights
    m = HiFiGANGenerator(conv_backend=backend)
    w = seeded_weights({k: tuple(v.shape) for k, v in m.state_dict().items()}, int(g["seed"]))
    m.load_reference_state_dict({k: torch.from_numpy(v) for k, v in w.items()})
    return m.to(device).eval()


def test_hifigan_v1_torch_backend_vs_reference_full_width(golden_dir):
    g = load(golden_dir, "hifigan_v1_seeded")
    m = _hifigan_v1_from_seed(g, "torch", "cpu")
    with torch.no_grad():
        wav = m(torch.from_numpy(g["mel"]), lengths=torch.from_numpy(g["lens"]))          # per-utterance, as the reference vocodes
    for b, n in enumerate(g["lens"]):
        np.testing.assert_allclose(wav[b, 0, : n * 256].numpy(), g[f"wav{b}"], rtol=0, atol=2e-5)


# Tolerances of the HIP vocoder against the REFERENCE's fp32 waveform.
#   "hip" (csrc/hifigan_conv_f32.hip): fp32 activations and weights as the reference, operands split on the fp16 matrix cores, every
#       product exact in the fp32 accumulator (2^-22 relative per convolution) — the reference's own arithmetic up to summation order:
#       measured max 1.3e-6 / mean 2.2e-7 (profiles/r03c_hifigan_parity.txt; the torch fp32 path of the same module: 1.5e-6 / 2.4e-7),
#       asserted at 1e-5 / 2e-6 — ten times inside the 1e-4 bar VERDICT r02 set for an fp32-accurate mode.
#   "hip_fp16" (csrc/hifigan_conv.hip): activations in fp16 (11-bit significand) with fp32 accumulation: every one of the ~50 stored
#       layers adds a relative rounding error of 2^-12 rms to O(1) activations, which the following layers carry with gain ~1 (residual
#       units), i.e. ~sqrt(50) * 2.4e-4 ~ 2e-3 rms before conv_post + tanh — a few e-3 max over ~10^4 samples.  Measured
#       (profiles/r02_hifigan_parity.txt): max 1.5e-3, mean 2.6e-4 on a waveform of rms 0.3.  NARROWER arithmetic than the reference:
#       a fast mode, never the number quoted as the reference's precision.  Asserted with 3x headroom.
HIP_WAV_TOL = {"hip": (1e-5, 2e-6), "hip_fp16": (5e-3, 8e-4)}


@pytest.mark.gpu
@pytest.mark.parametrize("backend", ["hip", "hip_fp16"])
def test_hifigan_v1_hip_backend_vs_reference_full_width(golden_dir, backend):
    """The HIP vocoder against the waveform the REFERENCE Generator produced for the same seeded V1 weights — single utterances (the
    reference's own loop) and the padded batch with per-utterance lengths."""
    g = load(golden_dir, "hifigan_v1_seeded")
    m = _hifigan_v1_from_seed(g, backend, "cuda")
    mel, lens = torch.from_numpy(g["mel"]).cuda(), torch.from_numpy(g["lens"]).cuda()
    tol_max, tol_mean = HIP_WAV_TOL[backend]
    with torch.no_grad():
        batch = m(mel, lengths=lens)
        for b, n in enumerate(g["lens"]):
            single = m(mel[b:b + 1, :, :n].contiguous())[0, 0]
            err = (single.cpu().numpy() - g[f"wav{b}"])
            assert np.abs(err).max() < tol_max and np.abs(err).mean() < tol_mean, (backend, b, np.abs(err).max(), np.abs(err).mean())
            # grouped vocoding == vocoding alone on the utterance's own samples, bit for bit (per-layer length masking in the kernels)
            assert torch.equal(batch[b, 0, : n * 256], single), (b, (batch[b, 0, : n * 256] - single).abs().max().item())


@pytest.mark.gpu
def test_hifigan_fp32_mode_grouped_equals_per_utterance_and_tracks_torch_fp32():
    """The fp32 (split-operand) chain on ragged groups that cross tile edges: bit-identical to one-at-a-time vocoding, and within 1e-4 of
    the torch fp32 path of the same module on random weights."""
    from daspeech_amd.hifigan_ops import HiFiGANHipRunner
    from daspeech_amd.models import HiFiGANGenerator
    torch.manual_seed(6)
    gmod = HiFiGANGenerator().cuda().eval()
    with torch.no_grad():
        for p in gmod.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[1] * p.shape[2]) ** 0.5 if p.dim() > 1 else torch.randn_like(p) * 0.05)
    run = HiFiGANHipRunner(gmod, precision="fp32")
    chain = HiFiGANHipRunner(gmod, precision="fp32", fuse_units=False)
    lens = torch.tensor([61, 60, 33, 7, 1], device="cuda")
    mel = torch.randn(5, 80, 61, device="cuda")
    mel = mel.masked_fill(torch.arange(61, device="cuda").view(1, 1, -1) >= lens.view(-1, 1, 1), 0)
    batch = run(mel, lens)
    assert torch.equal(batch, chain(mel, lens))            # fused ResBlock units (intermediate in LDS) == the layer-at-a-time chain, bit for bit
    with torch.no_grad():
        for b, n in enumerate(lens.tolist()):
            single = run(mel[b:b + 1, :, :n].contiguous())[0, 0]
            assert torch.equal(batch[b, 0, : n * 256], single), (b, n)
            ref = gmod(mel[b:b + 1, :, :n].contiguous())[0, 0]
            assert (single - ref).abs().max().item() < 1e-4, (b, (single - ref).abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("fuse", [True, False])
def test_hifigan_grouped_equals_per_utterance(fuse):
    """Length-masked batch vs one-at-a-time on ragged groups that cross tile edges (fused ResBlock units and the layer chain)."""
    from daspeech_amd.hifigan_ops import HiFiGANHipRunner
    from daspeech_amd.models import HiFiGANGenerator
    torch.manual_seed(5)
    gmod = HiFiGANGenerator().cuda().eval()
    with torch.no_grad():
        for p in gmod.parameters():
            p.copy_(torch.randn_like(p) / (p.shape[1] * p.shape[2]) ** 0.5 if p.dim() > 1 else torch.randn_like(p) * 0.05)
    run = HiFiGANHipRunner(gmod, fuse_units=fuse)
    lens = torch.tensor([61, 60, 33, 7, 1], device="cuda")
    mel = torch.randn(5, 80, 61, device="cuda")
    mel = mel.masked_fill(torch.arange(61, device="cuda").view(1, 1, -1) >= lens.view(-1, 1, 1), 0)
    batch = run(mel, lens)
    for b, n in enumerate(lens.tolist()):
        single = run(mel[b:b + 1, :, :n].contiguous())[0, 0]
        assert torch.equal(batch[b, 0, : n * 256], single), (b, n)
