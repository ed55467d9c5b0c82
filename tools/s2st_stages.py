#!/usr/bin/env python3
"""Host-issue time vs GPU time of the S2ST pipeline's stages (acoustic model up to the mel lengths, vocoder groups): is the acoustic
stage launch bound (then the vocoder of the previous batch on a second stream can fill its gaps)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.models.daspeech import S2SConformerDAGFastSpeech2Model
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
dev = torch.device("cuda:0"); torch.manual_seed(1234)
model = calibrate_synthetic_weights(S2SConformerDAGFastSpeech2Model()).to(dev).eval()
voc = HiFiGANGenerator(conv_backend="hip").to(dev).eval()
gen = S2SNATGenerator(voc, torch.zeros(80, device=dev), torch.ones(80, device=dev), vocoder_group=8)
batches = [make_s2st_batch(32, dev, seed=i) for i in range(2)]
for i in range(4): gen.generate(model, batches[i % 2])
torch.cuda.synchronize()
def acoustic(sample):
    net = sample["net_input"]
    enc = model.forward_encoder(net["src_tokens"], net["src_lengths"])
    prev = model.initialize_output_tokens_by_src(net["src_lengths"], max_src_len=net["src_tokens"].shape[1])
    dec = model.forward_decoder(prev, enc)
    tts_in = model.adaptor(dec["features"])
    mel, out_lens, _, _, _ = model.tts(tts_in, dec["features_padding_mask"])
    return mel, out_lens
iss, tot = [], []
with torch.no_grad():
    for i in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        mel, out_lens = acoustic(batches[i % 2]); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        iss.append(t1 - t0); tot.append(t2 - t0)
print(f"acoustic model (B=32): host issue {1e3 * sorted(iss)[5]:.2f} ms, until GPU done {1e3 * sorted(tot)[5]:.2f} ms")
with torch.no_grad():
    mel = torch.randn(8, 80, 330, device=dev); lens = torch.full((8,), 330, device=dev, dtype=torch.int32)
    for _ in range(3): voc(mel, lengths=lens)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4): voc(mel, lengths=lens)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"vocoder 4 groups of 8 x 330: host issue {1e3 * (t1 - t0):.2f} ms, until GPU done {1e3 * (t2 - t0):.2f} ms")
with torch.no_grad():
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(10): gen.generate(model, batches[i % 2])
    torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"generate(): {1e2 * (t2 - t0):.2f} ms per batch")
