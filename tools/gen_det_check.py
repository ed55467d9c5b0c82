import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from daspeech_amd.generator import S2SNATGenerator
from daspeech_amd.models import HiFiGANGenerator
from daspeech_amd.synthetic import calibrate_synthetic_weights, make_s2st_batch
from tests.test_gpu_model import small_model
m = calibrate_synthetic_weights(small_model().eval())
voc = HiFiGANGenerator(conv_backend="hip").cuda().eval()
gen = S2SNATGenerator(voc, torch.zeros(80), torch.ones(80), vocoder_group=2)
batches = [make_s2st_batch(3, "cuda", seed=20 + i, min_frames=90 + 10 * i, max_frames=150) for i in range(4)]
a = [gen.generate(m, s) for s in batches]; b = [gen.generate(m, s) for s in batches]
torch.cuda.synchronize()
def cmp(x, y, name):
    for bi, (gb, wb) in enumerate(zip(x, y)):
        for ui, (g, w) in enumerate(zip(gb, wb)):
            df = (g["feature"] - w["feature"]).abs().max().item() if g["feature"].shape == w["feature"].shape else "shape"
            dw = (g["waveform"] - w["waveform"]).abs().max().item() if g["waveform"].shape == w["waveform"].shape else "shape"
            print(name, bi, ui, "mel maxdiff", df, "wav maxdiff", dw, "tokens eq", torch.equal(g["tokens"], w["tokens"]))
cmp(a, b, "seq-vs-seq")
c = list(gen.generate_batches(m, batches)); torch.cuda.synchronize()
cmp(c, a, "pipe-vs-seq")
