"""Host-side (PyTorch-ROCm) modules of the DASpeech hot path that sit on top of the HIP ops: FastSpeech2-NoEmb blocks with the
HIP variance-adaptor glue / length regulator, the FFN adapter, and the HiFi-GAN V1 generator."""
from .fastspeech2 import FFNAdapter, FFTLayer, FastSpeech2NoEmb, VarianceAdaptor, VariancePredictor  # noqa: F401
from .hifigan import HiFiGANGenerator, HIFIGAN_V1  # noqa: F401
