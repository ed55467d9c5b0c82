"""A short slice of every randomised sweep in tools/fuzz_*.py as part of the GPU suite (the long runs — ~4 000 cases in r02 — are
launched by hand; each tool prints `<n> cases, <k> failures`)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool,args", [
    ("fuzz_dag.py", ["40", "7"]),                    # banded windows (TR <= 32): strip4g / maxstrip / exp-space K5
    ("fuzz_dag.py", ["30", "8", "mid"]),             # TR 33 .. 64: generic row kernels, tiled K5
    ("fuzz_dag.py", ["40", "9", "dense"]),           # dense windows: matrix-core DP, block-product K5, max-plus alignment, stand-by path
    ("fuzz_lsg.py", ["40", "3"]),                    # K1 forward / softmax / backward
    ("fuzz_decode.py", ["40", "5"]),                 # viterbi / jointviterbi / lookahead / greedy graph decode
    ("fuzz_links.py", ["40", "2"]),                  # fused extract_links vs the numpy oracle
    ("fuzz_layers.py", ["30", "6"]),                 # split-precision conv / GEMM tiles, layer norm, depthwise conv + BN + SiLU
    ("fuzz_tts_glue.py", ["40", "4"]),               # length regulator, durations, bucketize + embed, posterior / expected features
    ("fuzz_attention.py", ["30", "1"]),              # matrix-core attention / relative-position attention / feed-forward module / ragged tiles
])
def test_randomised_sweep(tool, args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    tail = [l for l in r.stdout.splitlines() if "failures" in l or l.startswith("FAIL")]
    assert r.returncode == 0, r.stdout[-2000:]
    assert tail and tail[-1].endswith(", 0 failures"), "\n".join(tail[-10:])
