"""Host-side logic added in round 4 (no GPU): the wrappers decline what they do not serve (the callers then keep the torch formulation),
stacked projections fall back to separate ones, position tables are constants of their key."""
import torch

from daspeech_amd import decode_ops
from daspeech_amd.models.daspeech import RelPosSelfAttention, _MHA, _rel_positional_encoding, rel_positional_encoding
from daspeech_amd.models.fastspeech2 import FFTLayer


def test_gpu_only_wrappers_decline_cpu_tensors():
    q = torch.randn(2, 10, 128)
    assert decode_ops.attention(q, q, q, None, 2) is None
    assert decode_ops.relpos_attention(q, q, q, torch.randn(1, 19, 128), torch.zeros(2, 64), torch.zeros(2, 64), None, 2) is None
    l1, l2 = torch.nn.Linear(256, 512).eval(), torch.nn.Linear(512, 256).eval()
    assert decode_ops.ffn_fused(torch.randn(2, 70, 256), None, l1, l2, "relu") is None
    assert decode_ops.valid_lengths(torch.zeros(2, 5, dtype=torch.bool)) is None and decode_ops.valid_lengths(None) is None


def test_linear_fused_falls_back_to_separate_projections():
    torch.manual_seed(0)
    lins = [torch.nn.Linear(32, 16).eval() for _ in range(3)]
    x = torch.randn(2, 7, 32)
    with torch.no_grad():
        got = decode_ops.linear_fused(x, lins)
        for g, l in zip(got, lins):
            assert torch.equal(g, l(x))
    y = decode_ops.linear_fused(x.requires_grad_(), lins)          # under autograd as well
    sum(t.sum() for t in y).backward()
    assert all(l.weight.grad is not None for l in lins)


def test_position_tables_are_cached_constants():
    a = rel_positional_encoding(17, 64, torch.device("cpu"), torch.float32)
    b = rel_positional_encoding(17, 64, torch.device("cpu"), torch.float32)
    assert a is b and torch.equal(a, _rel_positional_encoding(17, 64, torch.device("cpu"), torch.float32))
    assert rel_positional_encoding(18, 64, torch.device("cpu"), torch.float32).shape == (1, 35, 64)
    att = RelPosSelfAttention(128, 2).eval()
    with torch.no_grad():
        t9 = rel_positional_encoding(9, 128, torch.device("cpu"), torch.float32)
        assert att._projected_positions(t9) is not att._projected_positions(t9)           # off by default: computed per forward, as the reference does
        att.cache_position_projection = True
        p1, p2 = att._projected_positions(rel_positional_encoding(9, 128, torch.device("cpu"), torch.float32)), \
            att._projected_positions(rel_positional_encoding(9, 128, torch.device("cpu"), torch.float32))
        assert p1 is p2
        att.linear_pos.weight.mul_(2.0)                              # a new weight version invalidates the projection
        p3 = att._projected_positions(rel_positional_encoding(9, 128, torch.device("cpu"), torch.float32))
    assert p3 is not p1 and torch.allclose(p3, 2 * p1)
    p4 = att._projected_positions(rel_positional_encoding(9, 128, torch.device("cpu"), torch.float32))   # with autograd on: never the cache
    assert p4.requires_grad


def test_eval_mode_attention_blocks_agree_with_their_training_formulation_on_cpu():
    """the eval branches (stacked projections, optional HIP attention) and the training branches of the attention modules are the same math"""
    torch.manual_seed(1)
    x, mem = torch.randn(2, 9, 128), torch.randn(2, 6, 64)
    pad, mpad = torch.zeros(2, 9, dtype=torch.bool), torch.zeros(2, 6, dtype=torch.bool)
    pad[1, 7:] = True; mpad[1, 4:] = True
    for mod, args in ((_MHA(128, 2), (x, x, pad)), (_MHA(128, 2, kdim=64), (x, mem, mpad))):
        with torch.no_grad():
            a = mod.eval()(*args, residual=x)
            b = mod.train()(*args, residual=x)                    # dropout 0: deterministic
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-5)
    f = FFTLayer(128, 2, 256, 9)
    with torch.no_grad():
        torch.testing.assert_close(f.eval()(x, pad), f.train()(x, pad), rtol=1e-5, atol=1e-5)


def test_task_train_step_carries_the_reference_profiler_scopes():
    """tasks/nat_speech_to_speech.py:299,304: criterion under record_function("forward"), optimizer.backward under record_function("backward");
    valid_step in eval mode without a graph.  A stand-in model / criterion: the scopes and the call order are what is checked."""
    import torch
    from torch.profiler import profile, ProfilerActivity
    from daspeech_amd.synthetic import NATSpeechToSpeechTask
    lin = torch.nn.Linear(4, 1)
    seen = {}

    def criterion(model, sample):
        seen["update_num"] = sample.get("update_num")
        seen["training"] = model.training
        loss = model(sample["x"]).sum()
        return loss, 1, {"loss": float(loss)}

    class Opt:
        def backward(self, loss):
            seen["backward"] = True
            loss.backward()
    task = NATSpeechToSpeechTask()
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        loss, n, log = task.train_step({"x": torch.ones(2, 4)}, lin, criterion, Opt(), update_num=7)
    names = {e.name for e in prof.events()}
    assert {"forward", "backward"} <= names
    assert seen == {"update_num": 7, "training": True, "backward": True} and n == 1 and lin.weight.grad is not None
    lin.zero_grad()
    loss0, _, _ = task.train_step({"x": torch.ones(2, 4)}, lin, criterion, torch.optim.SGD(lin.parameters(), lr=0.1), update_num=8, ignore_grad=True)
    assert float(loss0) == 0.0 and float(lin.weight.grad.abs().sum()) == 0.0
    vloss, _, _ = task.valid_step({"x": torch.ones(2, 4)}, lin, criterion)
    assert seen["training"] is False and not vloss.requires_grad
