"""Criteria of the DASpeech hot path on top of the HIP ops — caller contract of SURVEY.md §8 a10/a11.

  nat_dag_loss                 DASpeech/criterions/nat_dag_loss.py:45-366     (_compute_dag_loss :114-156, glat_function :202-264)
  s2s_dag_fastspeech2_loss     DASpeech/criterions/s2s_dag_fastspeech2_loss.py:26-370 (_compute_dag_loss_with_alpha_beta :53-91,
                               expect strategy :252-265, TTS losses :275-298)
"""
from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from . import custom_ops, decode_ops


def compute_dag_loss(outputs: Tensor, output_masks: Tensor, targets: Tensor, target_masks: Tensor, links: Tensor,
                     glat_keep_mask: Tensor = None, matchmask: Tensor = None, with_alpha_beta: bool = False):
    """loss = -(dag_loss / T_b).mean(), non-finite samples zeroed and counted (nat_dag_loss.py:114-156)."""
    B, L, _ = outputs.shape
    out_len = output_masks.sum(-1)
    tgt_len = target_masks.sum(-1)
    # `outputs` is never read again (the reference forbids it, dag_loss.py:249-251): keep the gather's backward state as two
    # floats per row instead of an in-place softmax — the forward's B*L*V store disappears, match and gradient are unchanged
    prev_mode = custom_ops.set_lazy_softmax(True)
    try:
        _, match = custom_ops.dag_logsoftmax_gather_inplace(outputs, targets.unsqueeze(1).expand(-1, L, -1))
    finally:
        custom_ops.set_lazy_softmax(prev_mode)
    match = match.transpose(1, 2)                                                     # [B,T,L], already contiguous
    if glat_keep_mask is not None and matchmask is not None:                          # force-emit mask (:130-132)
        gl = glat_keep_mask.unsqueeze(1)                                              # [B,1,L] glanced vertices
        match = match.masked_fill(gl, 0) + match.masked_fill(~matchmask, float("-inf")).masked_fill(~gl, 0).detach()
    if with_alpha_beta:
        loss_b, (alpha, beta) = custom_ops.dag_loss_with_alpha_beta(match, links, out_len, tgt_len)
    else:
        loss_b, alpha, beta = custom_ops.dag_loss(match, links, out_len, tgt_len), None, None
    bad = ~torch.isfinite(loss_b)
    loss_b = loss_b.masked_fill(bad, 0)
    loss = -(loss_b / tgt_len).mean()
    return {"loss": loss, "invalid": bad.sum(), "alpha": alpha, "beta": beta, "match": match, "out_len": out_len, "tgt_len": tgt_len}


@torch.no_grad()
def glat_function(model, logits: Tensor, links: Tensor, prev_output_tokens: Tensor, tgt_tokens: Tensor, glat: Dict):
    """Glancing with the Viterbi alignment, "number-random" strategy (nat_dag_loss.py:202-264)."""
    B, L, _ = logits.shape
    pad = model.pad
    tgt_len = tgt_tokens.ne(pad).sum(-1)
    out_len = prev_output_tokens.ne(pad).sum(-1)
    # (no gradient here: the HIP operator then leaves the logits untouched, so the reference's defensive .clone() — a full
    # B*L*V copy — is not needed)
    _, match = custom_ops.dag_logsoftmax_gather_inplace(logits, tgt_tokens.unsqueeze(1).expand(-1, L, -1))
    match = match.transpose(1, 2)
    path = custom_ops.dag_best_alignment(match, links, out_len, tgt_len)              # [B,L], -1 off-path
    predict_align_mask = path >= 0
    matchmask = torch.zeros(B, tgt_tokens.shape[1] + 1, L, device=logits.device, dtype=torch.bool) \
        .scatter_(1, path.unsqueeze(1) + 1, 1)[:, 1:]                                 # (:225)
    oracle = tgt_tokens.gather(-1, path.clip(min=0))
    same = ((logits.argmax(-1) == oracle) & predict_align_mask).sum(1)
    keep_prob = ((tgt_len - same) / tgt_len.clamp(min=1) * glat["context_p"]).unsqueeze(-1) * predict_align_mask.float()
    keep_mask = (torch.rand_like(keep_prob) < keep_prob) & predict_align_mask
    glat_prev = prev_output_tokens.masked_fill(keep_mask, 0) + oracle.masked_fill(~keep_mask, 0)
    return glat_prev, tgt_tokens, {"glat_keep": keep_mask, "matchmask": matchmask,
                                   "glat_acc": (same.sum() / tgt_len.sum().clamp(min=1)), "path": path}


def s2s_dag_fastspeech2_loss(model, sample: Dict[str, Tensor], glat_p: float = 0.1, tts_loss_weight: float = 5.0):
    """One training objective evaluation: DAG loss + 5.0 x FastSpeech2 losses with the "expect" TTS input
    (s2s_dag_fastspeech2_loss.py:93-306)."""
    net = sample["net_input"]
    tgt = sample["target_text"]
    prev = model.initialize_output_tokens_by_src(net["src_lengths"])
    glat_state = {}

    def _glat(m, logits, links, p, t, g):
        out = glat_function(m, logits, links, p, t, g)
        glat_state.update(out[2])
        return out
    out = model(net["src_tokens"], net["src_lengths"], prev, tgt, glat={"context_p": glat_p}, glat_function=_glat)
    logits, links, feats = out["word_ins"]["out"], out["links"], out["word_ins"]["features"]
    prev = out["prev_output_tokens"]
    dag = compute_dag_loss(logits, prev.ne(model.pad), tgt, tgt.ne(model.pad), links, glat_state.get("glat_keep"),
                           glat_state.get("matchmask"), with_alpha_beta=True)
    # expect strategy: z_i = sum_j P(a_i = j | x, y) v_j   (:252-265)
    expect = decode_ops.posterior(dag["alpha"], dag["beta"]).to(feats.dtype)
    tts_in = model.adaptor(torch.matmul(expect, feats)[:, 1:, :])
    tlen = sample["target_text_lengths"] - 1
    pmask = torch.arange(tts_in.shape[1], device=tts_in.device).unsqueeze(0) >= tlen.unsqueeze(1)
    mel, out_lens, log_dur, pitch, energy = model.tts(tts_in, pmask, durations=sample["durations"], pitches=sample["pitches"],
                                                      energies=sample["energies"])
    # TTS losses (:275-298): L1 on mel frames, MSE on log-duration / pitch / energy over non-pad phonemes
    tgt_mel, tgt_mel_len = sample["target_audio"], sample["target_audio_lengths"]
    F_ = min(mel.shape[1], tgt_mel.shape[1])
    fmask = (torch.arange(F_, device=mel.device).unsqueeze(0) < tgt_mel_len.unsqueeze(1)).unsqueeze(-1)
    l1 = (F.l1_loss(mel[:, :F_], tgt_mel[:, :F_], reduction="none") * fmask).sum() / fmask.sum().clamp(min=1) / mel.shape[-1]
    nonpad = ~pmask
    log_dur_tgt = torch.log(sample["durations"].float() + 1)
    dur_l = F.mse_loss(log_dur[nonpad], log_dur_tgt[nonpad])
    pit_l = F.mse_loss(pitch[nonpad], sample["pitches"][nonpad])
    ene_l = F.mse_loss(energy[nonpad], sample["energies"][nonpad])
    tts = l1 + dur_l + pit_l + ene_l
    loss = dag["loss"] + tts_loss_weight * tts
    return loss, {"loss": loss.detach(), "dag": dag["loss"].detach(), "tts": tts.detach(), "l1": l1.detach(), "dur": dur_l.detach(),
                  "pitch": pit_l.detach(), "energy": ene_l.detach(), "invalid": dag["invalid"], "glat_acc": glat_state.get("glat_acc")}
