"""Criteria of the DASpeech hot path on top of the HIP ops — caller contract of SURVEY.md §8 a10/a11, with the reference's
criterion interface `forward(model, sample, reduce=True) -> (loss, sample_size, logging_output)`.

  NATDAGLoss                   DASpeech/criterions/nat_dag_loss.py:45-366     (_compute_dag_loss :114-156, set_update_num :161-162,
                               forward :164-300, glat_function :202-264)
  S2SDAGFastSpeech2Loss        DASpeech/criterions/s2s_dag_fastspeech2_loss.py:26-370 (_compute_dag_loss_with_alpha_beta :53-91,
                               argmax strategy :213-251, expect strategy :252-265, TTS losses :267-298)
  parse_anneal_argument / get_anneal_value      DASpeech/criterions/utilities.py:17-37

fairseq is not a dependency: `cfg` is any attribute bag carrying the reference's argument names (`glat_p`, `glance_strategy`,
`no_force_emit`, `torch_dag_loss`, `torch_dag_best_alignment`, `torch_dag_logsoftmax_gather`, `training_strategy`,
`tts_loss_weight`, `dag_freezing_steps`); `task` only supplies `tgt_dict.pad()` and may be None (pad = model.pad).
"""
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn.functional as F
from torch import Tensor

from . import custom_ops, decode_ops


# ------------------------------------------------------------------------------------------------ glat-p annealing (utilities.py)
def parse_anneal_argument(anneal_str: str):
    """"0.5:0.1@200k" -> [(0.5, 0.0), (0.1, 200000.0)]   (utilities.py:17-29)."""
    res = []
    for value_str in str(anneal_str).split(":"):
        value, pos = value_str.split("@") if "@" in value_str else (value_str, "0")
        res.append((float(value), float(pos.replace("k", "000"))))
    return res


def get_anneal_value(anneal_params, update_num):
    """Piecewise-linear schedule, including the reference's `+ 1` in the slope denominator (utilities.py:31-37)."""
    last_value, last_pos = anneal_params[0][0], 0
    for value, pos in anneal_params:
        if update_num < pos:
            return last_value + (value - last_value) * (update_num - last_pos) / (pos - last_pos + 1)
        last_value, last_pos = value, pos
    return anneal_params[-1][0]


# ------------------------------------------------------------------------------------------------ glancing (nat_dag_loss.py:202-264)
GLANCE_STRATEGIES = (None, "number-random", "cmlm")


def _viterbi_path(model, logits: Tensor, tgt_tokens: Tensor, links: Tensor, output_length: Tensor, target_length: Tensor,
                  torch_gather: bool, torch_align: bool) -> Tensor:
    """[B, L] target index per vertex on the best alignment, -1 off it.  The two operator families are selected independently, as the
    reference's --torch-dag-logsoftmax-gather / --torch-dag-best-alignment flags are (nat_dag_loss.py:210-222)."""
    idx = tgt_tokens.unsqueeze(1).expand(-1, links.shape[1], -1)
    gather = custom_ops.torch_dag_logsoftmax_gather_inplace if torch_gather else custom_ops.dag_logsoftmax_gather_inplace
    # (no gradient here: the HIP operator then leaves the logits untouched, so a defensive clone of B*L*V is not needed)
    match = gather(logits, idx)[1].transpose(1, 2)
    if torch_align:
        return custom_ops.torch_dag_best_alignment(match.detach().clone(), decode_ops.restore_valid_links(links), output_length, target_length)
    return custom_ops.dag_best_alignment(match, links, output_length, target_length)


def _top_scored(scores: Tensor, counts: Tensor) -> Tensor:
    """1.0 where a position's score reaches its row's `counts[b]`-th largest score (nothing for a zero count) — the reference's
    threshold form (nat_dag_loss.py:236-239): ties with the threshold are all kept, and a count beyond the number of aligned vertices
    (a sample without a valid alignment: every score is the -100 fill) keeps every position."""
    thresh = scores.sort(descending=True)[0].gather(-1, (counts - 1).clip(min=0).unsqueeze(-1)).squeeze(-1)
    thresh = thresh.masked_fill(counts == 0, 100)
    return (scores >= thresh.unsqueeze(-1)).to(scores.dtype)


@torch.no_grad()
def glat_function(model, word_ins_out: Tensor, tgt_tokens: Tensor, prev_output_tokens: Tensor, glat: Dict, links: Tensor = None,
                  glance_strategy: Optional[str] = None, torch_ops: bool = False, noise: Tensor = None, unif: Tensor = None,
                  torch_gather: Optional[bool] = None, torch_align: Optional[bool] = None, unif_n: Tensor = None):
    """Glancing with the Viterbi alignment.  Same positional signature and return value as the reference's closure
    (`glat_function(model, word_ins_out, tgt_tokens, prev_output_tokens, glat, links=links)` -> (glat_prev_output_tokens,
    glat_tgt_tokens, glat_info)).

    Which aligned vertices are revealed (`glance_strategy`):
      None             every aligned vertex independently with probability (T - same) / T * p                     (:229-230)
      "number-random"  exactly round((T - same) * p) of them, the ones with the largest random scores — the released recipe
                       (README.md:240,305)                                                                        (:232-239)
      "cmlm"           round(T * U) of them, U ~ uniform per sentence                                             (:241-248)
    `noise` (the scores), `unif_n` (cmlm's U) and `unif` (the final per-position draw, :251) replay the random draws in the order the
    reference takes them; each is drawn on the device when None.  `torch_gather` / `torch_align` select the torch_* DAG ops
    independently (`torch_ops` sets both)."""
    if glance_strategy not in GLANCE_STRATEGIES:
        raise ValueError(f"glance strategy {glance_strategy!r} (supported: {GLANCE_STRATEGIES})")
    torch_gather = torch_ops if torch_gather is None else torch_gather
    torch_align = torch_ops if torch_align is None else torch_align
    B, L, _ = links.shape
    T = tgt_tokens.shape[1]
    dev = tgt_tokens.device
    n_tgt = tgt_tokens.ne(model.pad).sum(1)
    n_out = prev_output_tokens.ne(model.pad).sum(1)
    guess = word_ins_out.argmax(-1)
    path = _viterbi_path(model, word_ins_out, tgt_tokens, links, n_out, n_tgt, torch_gather, torch_align)
    on_path = path >= 0
    oracle = tgt_tokens.gather(-1, path.clip(min=0))                       # the token each vertex is aligned to (pad-free on the path)
    n_right = ((guess == oracle) & on_path).sum(1)
    # vertex j emits target path[j]: the [B, T, L] emission mask, built through a scratch row for the off-path -1
    matchmask = torch.zeros(B, T + 1, L, device=dev, dtype=torch.bool).scatter_(1, path.unsqueeze(1) + 1, 1)[:, 1:]
    if glance_strategy is None:
        keep_prob = ((n_tgt - n_right) / n_tgt * glat["context_p"]).unsqueeze(-1) * on_path.float()
    else:
        scores = torch.randn(oracle.shape, device=dev, dtype=torch.float) if noise is None else noise.to(dev, torch.float).clone()
        scores.masked_fill_(~on_path, -100)
        if glance_strategy == "number-random":
            counts = ((n_tgt - n_right) * glat["context_p"] + 0.5).to(torch.long)
        else:
            draw = torch.rand_like(n_tgt, dtype=torch.float) if unif_n is None else unif_n.to(dev, torch.float)
            counts = (n_tgt * draw + 0.5).to(torch.long)
        keep_prob = _top_scored(scores, counts)
    u = torch.rand(prev_output_tokens.shape, device=prev_output_tokens.device) if unif is None else unif.to(prev_output_tokens.device)
    revealed = u < keep_prob
    glanced = torch.where(revealed, oracle, prev_output_tokens)
    glat_info = {
        "glat_accu": (n_right.sum() / n_tgt.sum()).detach(),
        "glat_context_p": glat["context_p"],
        "glat_keep": keep_prob.mean().detach(),
        "matchmask": matchmask,
        "keep_word_mask": revealed,
        "glat_prev_output_tokens": glanced,
        # extras (not in the reference's dict): the RNG-free intermediates the parity tests compare
        "path": path, "oracle": oracle, "same_num": n_right,
    }
    return glanced, tgt_tokens, glat_info


DEFAULT_CFG = dict(label_smoothing=0, glat_p="0", glance_strategy=None, no_force_emit=False, torch_dag_logsoftmax_gather=False,
                   torch_dag_best_alignment=False, torch_dag_loss=False, training_strategy="expect", tts_loss_weight=1.0,
                   dag_freezing_steps=-1,
                   # not a reference flag.  With --training-strategy argmax the reference's GPU path takes the Viterbi alignment on the
                   # logits buffer AFTER dag_logsoftmax_gather_inplace has overwritten it with its soft-max (s2s_dag_fastspeech2_loss.py:215
                   # reads outputs["word_ins"]["out"], which :63 mutated), i.e. on log_softmax(softmax(x)); its torch gather leaves the logits
                   # alone.  False (default) reproduces whichever the --torch-dag-logsoftmax-gather flag selects in the reference; True
                   # always aligns on the logits.
                   argmax_on_logits=False)


class NATDAGLoss:
    """registered name: nat_dag_loss (nat_dag_loss.py:45)."""

    def __init__(self, cfg=None, task=None, **overrides):
        base = dict(DEFAULT_CFG)
        if cfg is not None:
            base.update(vars(cfg) if not isinstance(cfg, dict) else cfg)
        base.update(overrides)
        self.cfg = SimpleNamespace(**base)
        self.task = task
        assert self.cfg.label_smoothing == 0, "DAG does not support label smoothing"
        self.glance_strategy = self.cfg.glance_strategy
        self._glat_p_anneal_params = parse_anneal_argument(self.cfg.glat_p)
        self.training = True
        self.set_update_num(0)

    def train(self, mode: bool = True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def set_update_num(self, update_num):
        self.glat_p = get_anneal_value(self._glat_p_anneal_params, update_num)

    def _pad(self, model):
        return self.task.tgt_dict.pad() if self.task is not None and hasattr(self.task, "tgt_dict") else model.pad

    # ---- nat_dag_loss.py:114-156 / s2s_dag_fastspeech2_loss.py:53-91
    def _dag_loss_core(self, outputs, output_masks, targets, target_masks, links, name, factor, matchmask, keep_word_mask, model,
                       with_alpha_beta: bool):
        prelen = outputs.shape[1]
        output_length = output_masks.sum(dim=-1)
        target_length = target_masks.sum(dim=-1)
        idx = targets.unsqueeze(1).expand(-1, prelen, -1)
        if self.cfg.torch_dag_logsoftmax_gather:
            outputs, match_all = custom_ops.torch_dag_logsoftmax_gather_inplace(outputs, idx)
        else:
            # `outputs` is never read again (the reference forbids it, dag_loss.py:249-251): keep the gather's backward state as
            # two floats per row instead of an in-place softmax — the forward's B*L*V store disappears, match and gradient unchanged
            prev_mode = custom_ops.set_lazy_softmax(self._lazy_softmax())
            try:
                outputs, match_all = custom_ops.dag_logsoftmax_gather_inplace(outputs, idx)
            finally:
                custom_ops.set_lazy_softmax(prev_mode)
        match_all = match_all.transpose(1, 2)                                                   # [B,T,L], already contiguous
        if matchmask is not None and not self.cfg.no_force_emit:                                # force-emit (:130-132)
            glat_prev_mask = keep_word_mask.unsqueeze(1)
            match_all = match_all.masked_fill(glat_prev_mask, 0) + \
                match_all.masked_fill(~matchmask, float("-inf")).masked_fill(~glat_prev_mask, 0).detach()
        nvalidtokens = output_masks.sum()
        alpha = beta = None
        if with_alpha_beta:
            assert not self.cfg.torch_dag_loss, "must use the HIP dag loss to obtain alpha and beta"
            loss_result, (alpha, beta) = custom_ops.dag_loss_with_alpha_beta(match_all, links, output_length, target_length)
        elif self.cfg.torch_dag_loss:
            loss_result = custom_ops.torch_dag_loss(match_all, decode_ops.restore_valid_links(links), output_length, target_length)
        else:
            loss_result = custom_ops.dag_loss(match_all, links, output_length, target_length)
        invalid_masks = loss_result.isinf().logical_or(loss_result.isnan())
        loss_result = loss_result.masked_fill(invalid_masks, 0)
        invalid_nsentences = invalid_masks.sum().detach()
        loss = -(loss_result / target_length).mean()
        nll_loss = loss.detach()
        nsentences, ntokens = targets.shape[0], targets.ne(self._pad(model)).sum()
        res = {"name": name, "loss": loss * factor, "nll_loss": nll_loss, "factor": factor, "ntokens": ntokens,
               "nvalidtokens": nvalidtokens, "nsentences": nsentences, "loss_nofactor": loss, "invalid_nsentences": invalid_nsentences}
        return res, alpha, beta

    def _lazy_softmax(self) -> bool:
        return True

    def _compute_dag_loss(self, outputs, output_masks, targets, target_masks, links, label_smoothing=0.0, name="loss", factor=1.0,
                          matchmask=None, keep_word_mask=None, model=None):
        return self._dag_loss_core(outputs, output_masks, targets, target_masks, links, name, factor, matchmask, keep_word_mask, model, False)[0]

    def _glat_args(self):
        if self.glat_p == 0:
            return None
        return {"context_p": max(self.glat_p, 0), "require_glance_grad": False}

    # `glat_draws` = {"noise": ..., "unif": ..., "unif_n": ...} replays the glancing's random draws (parity tests against the reference's
    # recorded draws); None = drawn on the device
    glat_draws = None

    def _glat_function(self):
        def fn(model, word_ins_out, tgt_tokens, prev_output_tokens, glat, links=None):
            return glat_function(model, word_ins_out, tgt_tokens, prev_output_tokens, glat, links=links,
                                 glance_strategy=self.glance_strategy, torch_gather=bool(self.cfg.torch_dag_logsoftmax_gather),
                                 torch_align=bool(self.cfg.torch_dag_best_alignment), **(self.glat_draws or {}))
        return fn

    # ---- nat_dag_loss.py:164-300
    def forward(self, model, sample, reduce=True):
        src_tokens, src_lengths = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
        tgt_tokens = sample["target"]
        if sample.get("update_num", None) is not None:           # in training
            self.set_update_num(sample["update_num"])
        prev_output_tokens = model.initialize_output_tokens_by_tokens(src_tokens, src_lengths)
        outputs = model(src_tokens, src_lengths, prev_output_tokens, tgt_tokens, self._glat_args(), self._glat_function())
        _losses = self._compute_dag_loss(
            outputs["word_ins"].get("out"), prev_output_tokens.ne(self._pad(model)), outputs["word_ins"].get("tgt"),
            outputs["word_ins"].get("mask", None), outputs["links"], name="dag-loss", factor=1,
            matchmask=outputs.get("matchmask", None), keep_word_mask=outputs.get("keep_word_mask", None), model=model)
        loss = _losses["loss"]
        sample_size = 1
        logging_output = {
            "loss": loss.data, "dag_nll-loss": _losses["nll_loss"].data, "ntokens": _losses["ntokens"],
            "nvalidtokens": _losses["nvalidtokens"], "nsentences": _losses["nsentences"],
            "invalid_nsentences": _losses["invalid_nsentences"], "sample_size": sample_size,
            "glat_acc": outputs.get("glat_accu", 0), "glat_keep": outputs.get("glat_keep", 0),
            "dag-loss": _losses["loss_nofactor"].item() if reduce else _losses["loss_nofactor"],
        }
        return loss, sample_size, logging_output

    __call__ = forward

    @staticmethod
    def logging_outputs_can_be_summed() -> bool:
        return True


def _lengths_to_mask(lens: Tensor, n: int = None) -> Tensor:
    n = int(lens.max()) if n is None else n
    return torch.arange(n, device=lens.device).unsqueeze(0) < lens.unsqueeze(1)


class S2SDAGFastSpeech2Loss(NATDAGLoss):
    """registered name: s2s_dag_fastspeech2_loss (s2s_dag_fastspeech2_loss.py:26)."""

    def _lazy_softmax(self) -> bool:
        # argmax strategy, reference GPU behaviour: the gather must really overwrite the logits with their soft-max, because the
        # alignment below reads that buffer (see DEFAULT_CFG["argmax_on_logits"]); everywhere else the buffer is dead after the gather
        return not (self.cfg.training_strategy == "argmax" and not self.cfg.argmax_on_logits)

    def _compute_dag_loss_with_alpha_beta(self, outputs, output_masks, targets, target_masks, links, label_smoothing=0.0, name="loss",
                                          factor=1.0, matchmask=None, keep_word_mask=None, model=None):
        return self._dag_loss_core(outputs, output_masks, targets, target_masks, links, name, factor, matchmask, keep_word_mask, model, True)

    def forward(self, model, sample, reduce=True):
        src_tokens, src_lengths = sample["net_input"]["src_tokens"], sample["net_input"]["src_lengths"]
        tgt_tokens = sample["target_text"]
        if sample.get("update_num", None) is not None:
            self.set_update_num(sample["update_num"])
        prev_output_tokens = model.initialize_output_tokens_by_tokens(src_tokens, src_lengths)
        upd = sample.get("update_num")
        train_dag = self.training and (upd if upd is not None else 0) > self.cfg.dag_freezing_steps          # (:191)
        with torch.set_grad_enabled(train_dag and torch.is_grad_enabled()):
            outputs = model(src_tokens, src_lengths, prev_output_tokens, tgt_tokens, self._glat_args(), self._glat_function())
        dag_loss, alpha, beta = self._compute_dag_loss_with_alpha_beta(
            outputs["word_ins"].get("out"), prev_output_tokens.ne(self._pad(model)), outputs["word_ins"].get("tgt"),
            outputs["word_ins"].get("mask", None), outputs["links"], name="dag-loss", factor=1,
            matchmask=outputs.get("matchmask", None), keep_word_mask=outputs.get("keep_word_mask", None), model=model)
        features = outputs["word_ins"]["features"]                                                           # B x L x D
        if self.cfg.training_strategy == "argmax":
            # z_i = v_{a*_i}, a* = the Viterbi alignment of (y, x)   (:213-251)
            with torch.no_grad():
                links_d = outputs["links"].detach()
                prelen = links_d.shape[1]
                target_length = tgt_tokens.ne(model.pad).sum(1)
                output_length = prev_output_tokens.ne(model.pad).sum(1)
                idx = tgt_tokens.unsqueeze(1).expand(-1, prelen, -1)
                # `out` is the buffer the loss above ran its gather on: soft-max values where the reference's GPU gather would have left
                # them (HIP gather, gradient required, argmax_on_logits off), the logits otherwise — see DEFAULT_CFG["argmax_on_logits"]
                logits_d = outputs["word_ins"]["out"].detach()
                gather = custom_ops.torch_dag_logsoftmax_gather_inplace if self.cfg.torch_dag_logsoftmax_gather \
                    else custom_ops.dag_logsoftmax_gather_inplace                                            # (:219-222; no gradient: no mutation)
                match = gather(logits_d, idx)[1].transpose(1, 2)
                if self.cfg.torch_dag_best_alignment:                                                        # (:226-232)
                    path = custom_ops.torch_dag_best_alignment(match.clone(), decode_ops.restore_valid_links(links_d), output_length, target_length)
                else:
                    path = custom_ops.dag_best_alignment(match, links_d, output_length, target_length)
                path = path.clone()
                path[:, 0] = -1                                                                              # mask <bos>  (:241)
                features_mask = path >= 0
                n_on = features_mask.sum(-1)
                order = torch.argsort((~features_mask).to(torch.int8), dim=1, stable=True)[:, : int(n_on.max())]
                features_padding_mask = ~_lengths_to_mask(n_on, order.shape[1])
            features_on_path = features.gather(1, order.unsqueeze(-1).expand(-1, -1, features.shape[-1])) \
                .masked_fill(features_padding_mask.unsqueeze(-1), 0)                                         # _collate_frames (:244-246)
            input_to_tts = model.adaptor(features_on_path)
        else:
            # expect: z_i = sum_j P(a_i = j | x, y) v_j   (:252-265)
            # (fused: the [B,T,L] posterior never exists; gradient to the features through dsp_posterior_features_bwd)
            # (without a gradient — validation, or --dag-freezing-steps not yet passed — the reference's beta is all zeros, dag_loss.cu:340, and
            #  so is the one handed back here: the "posterior" is then soft-max_j alpha[t, j], as in the reference)
            input_to_tts = model.adaptor(decode_ops.posterior_features(alpha, beta, features)[:, 1:, :])
            features_padding_mask = ~_lengths_to_mask(sample["target_text_lengths"] - 1, input_to_tts.shape[1])
        _feat_out, _feat_out_post, _, log_dur_out, pitch_out, energy_out = model.tts(
            input_to_tts, features_padding_mask, durations=sample["durations"], pitches=sample["pitches"], energies=sample["energies"])
        src_mask = _lengths_to_mask(sample["target_text_lengths"] - 1, log_dur_out.shape[1])                 # -1: no <bos>
        F_ = min(_feat_out.shape[1], sample["target_audio"].shape[1])
        tgt_mask = _lengths_to_mask(sample["target_audio_lengths"].clamp(max=F_), F_)
        feat_out, feat = _feat_out[:, :F_][tgt_mask], sample["target_audio"][:, :F_][tgt_mask]
        l1_loss = F.l1_loss(feat_out, feat, reduction="mean")
        if _feat_out_post is not None:                                                                       # --add-postnet (:281-282)
            l1_loss = l1_loss + F.l1_loss(_feat_out_post[:, :F_][tgt_mask], feat, reduction="mean")
        pitch_loss = F.mse_loss(pitch_out[src_mask], sample["pitches"][src_mask], reduction="mean")
        energy_loss = F.mse_loss(energy_out[src_mask], sample["energies"][src_mask], reduction="mean")
        log_dur = torch.log(sample["durations"].to(log_dur_out.dtype) + 1)[src_mask]
        dur_loss = F.mse_loss(log_dur_out[src_mask], log_dur, reduction="mean")
        tts_loss = l1_loss + dur_loss + pitch_loss + energy_loss
        loss = dag_loss["loss"] + tts_loss * self.cfg.tts_loss_weight
        sample_size = 1
        logging_output = {
            "loss": loss.data, "dag-loss": dag_loss["loss"].data, "tts-loss": tts_loss.data, "l1-loss": l1_loss.data,
            "dur-loss": dur_loss.data, "pitch-loss": pitch_loss.data, "energy-loss": energy_loss.data,
            "ntokens": dag_loss["ntokens"], "nvalidtokens": dag_loss["nvalidtokens"], "nsentences": dag_loss["nsentences"],
            "invalid_nsentences": dag_loss["invalid_nsentences"], "sample_size": sample_size,
            "glat_acc": outputs.get("glat_accu", 0), "glat_keep": outputs.get("glat_keep", 0),
        }
        return loss, sample_size, logging_output

    __call__ = forward


def s2s_dag_fastspeech2_loss(model, sample: Dict[str, Tensor], glat_p="0.1", tts_loss_weight: float = 5.0,
                             glance_strategy: Optional[str] = "number-random", update_num: int = 1):
    """Functional shorthand used by bench.py: one evaluation of the released training objective (README.md:299-307:
    --glat-p 0.5:0.1@..., --glance-strategy number-random, --training-strategy expect, --tts-loss-weight 5.0)."""
    crit = S2SDAGFastSpeech2Loss(glat_p=str(glat_p), glance_strategy=glance_strategy, tts_loss_weight=tts_loss_weight)
    s = dict(sample)
    s.setdefault("update_num", update_num)
    loss, _, log = crit(model, s)
    return loss, log
