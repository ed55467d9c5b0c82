/* TEST INFRASTRUCTURE ONLY (see oracle/README.md).  C restatement of the DASpeech hot path used as the
 * parity oracle.  Built by oracle/Makefile into oracle/_build/libdag_oracle.so.  Two instantiations:
 *   *_f32 : REAL=float, exp/log in float  (same arithmetic class as the HIP kernels; Viterbi bit-exact)
 *   *_f64 : REAL=double                   (the "truth" the f32 results are toleranced against)       */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>

#define REAL float
#define ACC float
#define EXP expf
#define LOG logf
#define FN(n) n##_f32
#include "dag_oracle_impl.inc"
#undef REAL
#undef ACC
#undef EXP
#undef LOG
#undef FN

#define REAL double
#define ACC double
#define EXP exp
#define LOG log
#define FN(n) n##_f64
#include "dag_oracle_impl.inc"
#undef REAL
#undef ACC
#undef EXP
#undef LOG
#undef FN

/* ---- decode / TTS-glue pieces (integer + copy work: one instantiation is enough) ---- */

/* F2 — DASpeech/models/s2s_conformer_dag_fastspeech2.py:207-217 on the COMPACT links layout:
 * tok[b,j] = argmax_v logits (first max), sc[b,j] = max_v log_softmax(logits) = -log sum exp(x - max);
 * nxt[b,i] = argmax_j (D[b,i,j] + beta*sc[b,j]) over the dense row D of restore_valid_links
 * (s2t_conformer_dag.py:157-169): columns j<=i or j-i>TR or j>=L are -inf.  torch.max keeps the FIRST
 * maximum => smallest j; an all -inf row gives index 0.  The add is done in float like the reference
 * (links + unreduced_logits.unsqueeze(1) * beta, :214). greedy: beta term dropped (:217). */
void orc_lookahead_next(const float* links, const float* sc, float beta, int use_sc,
                        int32_t* nxt, int B, int L, int TR)
{
    for (int b = 0; b < B; ++b) for (int i = 0; i < L; ++i) {
        float best = -INFINITY; int arg = 0;
        /* dense columns 0..i are -inf (+ finite) = -inf: the first of them (0) is the running winner */
        for (int d = 0; d < TR; ++d) {
            int j = i + d + 1; if (j >= L) break;
            float v = links[((size_t)b * L + i) * TR + d];
            if (use_sc) v = v + sc[(size_t)b * L + j] * beta;
            if (v > best) { best = v; arg = j; }
        }
        nxt[(size_t)b * L + i] = arg;
    }
}

/* tok/sc of F2: s2s_conformer_dag_fastspeech2.py:207-208. */
void orc_argmax_logp(const float* logits, int32_t* tok, float* sc, int B, int L, int V)
{
    for (size_t r = 0; r < (size_t)B * L; ++r) {
        const float* x = logits + r * V;
        float m = -INFINITY; int a = 0;
        for (int v = 0; v < V; ++v) if (x[v] > m) { m = x[v]; a = v; }
        float s = 0; for (int v = 0; v < V; ++v) s += expf(x[v] - m);
        tok[r] = a; sc[r] = -logf(s);          /* log_softmax at the argmax = (m - m) - log s */
    }
}

/* F3 — s2s_conformer_dag_fastspeech2.py:219-243: follow nxt from vertex 0 to L_b-1, collapse repeats,
 * drop pads, gather the hidden state of every KEPT non-bos vertex.
 * out_tokens[B,Nmax] (pad-filled), keep_idx[B,Nmax] (vertex index of each kept feature, -1 padded),
 * n_tok[b] (= 1 + n_feat[b]), features gathered into out_feat[B,Fmax,D] zero-padded (_collate_frames). */
void orc_follow_path(const int32_t* nxt, const int32_t* tok, const int64_t* out_len, int pad,
                     int64_t* out_tokens, int32_t* keep_idx, int32_t* n_feat, int B, int L, int Nmax)
{
    for (int b = 0; b < B; ++b) {
        for (int k = 0; k < Nmax; ++k) { out_tokens[(size_t)b * Nmax + k] = pad; keep_idx[(size_t)b * Nmax + k] = -1; }
        int Lb = (int)out_len[b];
        int last = tok[(size_t)b * L + 0], j = 0, n = 0;
        out_tokens[(size_t)b * Nmax + 0] = last;
        int guard = 0;
        while (j != Lb - 1 && guard++ < L) {
            j = nxt[(size_t)b * L + j];
            int now = tok[(size_t)b * L + j];
            if (now != pad && now != last) {
                if (n + 1 < Nmax) out_tokens[(size_t)b * Nmax + n + 1] = now;
                if (n < Nmax) keep_idx[(size_t)b * Nmax + n] = j;
                ++n;
            }
            last = now;
        }
        n_feat[b] = n;
    }
}

/* F7 — fairseq/fairseq/models/text_to_speech/fastspeech2.py:98-114 (LengthRegulator.forward):
 * out[b, :sum dur] = x[b].index_select(repeat(t, dur[b,t])), zero padded to max_b sum dur. Pure copy. */
void orc_length_regulate(const float* x, const int64_t* dur, float* out, int64_t* out_lens,
                         int B, int N, int C, int maxlen)
{
    memset(out, 0, sizeof(float) * (size_t)B * maxlen * C);
    for (int b = 0; b < B; ++b) {
        int64_t o = 0;
        for (int t = 0; t < N; ++t) for (int64_t r = 0; r < dur[(size_t)b * N + t]; ++r, ++o)
            if (o < maxlen) memcpy(out + ((size_t)b * maxlen + o) * C, x + ((size_t)b * N + t) * C, sizeof(float) * C);
        out_lens[b] = o;
    }
}

/* F6 (integer part) — fastspeech2.py:202-205: dur = clamp(round((exp(logdur)-1)*factor), 0), 0 at pads.
 * torch.round is round-half-to-even == rintf under the default rounding mode; exp in float like torch. */
void orc_durations(const float* log_dur, const uint8_t* pad_mask, float factor, int64_t* dur, int n)
{
    for (int i = 0; i < n; ++i) {
        float v = rintf((expf(log_dur[i]) - 1.0f) * factor);
        int64_t d = (int64_t)v; if (d < 0) d = 0;
        dur[i] = pad_mask[i] ? 0 : d;
    }
}

/* F6 (bucketize) — fastspeech2.py:169-177: torch.bucketize(v, bins) right=False: first i with bins[i] >= v. */
void orc_bucketize(const float* v, const float* bins, int nb, int64_t* out, int n)
{
    for (int i = 0; i < n; ++i) {
        int lo = 0, hi = nb;
        while (lo < hi) { int mid = (lo + hi) / 2; if (bins[mid] >= v[i]) hi = mid; else lo = mid + 1; }
        out[i] = lo;
    }
}

/* F1 — DASpeech/criterions/s2s_dag_fastspeech2_loss.py:259-263 ("expect" posterior):
 * score[b,t,:] = exp(a+b - LSE_j(a+b)), NaN -> 0 (rows that are all -inf); double accumulate;
 * expect[b,t,:] = score[b,t,:] @ features[b] (caller drops row 0). */
void orc_posterior_expect(const float* alpha, const float* beta, const float* feat, float* score,
                          float* expect, int B, int T, int L, int D)
{
    for (int b = 0; b < B; ++b) for (int t = 0; t < T; ++t) {
        const float* a = alpha + ((size_t)b * T + t) * L; const float* be = beta + ((size_t)b * T + t) * L;
        float* sc = score + ((size_t)b * T + t) * L;
        float m = -INFINITY;
        for (int j = 0; j < L; ++j) { float v = a[j] + be[j]; if (v > m) m = v; }
        if (isinf(m) || isnan(m)) { for (int j = 0; j < L; ++j) sc[j] = 0; }
        else {
            double s = 0; for (int j = 0; j < L; ++j) s += exp((double)(a[j] + be[j]) - m);
            double lse = log(s) + m;
            for (int j = 0; j < L; ++j) { float v = (float)exp((double)(a[j] + be[j]) - lse); sc[j] = isnan(v) ? 0 : v; }
        }
        float* e = expect + ((size_t)b * T + t) * D;
        for (int c = 0; c < D; ++c) {
            double acc = 0; for (int j = 0; j < L; ++j) acc += (double)sc[j] * feat[((size_t)b * L + j) * D + c];
            e[c] = (float)acc;
        }
    }
}
