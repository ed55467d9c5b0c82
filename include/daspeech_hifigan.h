/* daspeech_hifigan.h — C ABI of the HiFi-GAN generator convolutions for gfx950 (libdaspeech_hip.so).
 *
 * Replaces the torch conv stack of hifi-gan/models.py:35-43,100-119 (Generator.forward / ResBlock1.forward; twin
 * fairseq/fairseq/models/text_to_speech/hifigan.py:111-170) with ONE fused unit used for every layer:
 *
 *     out = [res +] bias + conv_shifted( leaky_relu(x, slope) )          (fp16 activations, fp32 accumulate on MFMA)
 *
 * Activations are CHANNELS-LAST fp16: x[b][t][c].  A layer is described as a set of TAPS: tap k multiplies the input row
 * t + shift[k] with the weight slab w[k][co][ci] (fp16, [ntaps][M][CI]).  A dilated Conv1d(K, dil) is K taps with
 * shift = (k - (K-1)/2)*dil; a ConvTranspose1d(kernel 2u, stride u, pad u/2) is the 2-tap layer with shifts {0,-1}, M = u*Cout
 * "phase-major" output rows (r, co) and out_mode = DSP_HG_OUT_UPSAMPLE, which scatters row (r, co) of column q to time
 * q*u + r - pad (weights pre-arranged by the host, daspeech_amd/hifigan_ops.py).
 */
#ifndef DASPEECH_HIFIGAN_H
#define DASPEECH_HIFIGAN_H

#include "daspeech_dag.h"

#ifdef __cplusplus
extern "C" {
#endif

#define DSP_HG_OUT_STORE 0      /* out[t][co]  = v                      */
#define DSP_HG_OUT_ACCUM 1      /* out[t][co] += v   (MRF sum)          */
#define DSP_HG_OUT_UPSAMPLE 2   /* out[q*u + r - pad][co] = v, M = u*Cout */
#define DSP_HG_MAX_TAPS 16

/* Weights are consumed in MFMA FRAGMENT ORDER: dsp_hifigan_pack_weights turns the tap-major [ntaps][M][CI] fp16 slab into
 * [ntaps][CI/32][ceil(M/16)][64][8] (dsp_hifigan_packed_weight_elems halves; rows >= M zero), so that one A fragment is one
 * contiguous 1 KB block.  Every `w` / `w1` / `w2` below is such a packed buffer. */
long dsp_hifigan_packed_weight_elems(int ntaps, int M, int CI);
int dsp_hifigan_pack_weights(const void* w_tap_major, void* out, int ntaps, int M, int CI, dsp_stream_t stream);

/* x [B,T,CI] fp16 (CI multiple of 32 in {32,64,96,128,256,512}); w packed from [ntaps,M,CI] fp16; bias [Cout] fp32 or NULL;
 * res [B,Tout,Cout] fp16 or NULL (added before `scale`); out [B,Tout,Cout] fp16; v = scale * (acc + bias + res).
 * pre_slope: leaky_relu slope applied to x while staging (1.0 = none).  For STORE/ACCUM Tout == T and Cout == M. */
int dsp_hifigan_conv(const void* x, const void* w, const float* bias, const void* res, void* out,
                     int B, int T, int CI, int M, int ntaps, const int* host_shifts, float pre_slope, float scale,
                     int out_mode, int up_u, int up_pad, int Tout, int Cout, dsp_stream_t stream);

/* A table of such layers launched back to back on one stream (the generator is ~100 of them per call: Generator.forward,
 * hifi-gan/models.py:100-119, unrolled by the host once per input shape).  Same arguments as dsp_hifigan_conv, per layer. */
typedef struct dsp_hg_layer {
    const void* x; const void* w; const float* bias; const void* res; void* out;
    int T, CI, M, ntaps;
    int shifts[DSP_HG_MAX_TAPS];
    float pre_slope, scale;
    int out_mode, up_u, up_pad, Tout, Cout;
    const void* w2; const float* bias2;   /* non-NULL: the record is a fused ResBlock unit (dsp_hifigan_resunit) whose first conv is
                                             (w, bias, shifts) and whose residual is x; `res` is ignored, out_mode STORE or ACCUM */
} dsp_hg_layer;
int dsp_hifigan_conv_chain(const dsp_hg_layer* layers, int n_layers, int B, dsp_stream_t stream);
/* The same for a PADDED batch: lens [B] int32 (device) = mel frames of each utterance, T0 = frames of the padded batch (every layer's T
 * is a multiple of it).  Each layer reads rows at or beyond lens[b] * (layer T / T0) as zero — the zero padding the utterance would see if
 * it were vocoded alone, as the reference does (hifi-gan/inference_e2e.py:47-56: one file at a time) — so the first lens[b] * hop samples
 * of every waveform are those of the single-utterance call, bit for bit. */
int dsp_hifigan_conv_chain_lens(const dsp_hg_layer* layers, int n_layers, int B, const int* lens, int T0, dsp_stream_t stream);

/* One ResBlock1 unit (hifi-gan/models.py:38-42: xt = c1(lrelu(x)); xt = c2(lrelu(xt)); x = xt + x) in one launch, the
 * intermediate kept in LDS:  out = scale * (x + b2 + c2(lrelu(b1 + c1(lrelu(x))))) [+ out if accumulate].
 * c1 = Conv1d(C, C, ntaps, dilation dil), c2 = Conv1d(C, C, ntaps, dilation 1), "same" padding; w1, w2 packed from [ntaps][C][C] fp16;
 * x, out [B,T,C] fp16, out != x.  Bit-identical to the two dsp_hifigan_conv launches it replaces.  C in {32, 64, 128, 256}, ntaps odd.
 * dsp_hifigan_resunit_supported() says whether a (C, ntaps, dil) unit fits the LDS tiling. */
int dsp_hifigan_resunit(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* out,
                        int B, int T, int C, int ntaps, int dil, float slope, float scale, int accumulate, dsp_stream_t stream);
int dsp_hifigan_resunit_supported(int C, int ntaps, int dil);

/* fp32 [B,T,C] -> fp16 [B,T,Cpad] zero padded channels (mel input) */
int dsp_hifigan_pack_input(const float* x, void* out, int B, int T, int C, int Cpad, dsp_stream_t stream);

/* conv_post: wav[b][t] = tanh( bias + sum_{k,c} w[k][c] * leaky_relu(x[b][t+k-3][c], slope) ), x fp16 [B,T,C], w fp32 [K][C] */
int dsp_hifigan_post(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                     dsp_stream_t stream);
/* with per-sample valid lengths lens[b] * len_mul (see dsp_hifigan_conv_chain_lens) */
int dsp_hifigan_post_lens(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                          const int* lens, int len_mul, dsp_stream_t stream);

/* ---- the same generator at the REFERENCE's precision (fp32 activations and weights: hifi-gan/models.py:100-119 as run by
 * inference_e2e.py:47-56), still on the fp16 matrix cores: operands are split x = xh + xl/2048, w = wh + wl/2048 and the three
 * significant products accumulate in fp32 (csrc/hifigan_conv_f32.hip; result within 2^-22 relative of an fp32 convolution).
 *   dsp_hifigan_pack_weights_f32   fp32 tap-major [ntaps][M][CI] -> ONE buffer [hi | lo], 2 * dsp_hifigan_packed_weight_elems halves
 *   dsp_hifigan_conv_chain_f32     a dsp_hg_layer table as above with x / res / out FP32 [B,T,C] and every weight pointer such a
 *                                  [hi | lo] buffer; w2 != NULL marks a fused ResBlock unit exactly as in dsp_hifigan_conv_chain
 *                                  (bit-identical to its two layers, the intermediate stays in LDS); lens / T0 as
 *                                  dsp_hifigan_conv_chain_lens, lens may be NULL
 *   dsp_hifigan_resunit_f32_supported   whether a (C, ntaps, dil) unit fits the fused kernel's LDS tiling
 *   dsp_hifigan_pad_input_f32      fp32 [B,T,C] -> fp32 [B,T,Cpad], zero padded channels
 *   dsp_hifigan_post_f32           conv_post + tanh on an fp32 activation tensor (lens may be NULL) */
int dsp_hifigan_pack_weights_f32(const float* w_tap_major, void* w_hi_lo, int ntaps, int M, int CI, dsp_stream_t stream);
int dsp_hifigan_resunit_f32_supported(int C, int ntaps, int dil);
int dsp_hifigan_conv_chain_f32(const dsp_hg_layer* layers, int n_layers, int B, const int* lens, int T0, dsp_stream_t stream);
int dsp_hifigan_pad_input_f32(const float* x, float* out, int B, int T, int C, int Cpad, dsp_stream_t stream);
int dsp_hifigan_post_f32(const float* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                         const int* lens, int len_mul, dsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
