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

/* x [B,T,CI] fp16 (CI multiple of 32 in {32,64,96,128,256,512}); w [ntaps,M,CI] fp16; bias [Cout] fp32 or NULL;
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
} dsp_hg_layer;
int dsp_hifigan_conv_chain(const dsp_hg_layer* layers, int n_layers, int B, dsp_stream_t stream);

/* fp32 [B,T,C] -> fp16 [B,T,Cpad] zero padded channels (mel input) */
int dsp_hifigan_pack_input(const float* x, void* out, int B, int T, int C, int Cpad, dsp_stream_t stream);

/* conv_post: wav[b][t] = tanh( bias + sum_{k,c} w[k][c] * leaky_relu(x[b][t+k-3][c], slope) ), x fp16 [B,T,C], w fp32 [K][C] */
int dsp_hifigan_post(const void* x, const float* w, float bias, float* wav, int B, int T, int C, int K, float slope,
                     dsp_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
