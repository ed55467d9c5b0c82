// capi_dag.hip — extern "C" entry points of the DP ops (argument checks + kernel selection).
#include "common.h"
#include <string.h>

namespace dsp {
int launch_dag_fwd_generic(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int, hipStream_t);
int launch_pick_loss(const float*, const float*, const int64_t*, const int64_t*, float*, int, int, int, int, hipStream_t);
int launch_best_alignment_generic(const float*, const float*, const int64_t*, const int64_t*, float*, int32_t*, int64_t*, int, int, int, int, hipStream_t);
int launch_dag_bwd_generic(const float*, const float*, const float*, const float*, const float*, const int64_t*, const int64_t*,
                           float*, float*, int, int, int, int, int, int, int, hipStream_t);

bool banded_supported(int L, int TR);
void caller_ws_begin(void* p, size_t n);
void caller_ws_end();
void status_begin(hipStream_t st);
void status_end(hipStream_t st);
struct CallerWsScope {
    hipStream_t st;
    CallerWsScope(void* p, size_t n, hipStream_t s) : st(s) { caller_ws_begin(p, n); status_begin(st); }
    ~CallerWsScope() { status_end(st); caller_ws_end(); }
};
void set_k5_path(int v);
void set_k5_fuse(int v);
void set_mx_cpl(int v);
void set_bt_ring(int v);
int k5_diag(unsigned int* out);
int launch_dag_banded(int mode, const float*, const float*, const int64_t*, const int64_t*, float*, float*, int32_t*, int, int, int, int, hipStream_t);
int banded_last_error_word(hipStream_t st, unsigned int* word);
int launch_max_alpha_generic(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                             float* alpha, int32_t* trace, int B, int T, int L, int TR, hipStream_t st);
int launch_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L, hipStream_t st);

bool strip4g_supported(const void* match, const void* alpha, const void* beta, int L, int TR, int ldm, int ldo);
int launch_dag_strip4g(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int, int, int, hipStream_t);


bool dense_mfma_supported(int L, int TR);
int launch_dag_dense_mfma(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int, hipStream_t);
void set_dm_depth(int v);
void set_dm_mt(int v);
void set_dx_mt(int v);
void set_xl_tile(int v);
void set_xl_mfma(int v);
void set_xl_contract(int v);
void set_dm_budget(int v);
size_t dense_rows_gated_bytes(int B, int L, int ndir);
bool dense_rows_gated_supported(int L);

bool dense_max_supported(int L, int TR);
int launch_dag_dense_max(const float*, const float*, const int64_t*, const int64_t*, float*, int64_t*, int, int, int, int, hipStream_t, unsigned short* block_trace = nullptr);
int launch_dag_dense_backtrace(const float*, const unsigned short*, const float*, const int64_t*, const int64_t*, int64_t*, int, int, int, int, hipStream_t);

bool maxstrip_supported(const void* match, const void* alpha_max, int L, int TR, int ldm, int ldo);
int launch_dag_maxstrip(const float*, const float*, const int64_t*, const int64_t*, float*, int64_t*, int, int, int, int, int, int, hipStream_t);

bool maxstripw_supported(int L, int TR);
size_t maxstripw_ws_bytes(int B, int T, int L, int TR);
int launch_dag_maxstripw(const float*, const float*, const int64_t*, const int64_t*, float*, int64_t*, int, int, int, int, int, int, hipStream_t);
bool strip1g_supported(int L, int TR);
size_t strip1g_ws_bytes(int B, int T, int L, int ndir);
int launch_dag_strip1g(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int, int, int, hipStream_t);
bool strip2g_supported(int L, int TR);
int launch_dag_strip2g(const float*, const float*, const int64_t*, const int64_t*, float*, float*, int, int, int, int, int, int, hipStream_t);

bool strip2_supported(const void* match, const void* alpha, const void* beta, const void* trace, int L, int TR);
int launch_dag_strip2(int mode, const float*, const float*, const int64_t*, const int64_t*, float*, float*, int32_t*, int, int, int, int, hipStream_t);

// test hook: dsp_dag_set_option("dp_path", n): 0 = auto, 1 = generic row-sequential, 2 = banded 2-column log-space,
// 4 = strip2 (2 columns/lane, loader wave), 5 = strip4g (4 columns/lane, exp space, one exponent per lane group),
// 7 = values-only max-DP strips + lazy back-trace for dag_best_alignment (the auto choice when trace == NULL),
// 8 = strip2g / strip1g (2 columns/lane x 64 transitions, 1 x 128: exp space, windows 33 .. 64 / 65 .. 128: the auto choice there since r06),
// 9 = dense-window exp-space blocked product on the f32 matrix cores (the auto choice for TR > 32).
// (3 and 6 were the strip4 / strip4h generations, removed in r02.)  Per THREAD: a test pinning a kernel family does not change what
// another thread's calls launch.
static thread_local int g_path = 0;
static unsigned int g_last_fallbacks = 0;
static unsigned int g_dbg[64] = {0};

static int check_dims(const char* fn, int B, int T, int L, int TR) {
    if (B < 0 || T < 1 || L < 1 || TR < 1) { set_error("%s: bad sizes B=%d T=%d L=%d TR=%d", fn, B, T, L, TR); return DSP_EINVAL; }
    return DSP_OK;
}
}  // namespace dsp

using namespace dsp;

// Scratch a forward launch takes from the caller (mirrors the launchers' own sizing; a launcher that finds the workspace too small
// falls back to the library's scratch, so an under-estimate costs a hipMalloc, not correctness).
static size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }
static size_t dense_fwd_ws_bytes(int B, int T, int L, int TR)
{
    // dense window on the matrix cores: progress words + (exponent, first-live) per (row, block)
    const size_t NJ = (size_t)(L + 63) / 64;
    const size_t halo = align256((size_t)2 * B * NJ * 4) + align256((size_t)2 * B * T * NJ * 8);
    // ... and its stand-by log-space kernels (a batch whose transitions exp space cannot hold): hand-off rows + the re-laid-out
    // ("incoming") copy of the transition matrix
    if (dense_mfma_supported(L, TR) && dense_rows_gated_supported(L))
        return align256(256 + halo + dense_rows_gated_bytes(B, L, 2)) + align256((size_t)B * L * TR * 4) + 1024;
    return align256(256 + halo) + 512;
}
static size_t dense_align_ws_bytes(int B, int T, int L, int TR)
{
    if (dense_max_supported(L, TR)) {                 // blocked max-plus DP: progress words + one block maximum per (row, block)
        const size_t NJ = (size_t)(L + 63) / 64;
        return align256(256 + align256((size_t)B * NJ * 4) + align256((size_t)B * T * NJ * 4) + (size_t)B * T * L * 2) + 512;     // (+ the block trace)
    }
    return align256((size_t)B * L * TR * 4) + align256(256 + (size_t)B * 2 * L * 8) + 1024;
}

extern "C" size_t dsp_dag_workspace_bytes(int B, int T, int L, int TR)
{
    if (B <= 0 || T <= 0 || L <= 0 || TR <= 0) return 0;
    if (TR <= 32 && !(L & 3)) {                       // strip4g: 1024-column strips when they still fill the chip, else 512
        const long ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
        const long NS = (2L * B * ns1024 >= 200) ? ns1024 : ns512;
        return align256(256 + (size_t)2 * B * NS * T * 32 * 8) + 512;
    }
    if (TR <= 64) {                                   // banded 2-column strips of 512 (log-space rows: windows 33 .. 64) ...
        const size_t banded = align256(256 + (size_t)2 * B * ((L + 511) / 512) * T * (TR <= 32 ? 32 : 64) * 8) + 512;
        // ... which the dense kernels serve instead under dp_path 9 (r05 ADVICE: sized for whichever family runs, so that the launch never falls
        // back to the library's own scratch — a hipMalloc per call, not capturable)
        const size_t dense = (TR > 32 && dense_mfma_supported(L, TR)) ? dense_fwd_ws_bytes(B, T, L, TR) : 0;
        return banded > dense ? banded : dense;
    }
    const size_t dense = dense_fwd_ws_bytes(B, T, L, TR);
    if (strip1g_supported(L, TR)) {                   // windows 65 .. 128: exp-space strips of 256 columns, 128 granules per row and strip
        const size_t strips = align256(strip1g_ws_bytes(B, T, L, 2)) + 512;
        return strips > dense ? strips : dense;
    }
    return dense;
}

// ... and the alignment (dsp_dag_best_alignment_ws): the value-only strip DP's hand-off rows, or for dense windows the re-laid-out
// ("incoming") copy of the transition matrix plus the row hand-off of the log-space max-DP.
extern "C" size_t dsp_dag_alignment_workspace_bytes(int B, int T, int L, int TR)
{
    if (B <= 0 || T <= 0 || L <= 0 || TR <= 0) return 0;
    if (TR <= 32) {
        const long ns1024 = (L + 1023) / 1024, ns512 = (L + 511) / 512;
        const long NS = ((long)B * ns1024 >= 200) ? ns1024 : ns512;
        const size_t strip = (size_t)B * ((L + 383) / 384) * T * 32 * 8;              // strip2 (with a trace buffer): 384-column strips
        const size_t mx = (size_t)B * NS * T * 32 * 8;
        return align256(256 + (strip > mx ? strip : mx)) + 512;
    }
    if (TR <= 64) {                                   // values-only strips (r06) / banded strips + trace walk; under dp_path 9 or L * 4 > 150 KB: the dense kernels
        const size_t banded = align256(256 + (size_t)B * ((L + 511) / 512) * T * 64 * 8) + 512;
        const size_t dense = dense_max_supported(L, TR) ? dense_align_ws_bytes(B, T, L, TR) : 0;
        return banded > dense ? banded : dense;
    }
    const size_t dense = dense_align_ws_bytes(B, T, L, TR);
    if (maxstripw_supported(L, TR)) {                 // windows 65 .. 128: values-only strips of 256 columns, 128 granules per row and strip
        const size_t strips = align256(maxstripw_ws_bytes(B, T, L, TR)) + 512;
        return strips > dense ? strips : dense;
    }
    return dense;
}

// Row pitches (r06, ABI 2): ld_match / ld_ab are the distances in ELEMENTS between consecutive target rows of match and of alpha / beta
// (batch stride = T * ld).  Dense tensors have ld = L (dsp_dag_loss_fwd).  A graph whose length is not a multiple of 4 — three in four are —
// keeps its rows on 16-byte boundaries by a pitch rounded up to 4: the strip kernels (windows <= 128) then serve it without a padded copy
// (dag_logsoftmax_gather_inplace writes `match` with such a pitch itself).  The other kernel families take dense tensors only.
static int check_ld(const char* fn, int L, int ld_match, int ld_ab)
{
    if (ld_match < L || ld_ab < L) { set_error("%s: row pitch smaller than L (ld_match=%d ld_ab=%d L=%d)", fn, ld_match, ld_ab, L); return DSP_EINVAL; }
    return DSP_OK;
}

extern "C" int dsp_dag_loss_fwd_ld(const float* match, int ld_match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                   float* alpha, float* beta, int ld_ab, float* loss, int B, int T, int L, int TR,
                                   void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    CallerWsScope ws_scope(workspace, workspace_bytes, as_stream(stream));
    int rc = check_dims("dag_loss_fwd", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || (!alpha && !beta)) { set_error("dag_loss_fwd: null pointer"); return DSP_EINVAL; }
    if ((rc = check_ld("dag_loss_fwd", L, ld_match, ld_ab))) return rc;
    hipStream_t st = as_stream(stream);
    const bool dense = ld_match == L && ld_ab == L;
    // auto: strip4g for the log-sum DP, strip2 for the max-DP with a trace
    if ((g_path == 0 || g_path == 5) && strip4g_supported(match, alpha, beta, L, TR, ld_match, ld_ab))
        rc = launch_dag_strip4g(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, ld_match, ld_ab, st);
    // windows 33 .. 64 (r06): exp-space strips with two vertices per lane (dag_dp_strip2g.hip); dp_path 2 keeps the log-space strips they replace
    else if ((g_path == 0 || g_path == 8) && strip2g_supported(L, TR))
        rc = launch_dag_strip2g(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, ld_match, ld_ab, st);
    // windows 65 .. 128 (r06): exp-space strips with one vertex per lane (dag_dp_strip1g.hip); dp_path 9 keeps the dense-window kernels on them
    else if ((g_path == 0 || g_path == 8) && strip1g_supported(L, TR))
        rc = launch_dag_strip1g(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, ld_match, ld_ab, st);
    else if (!dense) {
        set_error("dag_loss_fwd: pitched rows (ld_match=%d ld_ab=%d, L=%d) are served by the strip kernels of windows <= 128 only (TR <= 32: 16-byte "
                  "aligned pointers, pitches that are multiples of 4); TR=%d / this kernel pin needs dense tensors", ld_match, ld_ab, L, TR);
        return DSP_EINVAL;
    }
    else if (g_path == 4 && strip2_supported(match, alpha, beta, nullptr, L, TR))
        rc = launch_dag_strip2(0, match, links, out_len, tgt_len, alpha, beta, nullptr, B, T, L, TR, st);
    else if ((g_path == 0 || g_path == 2) && banded_supported(L, TR))
        rc = launch_dag_banded(0, match, links, out_len, tgt_len, alpha, beta, nullptr, B, T, L, TR, st);
    else if ((g_path == 0 || g_path == 9) && dense_mfma_supported(L, TR))
        rc = launch_dag_dense_mfma(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, st);
    else
        rc = launch_dag_fwd_generic(match, links, out_len, tgt_len, alpha, beta, B, T, L, TR, st);
    if (rc) return rc;
    if (loss) rc = launch_pick_loss(alpha, beta, out_len, tgt_len, loss, B, T, L, ld_ab, st);
    return rc;
}

extern "C" int dsp_dag_loss_fwd(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                float* alpha, float* beta, float* loss, int B, int T, int L, int TR,
                                void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    return dsp_dag_loss_fwd_ld(match, L, links, out_len, tgt_len, alpha, beta, L, loss, B, T, L, TR, workspace, workspace_bytes, stream);
}

extern "C" int dsp_dag_loss_bwd_ld(const float* grad_out, const float* alpha, const float* beta, int ld_ab, const float* match, int ld_match,
                                   const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                   float* grad_match, int ld_grad_match, float* grad_links, int B, int T, int L, int TR,
                                   void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    (void)workspace; (void)workspace_bytes;
    int rc = check_dims("dag_loss_bwd", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!grad_out || !alpha || !beta || !match || !links || !out_len || !tgt_len) { set_error("dag_loss_bwd: null pointer"); return DSP_EINVAL; }
    if ((rc = check_ld("dag_loss_bwd", L, ld_match, ld_ab))) return rc;
    if (grad_match && ld_grad_match < L) { set_error("dag_loss_bwd: ld_grad_match=%d smaller than L=%d", ld_grad_match, L); return DSP_EINVAL; }
    if (ld_ab != L && TR > 128) { set_error("dag_loss_bwd: pitched alpha / beta are served for TR <= 128 only (TR=%d)", TR); return DSP_EINVAL; }
    return launch_dag_bwd_generic(grad_out, alpha, beta, match, links, out_len, tgt_len, grad_match, grad_links, B, T, L, TR,
                                  ld_ab, ld_match, grad_match ? ld_grad_match : L, as_stream(stream));
}

extern "C" int dsp_dag_loss_bwd(const float* grad_out, const float* alpha, const float* beta, const float* match,
                                const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                float* grad_match, float* grad_links, int B, int T, int L, int TR,
                                void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    return dsp_dag_loss_bwd_ld(grad_out, alpha, beta, L, match, L, links, out_len, tgt_len, grad_match, L, grad_links, B, T, L, TR,
                               workspace, workspace_bytes, stream);
}

static int best_alignment_impl(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                               float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream,
                               int ld_match = 0, int ld_am = 0);

extern "C" int dsp_dag_best_alignment(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                      float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                                      dsp_stream_t stream)
{
    CallerWsScope ws_scope(nullptr, 0, as_stream(stream));                                                       // library scratch
    return best_alignment_impl(match, links, out_len, tgt_len, alpha_max, trace, path, B, T, L, TR, stream);
}

extern "C" int dsp_dag_best_alignment_ws(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                         float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                                         void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    CallerWsScope ws_scope(workspace, workspace_bytes, as_stream(stream));
    return best_alignment_impl(match, links, out_len, tgt_len, alpha_max, trace, path, B, T, L, TR, stream);
}

// ... with row pitches (see dsp_dag_loss_fwd_ld): ld_match / ld_am = elements between consecutive rows of match / alpha_max.  Pitched rows are
// served by the values-only strip DP + lazy back-trace (TR <= 32, no trace tensor); `trace` must be dense ([B,T,L]) if given and is left untouched.
extern "C" int dsp_dag_best_alignment_ld(const float* match, int ld_match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                         float* alpha_max, int ld_am, int32_t* trace, int64_t* path, int B, int T, int L, int TR,
                                         void* workspace, size_t workspace_bytes, dsp_stream_t stream)
{
    CallerWsScope ws_scope(workspace, workspace_bytes, as_stream(stream));
    return best_alignment_impl(match, links, out_len, tgt_len, alpha_max, trace, path, B, T, L, TR, stream, ld_match, ld_am);
}

static int best_alignment_impl(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                               float* alpha_max, int32_t* trace, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream,
                               int ld_match, int ld_am)
{
    int rc = check_dims("dag_best_alignment", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || !alpha_max || !path) { set_error("dag_best_alignment: null pointer"); return DSP_EINVAL; }
    hipStream_t st = as_stream(stream);
    if (ld_match == 0) ld_match = L;
    if (ld_am == 0) ld_am = L;
    if ((rc = check_ld("dag_best_alignment", L, ld_match, ld_am))) return rc;
    if (ld_match != L || ld_am != L) {
        if ((g_path == 0 || g_path == 7) && maxstrip_supported(match, alpha_max, L, TR, ld_match, ld_am))
            return launch_dag_maxstrip(match, links, out_len, tgt_len, alpha_max, path, B, T, L, TR, ld_match, ld_am, st);
        if ((g_path == 0 || g_path == 7) && maxstripw_supported(L, TR))
            return launch_dag_maxstripw(match, links, out_len, tgt_len, alpha_max, path, B, T, L, TR, ld_match, ld_am, st);
        set_error("dag_best_alignment: pitched rows (ld_match=%d ld_alpha_max=%d, L=%d) are served by the TR <= 32 strip kernels only (L <= 8192, "
                  "16-byte aligned pointers, pitches that are multiples of 4)", ld_match, ld_am, L);
        return DSP_EINVAL;
    }
    // windows 33 .. 128 (r06): values-only max-DP strips with 2 x 64 / 1 x 128 transitions per lane + the wide back-trace (dag_dp_maxstripw.hip):
    // no trace tensor.  dp_path 2 / 9 keep the log-space strips + trace walk / the blocked max-plus kernels on these windows.
    if ((g_path == 0 || g_path == 7) && maxstripw_supported(L, TR))
        return launch_dag_maxstripw(match, links, out_len, tgt_len, alpha_max, path, B, T, L, TR, L, L, st);
    // windows 33 .. 64 with a trace buffer: the banded log-space strips + trace walk (C2 at TR = 64: 2.0 ms against 3.1 for the dense kernels)
    const bool mid = TR > 32 && TR <= 64 && trace && (g_path == 0 || g_path == 2) && (size_t)L * 4 <= 160 * 1024 && banded_supported(L, TR);
    // dense window: blocked max-plus DP + trace-free back-trace (the trace buffer, if given, is left untouched)
    if ((g_path == 0 || g_path == 9) && !mid && dense_max_supported(L, TR))
        return launch_dag_dense_max(match, links, out_len, tgt_len, alpha_max, path, B, T, L, TR, st);
    // trace == NULL: values-only DP + lazy back-trace (no B*T*L trace tensor); only the banded strip kernel offers it
    if (!trace || g_path == 7) {
        if ((g_path == 0 || g_path == 7) && maxstrip_supported(match, alpha_max, L, TR, L, L))
            return launch_dag_maxstrip(match, links, out_len, tgt_len, alpha_max, path, B, T, L, TR, L, L, st);
        if (!trace) { set_error("dag_best_alignment: this shape / kernel family needs a trace buffer (see dsp_dag_alignment_trace_optional)"); return DSP_EINVAL; }
    }
    if ((size_t)L * 4 <= 160 * 1024) {
        if ((g_path == 0 || g_path == 4) && strip2_supported(match, alpha_max, nullptr, trace, L, TR)) {
            rc = launch_dag_strip2(1, match, links, out_len, tgt_len, alpha_max, nullptr, trace, B, T, L, TR, st);
            if (rc) return rc;
            return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, st);
        }
        if ((g_path == 0 || g_path == 2) && banded_supported(L, TR)) {
            rc = launch_dag_banded(1, match, links, out_len, tgt_len, alpha_max, nullptr, trace, B, T, L, TR, st);
            if (rc) return rc;
            return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, st);
        }
    }
    return launch_best_alignment_generic(match, links, out_len, tgt_len, alpha_max, trace, path, B, T, L, TR, st);
}

extern "C" int dsp_dag_max_alpha(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                 float* alpha_max, int32_t* trace, int B, int T, int L, int TR, dsp_stream_t stream)
{
    int rc = check_dims("dag_max_alpha", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || !alpha_max || !trace) { set_error("dag_max_alpha: null pointer"); return DSP_EINVAL; }
    CallerWsScope ws_scope(nullptr, 0, as_stream(stream));
    return launch_max_alpha_generic(match, links, out_len, tgt_len, alpha_max, trace, B, T, L, TR, as_stream(stream));
}

extern "C" int dsp_dag_backtrace(const int32_t* trace, const int64_t* out_len, const int64_t* tgt_len, int64_t* path, int B, int T, int L,
                                 dsp_stream_t stream)
{
    if (B < 0 || T < 1 || L < 1) { set_error("dag_backtrace: bad sizes"); return DSP_EINVAL; }
    if (B == 0) return DSP_OK;
    if (!trace || !out_len || !tgt_len || !path) { set_error("dag_backtrace: null pointer"); return DSP_EINVAL; }
    return launch_backtrace(trace, out_len, tgt_len, path, B, T, L, as_stream(stream));
}

// The same two halves on the dense-window kernels (blocked max-plus DP, 2-byte block trace, back-trace that recomputes the arg-max of the cells
// it visits): what the Viterbi graph decode runs when the window is wider than 32 (the model's default: --max-transition-length 99999).
extern "C" int dsp_dag_max_alpha_blocks_supported(int L, int TR) { return ((g_path == 0 || g_path == 9) && dense_max_supported(L, TR)) ? 1 : 0; }

extern "C" int dsp_dag_max_alpha_blocks(const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                                        float* alpha_max, uint16_t* block_trace, int B, int T, int L, int TR, dsp_stream_t stream)
{
    int rc = check_dims("dag_max_alpha_blocks", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!match || !links || !out_len || !tgt_len || !alpha_max || !block_trace) { set_error("dag_max_alpha_blocks: null pointer"); return DSP_EINVAL; }
    if (!dense_max_supported(L, TR)) { set_error("dag_max_alpha_blocks: L=%d TR=%d is not a dense-window shape (see dsp_dag_max_alpha_blocks_supported)", L, TR); return DSP_EINVAL; }
    CallerWsScope ws_scope(nullptr, 0, as_stream(stream));
    return launch_dag_dense_max(match, links, out_len, tgt_len, alpha_max, nullptr, B, T, L, TR, as_stream(stream), block_trace);
}

extern "C" int dsp_dag_backtrace_blocks(const float* alpha_max, const uint16_t* block_trace, const float* links, const int64_t* out_len,
                                        const int64_t* tgt_len, int64_t* path, int B, int T, int L, int TR, dsp_stream_t stream)
{
    int rc = check_dims("dag_backtrace_blocks", B, T, L, TR);
    if (rc) return rc;
    if (B == 0) return DSP_OK;
    if (!alpha_max || !block_trace || !links || !out_len || !tgt_len || !path) { set_error("dag_backtrace_blocks: null pointer"); return DSP_EINVAL; }
    if (!dense_max_supported(L, TR)) { set_error("dag_backtrace_blocks: L=%d TR=%d is not a dense-window shape", L, TR); return DSP_EINVAL; }
    return launch_dag_dense_backtrace(alpha_max, block_trace, links, out_len, tgt_len, path, B, T, L, TR, as_stream(stream));
}

// May the caller hand this op PITCHED rows (dsp_dag_loss_fwd_ld / _bwd_ld: op 0, dsp_dag_best_alignment_ld: op 1) for a graph of L vertices?  Only
// the strip families (windows <= 128) take them, so the answer follows the calling thread's dp_path pin like the dispatch itself does.
extern "C" int dsp_dag_pitch_supported(int op, int L, int TR)
{
    if (L < 1) return 0;
    if (TR > 32)           // windows 33 .. 128 (r06): the wide strips read match by 4-byte DMA and write their tables row by row — any pitch
        return op == 0 ? (((g_path == 0 || g_path == 8) && (strip2g_supported(L, TR) || strip1g_supported(L, TR))) ? 1 : 0)
                       : (((g_path == 0 || g_path == 7) && maxstripw_supported(L, TR)) ? 1 : 0);
    if (op == 0) return (g_path == 0 || g_path == 5) ? 1 : 0;
    return ((g_path == 0 || g_path == 7) && L <= 8192) ? 1 : 0;
}

extern "C" int dsp_dag_alignment_trace_optional(int L, int TR)
{
    if ((g_path == 0 || g_path == 7) && maxstripw_supported(L, TR)) return 1;     // windows 33 .. 128: values-only strips + wide back-trace
    if ((g_path == 0 || g_path == 2) && TR > 32 && TR <= 64) return 0;             // the banded log-space strips of this window keep a trace
    if ((g_path == 0 || g_path == 9) && dense_max_supported(L, TR)) return 1;
    return ((g_path == 0 || g_path == 7) && TR <= 32 && (L & 3) == 0 && L <= 8192) ? 1 : 0;
}

extern "C" int dsp_dag_set_option(const char* name, int value)
{
    if (name && !strcmp(name, "dp_path")) { g_path = value; return DSP_OK; }
    if (name && !strcmp(name, "k5_path")) { set_k5_path(value); return DSP_OK; }
    if (name && !strcmp(name, "k5_fuse")) { set_k5_fuse(value); return DSP_OK; }
    if (name && !strcmp(name, "mx_cpl")) { set_mx_cpl(value); return DSP_OK; }
    if (name && !strcmp(name, "bt_ring")) { set_bt_ring(value); return DSP_OK; }
    if (name && !strcmp(name, "dm_depth")) { set_dm_depth(value); return DSP_OK; }
    if (name && !strcmp(name, "dm_mt")) { set_dm_mt(value); return DSP_OK; }
    if (name && !strcmp(name, "dx_mt")) { set_dx_mt(value); return DSP_OK; }
    if (name && !strcmp(name, "xl_tile")) { set_xl_tile(value); return DSP_OK; }
    if (name && !strcmp(name, "xl_mfma")) { set_xl_mfma(value); return DSP_OK; }
    if (name && !strcmp(name, "xl_contract")) { set_xl_contract(value); return DSP_OK; }
    if (name && !strcmp(name, "dm_budget")) { set_dm_budget(value); return DSP_OK; }
    if (name && !strcmp(name, "force_generic")) { g_path = value ? 1 : 0; return DSP_OK; }
    set_error("dsp_dag_set_option: unknown option");
    return DSP_EINVAL;
}

extern "C" int dsp_dag_last_launch_status(dsp_stream_t stream, unsigned int* host_word)
{
    if (!host_word) { set_error("dsp_dag_last_launch_status: null pointer"); return DSP_EINVAL; }
    int rc = banded_last_error_word(as_stream(stream), g_dbg);
    host_word[0] = g_dbg[0];
    g_last_fallbacks = g_dbg[1];
    return rc;
}

extern "C" int dsp_dag_debug_k5(unsigned int* out4) { return out4 ? k5_diag(out4) : DSP_EINVAL; }

extern "C" const unsigned int* dsp_dag_debug_words(void) { return g_dbg; }

extern "C" unsigned int dsp_dag_last_fallback_count(void) { return g_last_fallbacks; }
