// dag_dp_banded.hip — banded (TR <= 64) DAG dynamic programming for gfx950: K2 alpha, K3 beta, K6 max-alpha + trace.
//
// Replaces calculate_alpha_kernel / calculate_beta_kernel (DASpeech/custom_ops/dag_loss.cu:40-140,178-274) and
// calculate_maxalpha_kernel (dag_best_alignment.cu:39-130) in the regime the reference's tuner uses (translen 32,
// dag_loss.py:599).  MI355X-first structure, not the reference's:
//
//   * COLUMN STRIPS.  A sample's L vertices are cut into strips of 512 columns; one 256-thread workgroup owns a strip
//     for ALL T rows (2 adjacent columns per lane).  B*ceil(L/512) workgroups per direction fill the 256 CUs
//     (C2: 32*8*2 = 512 workgroups) instead of B.
//   * LINKS LIVE IN REGISTERS.  The 2*TRP incoming (alpha) / outgoing (beta) transition log-probs of a lane's columns
//     are loaded once; HBM then sees match once, alpha/beta once, links once — the algorithmic traffic of
//     SURVEY.md §8(d).
//   * ROW STATE LIVES IN LDS.  The previous DP row of the strip (+ a TRP-wide halo) is double-buffered in LDS; one
//     workgroup barrier per row; each lane reads its 33-value window with 8-byte LDS loads.
//   * STRIP-TO-STRIP HAND-OFF BY TAGGED GRANULES (cdna_hip_programming.md G16 form R2).  Strip s needs, for every row,
//     the last TRP alphas of strip s-1.  The producer stores them as 8-byte {tag=row epoch, value} write-through (sc1)
//     granules; the consumer prefetches them PF rows ahead with sc1 loads and re-polls only on a tag mismatch.  No
//     flags, no fences, no grid barrier, and — unlike the reference's spin-wait on the previous segment's counter
//     (dag_loss.cu:86-88) — every value is handed over explicitly, so there is no window-wider-than-segment race.
//   * PLACEMENT-INDEPENDENT ORDER.  Workgroups draw tickets; a strip's producer always holds a smaller ticket, so the
//     oldest unfinished workgroup never waits on an unscheduled one (no residency assumption, no deadlock).  Every
//     spin is bounded and reports through an error word.
#include "common.h"
#include <mutex>
#include <unordered_map>

namespace dsp {

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int ST_THREADS = 256;
constexpr int ST_CPT = 2;
constexpr int ST_W = ST_THREADS * ST_CPT;     // 512 columns per strip
constexpr int ST_PF = 8;                      // halo prefetch distance in rows
constexpr u32 SPIN_LIMIT = 1u << 22;

struct StripParams {
    const float* match; const float* links; const int64_t* out_len; const int64_t* tgt_len;
    float* alpha; float* beta; int32_t* trace;
    u64* halo; u32* counters;                 // counters[0] = ticket, counters[1] = error word
    u32 tag_base;
    int B, T, L, TR, NS, ndir;
    int dbg;
};

__device__ __forceinline__ u64 gran_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void gran_store(u64* p, u32 tag, float v) {
    __hip_atomic_store(p, ((u64)tag << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// MODE 0: log-sum-exp DP (alpha and/or beta; direction from the ticket).  MODE 1: max DP + trace (alpha direction).
template <int TRP, int MODE>
__global__ __launch_bounds__(ST_THREADS) void dag_strip_kernel(StripParams p)
{
    constexpr int ROWLEN = ST_W + TRP + 4;
    __shared__ __attribute__((aligned(16))) float rows[2][ROWLEN];
    __shared__ u32 s_ticket;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    if (tid == 0) s_ticket = atomicAdd(&p.counters[0], 1u);
    __syncthreads();
    const u32 ticket = s_ticket;
    const int per = p.ndir * p.B;
    const int so = (int)(ticket / per);                 // position in dependency order
    const int rem = (int)(ticket % per);
    const bool is_beta = (MODE == 0) && (p.alpha == nullptr || (p.ndir == 2 && rem >= p.B));
    const int b = rem % p.B;
    const int dirslot = (p.ndir == 2 && rem >= p.B) ? 1 : 0;
    const int s = is_beta ? (p.NS - 1 - so) : so;
    const int j0 = s * ST_W;
    const int T = p.T, L = p.L, TR = p.TR;
    const int Lb = (int)p.out_len[b], Tb = (int)p.tgt_len[b];
    const float* M = p.match + (size_t)b * T * L;
    const float* K = p.links + (size_t)b * L * TR;
    float* O = (is_beta ? p.beta : p.alpha) + (size_t)b * T * L;
    int32_t* Tr = (MODE == 1) ? p.trace + (size_t)b * T * L : nullptr;
    const int j = j0 + ST_CPT * tid;                    // this lane's first column
    const bool c0_in = j < L, c1_in = (j + 1) < L;
    const bool vec2 = ((L & 1) == 0);

    auto store_row = [&](int t, float v0, float v1) {
        float* o = O + (size_t)t * L + j;
        if (vec2) { if (c0_in) *reinterpret_cast<float2*>(o) = make_float2(v0, v1); }
        else { if (c0_in) o[0] = v0; if (c1_in) o[1] = v1; }
    };
    auto store_trace = [&](int t, int a0, int a1) {
        int32_t* o = Tr + (size_t)t * L + j;
        if (vec2) { if (c0_in) *reinterpret_cast<int2*>(o) = make_int2(a0, a1); }
        else { if (c0_in) o[0] = a0; if (c1_in) o[1] = a1; }
    };

    const bool valid = !(Tb <= 0 || Lb <= 0 || Tb > T || Lb > L);
    if (!valid || j0 >= Lb) {                            // nothing reachable in this strip: all -inf, no hand-off needed
        for (int t = 0; t < T; ++t) { store_row(t, NEG_INF, NEG_INF); if (MODE == 1) store_trace(t, -1, -1); }
        return;
    }

    // ---- transition log-probs of my two columns -> registers (once) ----
    float lk0[TRP], lk1[TRP];
    if (!is_beta) {
#pragma unroll
        for (int d = 1; d <= TRP; ++d) {                 // incoming edge (j-d) -> j : links[j-d][d-1]
            const int i0 = j - d, i1 = j + 1 - d;
            float a = (d <= TR && i0 >= 0 && c0_in) ? K[(size_t)i0 * TR + (d - 1)] : NEG_INF;
            float c = (d <= TR && i1 >= 0 && c1_in) ? K[(size_t)i1 * TR + (d - 1)] : NEG_INF;
            lk0[d - 1] = (MODE == 0) ? a * LOG2E : a;
            lk1[d - 1] = (MODE == 0) ? c * LOG2E : c;
        }
    } else {
#pragma unroll
        for (int d = 1; d <= TRP; ++d) {                 // outgoing edge j -> j+d : links[j][d-1]; successors < L_b only
            float a = (d <= TR && c0_in && j + d < Lb) ? K[(size_t)j * TR + (d - 1)] : NEG_INF;
            float c = (d <= TR && c1_in && j + 1 + d < Lb) ? K[(size_t)(j + 1) * TR + (d - 1)] : NEG_INF;
            lk0[d - 1] = a * LOG2E; lk1[d - 1] = c * LOG2E;
        }
    }

    // ---- hand-off bookkeeping ----
    const bool has_producer = so > 0 && (is_beta ? (j0 + ST_W < Lb) : true);     // beta: right strip exists and is live
    const bool has_consumer = is_beta ? (s > 0) : (s < p.NS - 1 && j0 + ST_W < Lb);
    const int prod_strip = is_beta ? s + 1 : s - 1;
    const u64* hin = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + (has_producer ? prod_strip : 0)) * (size_t)T * TRP;
    u64* hout = p.halo + ((size_t)(dirslot * p.B + b) * p.NS + s) * (size_t)T * TRP;
    const bool halo_lane = (tid < TRP);                   // wave 0 (TRP <= 64) fetches the halo
    // producer lanes: alpha -> my strip's last TRP columns; beta -> first TRP columns
    const int pub_c = is_beta ? (ST_CPT * tid) : (ST_CPT * tid - (ST_W - TRP));
    const bool pub_lane = has_consumer && pub_c >= 0 && pub_c < TRP;

    // LDS indexing (see header): alpha li = col - j0 + TRP, window starts at 2*tid; beta li = col - j0 + 1, window 2*tid+2
    const int own_li = is_beta ? (ST_CPT * tid + 1) : (ST_CPT * tid + TRP);
    const int win0 = is_beta ? (ST_CPT * tid + 2) : (ST_CPT * tid);
    const int halo_li = is_beta ? (ST_W + 1 + tid) : tid;

    const int nrows = Tb;                                 // iterations; row index t(it) below
    u64 g[ST_PF];
#pragma unroll
    for (int k = 0; k < ST_PF; ++k) g[k] = 0;
    if (has_producer && halo_lane) {
#pragma unroll
        for (int k = 0; k < ST_PF; ++k) {
            const int it = k;
            if (it < nrows) { const int t = is_beta ? (Tb - 1 - it) : it; g[k] = gran_load(hin + (size_t)t * TRP + tid); }
        }
    }

    float m_next0 = 0.f, m_next1 = 0.f;                   // match prefetch
    auto load_match = [&](int t, float& a, float& c) {
        const float* mp = M + (size_t)t * L + j;
        if (vec2) { float2 v = c0_in ? *reinterpret_cast<const float2*>(mp) : make_float2(0.f, 0.f); a = v.x; c = v.y; }
        else { a = c0_in ? mp[0] : 0.f; c = c1_in ? mp[1] : 0.f; }
    };
    { const int t0 = is_beta ? (Tb - 1) : 0; load_match(t0, m_next0, m_next1); }

    for (int itb = 0; itb < nrows; itb += ST_PF) {
#pragma unroll
        for (int k = 0; k < ST_PF; ++k) {
            const int it = itb + k;
            if (it >= nrows) break;
            const int t = is_beta ? (Tb - 1 - it) : it;
            float* cur = rows[it & 1];
            const float* prev = rows[(it & 1) ^ 1];
            const float mt0 = m_next0, mt1 = m_next1;
            if (it + 1 < nrows) load_match(is_beta ? (t - 1) : (t + 1), m_next0, m_next1);

            // ---- halo of THIS row from the neighbouring strip -> cur (consumed by the next iteration) ----
            if (halo_lane) {
                float hv = NEG_INF;
                if (has_producer) {
                    const u32 want = p.tag_base + 1u + (u32)t;
                    u64 x = g[k];
                    u32 spins = 0;
                    while (!__all((u32)(x >> 32) == want)) {
                        if ((u32)(x >> 32) != want) x = gran_load(hin + (size_t)t * TRP + tid);
                        if (++spins > SPIN_LIMIT) { if (lane == 0) atomicOr(&p.counters[1], 1u); break; }
                        __builtin_amdgcn_s_sleep(2);
                    }
                    hv = __uint_as_float((u32)x);
                    const int itn = it + ST_PF;
                    if (itn < nrows) { const int tn = is_beta ? (Tb - 1 - itn) : itn; g[k] = gran_load(hin + (size_t)tn * TRP + tid); }
                }
                cur[halo_li] = hv;
            }

            // ---- my two cells of row t ----
            float r0 = NEG_INF, r1 = NEG_INF;
            int a0 = -1, a1 = -1;
            if (it == 0) {
                if (!is_beta) { if (j == 0) r0 = (MODE == 0) ? mt0 * LOG2E : mt0; }                 // alpha[0,0] = match[0,0]
                else { if (j == Lb - 1) r0 = mt0 * LOG2E; if (j + 1 == Lb - 1) r1 = mt1 * LOG2E; }   // beta[T_b-1,L_b-1]
            } else {
                float w[TRP + 2];
#pragma unroll
                for (int q = 0; q < (TRP + 2) / 2; ++q) {
                    float2 v = *reinterpret_cast<const float2*>(prev + win0 + 2 * q);
                    w[2 * q] = v.x; w[2 * q + 1] = v.y;
                }
                const bool act0 = (j >= t) && (j < Lb), act1 = (j + 1 >= t) && (j + 1 < Lb);
                if (MODE == 0) {
                    float v0[TRP], v1[TRP];
                    float mx0 = NEG_INF, mx1 = NEG_INF;
#pragma unroll
                    for (int d = 1; d <= TRP; ++d) {
                        const float p0 = is_beta ? w[d - 1] : w[TRP - d];
                        const float p1 = is_beta ? w[d] : w[TRP + 1 - d];
                        v0[d - 1] = p0 + lk0[d - 1]; v1[d - 1] = p1 + lk1[d - 1];
                        mx0 = fmaxf(mx0, v0[d - 1]); mx1 = fmaxf(mx1, v1[d - 1]);
                    }
                    if (act0 && mx0 != NEG_INF) {
                        float sum = 0.f;
#pragma unroll
                        for (int d = 0; d < TRP; ++d) sum += __builtin_amdgcn_exp2f(v0[d] - mx0);
                        r0 = __builtin_amdgcn_logf(sum) + mx0 + mt0 * LOG2E;       // v_log_f32 is log2
                    }
                    if (act1 && mx1 != NEG_INF) {
                        float sum = 0.f;
#pragma unroll
                        for (int d = 0; d < TRP; ++d) sum += __builtin_amdgcn_exp2f(v1[d] - mx1);
                        r1 = __builtin_amdgcn_logf(sum) + mx1 + mt1 * LOG2E;
                    }
                } else {
                    // ascending predecessor index (d descending), strict '>' keeps the smallest index among ties
                    float mx0 = NEG_INF, mx1 = NEG_INF;
#pragma unroll
                    for (int d = TRP; d >= 1; --d) {
                        const float x0 = w[TRP - d] + lk0[d - 1];
                        const float x1 = w[TRP + 1 - d] + lk1[d - 1];
                        if (x0 > mx0) { mx0 = x0; a0 = j - d; }
                        if (x1 > mx1) { mx1 = x1; a1 = j + 1 - d; }
                    }
                    if (act0) r0 = mx0 + mt0; else a0 = -1;
                    if (act1) r1 = mx1 + mt1; else a1 = -1;
                }
            }
            cur[own_li] = r0; cur[own_li + 1] = r1;
            const float o0 = (MODE == 0) ? r0 * LN2 : r0, o1 = (MODE == 0) ? r1 * LN2 : r1;
            store_row(t, o0, o1);
            if (MODE == 1) store_trace(t, a0, a1);
            if (pub_lane) {
                const u32 tag = p.tag_base + 1u + (u32)t;
                gran_store(hout + (size_t)t * TRP + pub_c, tag, r0);
                gran_store(hout + (size_t)t * TRP + pub_c + 1, tag, r1);
            }
            __syncthreads();
        }
    }
    // rows the recurrence never reaches
    if (!is_beta || true) for (int t = Tb; t < T; ++t) { store_row(t, NEG_INF, NEG_INF); if (MODE == 1) store_trace(t, -1, -1); }
}

// ------------------------------------------------------------------------------------------------ host side
// ---- scratch memory of the DP launches --------------------------------------------------------------------------------------
// 1) CALLER workspace (the `workspace` / `workspace_bytes` arguments of the C ABI, sized by dsp_dag_workspace_bytes /
//    dsp_dag_alignment_workspace_bytes): the entry point opens it for the calling thread (caller_ws_begin), the launchers carve what
//    they need out of it (caller_ws_take).  It is zeroed by a hipMemsetAsync ON THE LAUNCH STREAM and the hand-off tags start at 1, so
//    a launch holds no host-side state: the memory belongs to the caller's allocator and the memset + kernel pair is hipGraph-capturable.
// 2) LIBRARY scratch (workspace == NULL or too small): a grow-only per-(device, stream) hipMalloc buffer with monotonically increasing
//    tag epochs (never re-zeroed between launches; a regrow hipFrees, i.e. synchronises) — kept for callers that pass no workspace.
struct CallerWS { char* base; size_t bytes; size_t used; };
static thread_local CallerWS t_cws = {nullptr, 0, 0};
void caller_ws_begin(void* p, size_t n) { t_cws.base = (char*)p; t_cws.bytes = p ? n : 0; t_cws.used = 0; }
void caller_ws_end() { t_cws.base = nullptr; t_cws.bytes = 0; t_cws.used = 0; }
void* caller_ws_take(size_t n)
{
    n = (n + 255) & ~(size_t)255;
    const size_t off = (256 - ((uintptr_t)(t_cws.base + t_cws.used) & 255)) & 255;          // 256-byte aligned pieces
    if (!t_cws.base || t_cws.used + off + n > t_cws.bytes) return nullptr;
    void* r = t_cws.base + t_cws.used + off;
    t_cws.used += off + n;
    return r;
}

struct BandedWS { void* base = nullptr; size_t bytes = 0; u32 tag_base = 0; void* last_status = nullptr; void* status_copy = nullptr; bool in_caller = false; };
static std::mutex g_ws_mutex;
static std::unordered_map<u64, BandedWS> g_ws;       // key: (device << 48) ^ stream

static u64 ws_key(hipStream_t st) { int devid = 0; (void)hipGetDevice(&devid); return ((u64)devid << 48) ^ (u64)(uintptr_t)st; }

static int get_ws(hipStream_t st, size_t need, BandedWS** out)
{
    BandedWS& w = g_ws[ws_key(st)];
    if (w.bytes < need) {
        if (w.base) (void)hipFree(w.base);
        w.base = nullptr; w.bytes = 0;
        hipError_t e = hipMalloc(&w.base, need);
        if (e != hipSuccess) { set_error("dag banded workspace: hipMalloc(%zu): %s", need, hipGetErrorString(e)); return (int)e; }
        w.bytes = need; w.tag_base = 0;
        e = hipMemsetAsync(w.base, 0, need, st);
        if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    }
    *out = &w;
    return DSP_OK;
}

bool banded_supported(int L, int TR) { (void)L; return TR <= 64; }

// Shared by every DP launcher that hands rows between workgroups: 256 bytes of counters (ticket, status word, fallback counters,
// debug slots) + `halo_bytes` of tagged granules / progress words.
int banded_acquire_ws(hipStream_t st, size_t halo_bytes, int T, u32** counters, u64** halo, u32* tag_base)
{
    const size_t need = 256 + halo_bytes;
    if (void* c = caller_ws_take(need)) {
        hipError_t e = hipMemsetAsync(c, 0, need, st);               // stream-ordered, capturable; tags of this launch start at 1
        if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        *counters = reinterpret_cast<u32*>(c);
        *halo = reinterpret_cast<u64*>(reinterpret_cast<char*>(c) + 256);
        *tag_base = 0;
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        BandedWS& w = g_ws[ws_key(st)];
        w.last_status = c; w.in_caller = true;
        return DSP_OK;
    }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    BandedWS* ws = nullptr;
    int rc = get_ws(st, need, &ws);
    if (rc) return rc;
    if ((u64)ws->tag_base + (u64)T + 2 > 0xFFFFFF00ull) {          // tag space exhausted: start over on clean memory
        hipError_t e = hipMemsetAsync(ws->base, 0, ws->bytes, st);
        if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
        ws->tag_base = 0;
    }
    hipError_t e = hipMemsetAsync(ws->base, 0, 256, st);             // ticket, error word, fallback counter, debug slots
    if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return (int)e; }
    *counters = reinterpret_cast<u32*>(ws->base);
    *halo = reinterpret_cast<u64*>(reinterpret_cast<char*>(ws->base) + 256);
    *tag_base = ws->tag_base;
    ws->tag_base += (u32)T + 1u;
    ws->last_status = ws->base; ws->in_caller = false;
    return DSP_OK;
}

// The status words of the last launch on a stream (dsp_dag_last_launch_status).  An entry point opens a call with status_begin — a launch
// whose kernels keep no status words (the generic row kernels) must read as "clean", not as whatever an earlier launch left (r02 fuzzing:
// a freed caller workspace re-used for an alpha table read back as status 0xFF800000, -inf) — and closes it with status_end, which moves
// status words that live in CALLER memory into a 256-byte library buffer on the launch stream: the caller may free its workspace as soon
// as the call returns.  (The buffer is allocated on first use outside a stream capture; without it the status reads as clean.)
void status_begin(hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(ws_key(st));
    if (it != g_ws.end()) { it->second.last_status = nullptr; it->second.in_caller = false; }
}
static bool status_buffer(BandedWS& w, hipStream_t st)          // (caller holds g_ws_mutex)
{
    if (!w.status_copy) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusActive; }
        if (cs != hipStreamCaptureStatusNone || hipMalloc(&w.status_copy, 256) != hipSuccess) { (void)hipGetLastError(); w.status_copy = nullptr; }
    }
    return w.status_copy != nullptr;
}
// the same move, carried out by a kernel the call launches anyway (dag_pick_loss_kernel): true = the caller's kernel copies 64 words src -> dst
bool status_export(hipStream_t st, const u32** src, u32** dst)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(ws_key(st));
    if (it == g_ws.end() || !it->second.in_caller || !it->second.last_status) return false;
    BandedWS& w = it->second;
    if (!status_buffer(w, st)) return false;
    *src = reinterpret_cast<const u32*>(w.last_status); *dst = reinterpret_cast<u32*>(w.status_copy);
    w.last_status = w.status_copy; w.in_caller = false;
    return true;
}
void status_end(hipStream_t st)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(ws_key(st));
    if (it == g_ws.end() || !it->second.in_caller || !it->second.last_status) return;
    BandedWS& w = it->second;
    (void)status_buffer(w, st);
    if (w.status_copy && hipMemcpyAsync(w.status_copy, w.last_status, 256, hipMemcpyDeviceToDevice, st) == hipSuccess) w.last_status = w.status_copy;
    else { (void)hipGetLastError(); w.last_status = nullptr; }
    w.in_caller = false;
}

// mode 0: alpha and/or beta (logsum); mode 1: max-alpha + trace
int launch_dag_banded(int mode, const float* match, const float* links, const int64_t* out_len, const int64_t* tgt_len,
                      float* alpha, float* beta, int32_t* trace, int B, int T, int L, int TR, hipStream_t st)
{
    const int TRP = TR <= 32 ? 32 : 64;
    const int NS = (L + ST_W - 1) / ST_W;
    const int ndir = (mode == 0 && alpha && beta) ? 2 : 1;
    const size_t halo_bytes = (size_t)ndir * B * NS * T * TRP * sizeof(u64);
    StripParams p;
    p.match = match; p.links = links; p.out_len = out_len; p.tgt_len = tgt_len;
    p.alpha = alpha; p.beta = beta; p.trace = trace;
    int rc = banded_acquire_ws(st, halo_bytes, T, &p.counters, &p.halo, &p.tag_base);
    if (rc) return rc;
    p.B = B; p.T = T; p.L = L; p.TR = TR; p.NS = NS; p.ndir = ndir; p.dbg = 0;
    const dim3 grid((unsigned)(ndir * B * NS)), block(ST_THREADS);
    if (mode == 0) {
        if (TRP == 32) hipLaunchKernelGGL((dag_strip_kernel<32, 0>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((dag_strip_kernel<64, 0>), grid, block, 0, st, p);
    } else {
        if (TRP == 32) hipLaunchKernelGGL((dag_strip_kernel<32, 1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((dag_strip_kernel<64, 1>), grid, block, 0, st, p);
    }
    return check_launch(mode == 0 ? "dag_loss_fwd(banded)" : "dag_best_alignment(banded)");
}

// error word of the most recent DP launch on this stream (host-synchronising; used by tests / debugging only).  With a caller
// workspace the words live in the caller's memory: valid as long as the caller has not recycled it.
int banded_last_error_word(hipStream_t st, u32* word)
{
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    auto it = g_ws.find(ws_key(st));
    if (it == g_ws.end() || !it->second.last_status) { *word = 0; return DSP_OK; }
    hipError_t e = hipMemcpyAsync(word, reinterpret_cast<char*>(it->second.last_status) + 4, 252, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_error("banded_last_error_word: %s", hipGetErrorString(e)); return (int)e; }
    return DSP_OK;
}

}  // namespace dsp
