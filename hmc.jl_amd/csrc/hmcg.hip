// hmcg.hip -- host side of libhmcgibbs.so (C ABI in include/hmcg.h).
//
// One lazily created context per HIP device: a compute stream, a copy stream, events and two grow-only
// workspaces (device memory and pinned host staging), so that a call costs no hipMalloc/hipFree once the
// workspaces have reached the size of the largest call.  Calls on one device are serialised by the context's
// mutex; different devices run concurrently (hmcg_estimate_batch_multi drives one host thread per device).
//
// The host entries stream results: a long chain is cut into a few chunks (big first, small last); chunk c's
// per-draw outputs go by SDMA into pinned staging and from there into the caller's arrays while chunk c+1
// samples.  The chain state between chunks travels through the same checkpoint block (xstate, sumacc) that
// HMCG_FLAG_RESUME exposes, so a chunked run is bit-identical to a single launch.
//
// No CPU compute path exists here: without a HIP device every compute entry returns HMCG_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "host_util.hpp"
#include "moments.hpp"
#include "variants.hpp"

namespace {

thread_local char g_err[512] = "";

void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

// Diagnostic switches (fault injection, forced kernel forms, chunking overrides, virtual devices, trace) are armed by
// HMCG_DIAG=1, read ONCE when the library is first used: without it none of them is looked at -- no getenv on the call
// path, and a stray HMCG_* variable in a production environment changes nothing.  With it they are read per call, so a
// test process can switch them between calls.
bool diag_on()
{
    static const bool on = [] { const char* e = getenv("HMCG_DIAG"); return e && atoi(e) != 0; }();
    return on;
}
const char* diag_env(const char* name) { return diag_on() ? getenv(name) : nullptr; }

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                            \
        }                                                                              \
    } while (0)

// Grow-only allocation (device memory or pinned host memory).  Growing frees and reallocates: callers lay out a
// whole call before taking pointers, and nothing survives from one call to the next.
struct Arena {
    char* base = nullptr;
    size_t cap = 0;
    bool pinned = false;
    int ensure(size_t n)
    {
        if (n <= cap) return 0;
        release();
        const size_t want = std::max(n + n / 4, (size_t)1 << 20);
        hipError_t e = pinned ? hipHostMalloc((void**)&base, want, hipHostMallocDefault) : hipMalloc((void**)&base, want);
        if (e != hipSuccess) {
            base = nullptr;
            // retry with exactly what is needed before giving up
            e = pinned ? hipHostMalloc((void**)&base, n, hipHostMallocDefault) : hipMalloc((void**)&base, n);
            if (e != hipSuccess) { base = nullptr; cap = 0; return HMCG_E_NOMEM; }
            cap = n;
            return 0;
        }
        cap = want;
        return 0;
    }
    void release()
    {
        if (base) { if (pinned) (void)hipHostFree(base); else (void)hipFree(base); }
        base = nullptr; cap = 0;
    }
};

// offsets into an arena, 256-byte aligned; pointers are taken after ensure()
struct Layout {
    size_t total = 0;
    size_t add(size_t bytes) { const size_t off = total; total += (bytes + 255) & ~(size_t)255; return off; }
};

constexpr int RING = 3;            // chunk buffers in flight (device and pinned)
constexpr int MAXBUCKET = 8;       // length buckets of one call (steps-per-thread classes 1, 2, 3, 4, 6, 8, 16)

struct DeviceCtx {
    std::mutex mu;                 // serialises calls on this device
    bool ready = false;
    int device = -1;               // the id callers use (cfg->device, device_ids[])
    int phys = -1;                 // the HIP device behind it (== device unless HMCG_VIRTUAL_DEVICES is set)
    hipStream_t stream = nullptr, copy = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;          // kernel timing
    hipEvent_t ev_scr = nullptr;                      // device entry: last use of the shared scratch (scr, mom) on any stream
    hipEvent_t evk[RING] = {}, evc[RING] = {};        // chunk pipeline: kernel done / copy done
    hipStream_t bstream[MAXBUCKET - 1] = {};          // length-bucketed dispatch: the shorter buckets' launches run beside the longest
    hipEvent_t ev_fork = nullptr, ev_join[MAXBUCKET - 1] = {};
    int cu_count = 0;
    Arena dev, pin;
    Arena mom;                     // device entry: the draw-moment tables behind extras.corr
    Arena scr;                     // device entry: the LDS-resident kernel's pdf scratch
    Arena ord;                     // device entry: the bucketed dispatch's window lists (bucket_lists_kernel)
    hmcg_hostutil::ScatterPool pool;              // host entries: helpers for the scatter into the caller's arrays
};
DeviceCtx g_ctx[HMCG_MAXDEV];
std::mutex g_init_mu;

// HMCG_VIRTUAL_DEVICES=n (diagnostics): the library offers n device ids, id i living on physical device i mod
// (physical count), each with its own context -- streams, arenas, scatter helpers.  It lets a one-GPU box execute the
// G > 1 branch of hmcg_estimate_batch_multi (worker threads, per-context workspaces, error aggregation) that otherwise
// needs a multi-GPU node.  0 / unset: ids are physical devices.
int virtual_devices()
{
    const char* e = diag_env("HMCG_VIRTUAL_DEVICES");          // (diagnostics only: read per call, so a test can switch it)
    const int n = e ? atoi(e) : 0;
    return n > 0 ? std::min(n, (int)HMCG_MAXDEV) : 0;
}

// Returns the (created on first use) context of `device`; the caller then locks ctx->mu and calls hipSetDevice(ctx->phys).
int get_context(int device, DeviceCtx** out)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_err("no HIP device available (libhmcgibbs has no CPU fallback)");
        return HMCG_E_NODEVICE;
    }
    const int nphys = n;
    if (virtual_devices() > 0) n = virtual_devices();
    if (device < 0 || device >= n || device >= HMCG_MAXDEV) {
        set_err("device %d out of range (count %d, at most %d)", device, n, HMCG_MAXDEV);
        return HMCG_E_BADARG;
    }
    DeviceCtx& c = g_ctx[device];
    std::lock_guard<std::mutex> lk(g_init_mu);
    if (!c.ready) {
        c.phys = device % nphys;
        HIP_TRY(hipSetDevice(c.phys));
        HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&c.copy, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&c.ev0));
        HIP_TRY(hipEventCreate(&c.ev1));
        HIP_TRY(hipEventCreateWithFlags(&c.ev_scr, hipEventDisableTiming));
        for (int i = 0; i < RING; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&c.evk[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c.evc[i], hipEventDisableTiming));
        }
        {
            // the shorter buckets' streams take the lowest priority: where blocks of two buckets compete for a CU, the
            // longer windows (the call's critical path) are placed first
            int lo = 0, hi = 0;
            HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));
            for (int i = 0; i < MAXBUCKET - 1; ++i) {
                HIP_TRY(hipStreamCreateWithPriority(&c.bstream[i], hipStreamNonBlocking, lo));
                HIP_TRY(hipEventCreateWithFlags(&c.ev_join[i], hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&c.ev_fork, hipEventDisableTiming));
        }
        HIP_TRY(hipDeviceGetAttribute(&c.cu_count, hipDeviceAttributeMultiprocessorCount, c.phys));
        c.pin.pinned = true;
        {
            // HMCG_SCATTER_THREADS: helper threads per device for the host-side scatter (default 3, 0 = the caller alone)
            int nw = 3;
            if (const char* e = getenv("HMCG_SCATTER_THREADS")) nw = atoi(e);
            const int hc = (int)std::thread::hardware_concurrency();
            if (hc > 0) nw = std::min(nw, std::max(0, hc - 1));
            c.pool.start(std::max(0, std::min(nw, 16)));
        }
        c.device = device;
        c.ready = true;
    }
    *out = &c;
    return 0;
}

void destroy_context(DeviceCtx& c)
{
    if (!c.ready) return;
    (void)hipSetDevice(c.phys);
    (void)hipStreamSynchronize(c.stream);        // nothing of ours may still be running when the streams go
    (void)hipStreamSynchronize(c.copy);
    (void)hipStreamDestroy(c.stream);
    (void)hipStreamDestroy(c.copy);
    (void)hipEventDestroy(c.ev0);
    (void)hipEventDestroy(c.ev1);
    (void)hipEventDestroy(c.ev_scr);
    for (int i = 0; i < RING; ++i) { (void)hipEventDestroy(c.evk[i]); (void)hipEventDestroy(c.evc[i]); }
    for (int i = 0; i < MAXBUCKET - 1; ++i) {
        (void)hipStreamSynchronize(c.bstream[i]);
        (void)hipStreamDestroy(c.bstream[i]);
        (void)hipEventDestroy(c.ev_join[i]);
        c.bstream[i] = nullptr;
    }
    (void)hipEventDestroy(c.ev_fork);
    c.dev.release();
    c.pin.release();
    c.mom.release();
    c.scr.release();
    c.ord.release();
    c.pool.stop();
    c.stream = c.copy = nullptr;
    c.ready = false;
    c.device = c.phys = -1;
}

using namespace hmcg_host;
using namespace hmcg_hostutil;
int flavour_of(const Variant& v) { return v.NH > 0 ? H : (v.occ == 2 ? P2 : P1); }
const VariantGroup* const g_groups[] = { &g_group_k2, &g_group_k3, &g_group_mid, &g_group_k3_l16, &g_group_k4, &g_group_sig, &g_group_smooth, &g_group_sigsmooth };


// The variant for (K, longest window, threads per window, path): the fewest steps per thread that cover the
// window, then the flavour -- `force` (>= 0, diagnostics) or the table's preference for the batch size.
const Variant* pick_variant(int K, int maxT, int nt_req, bool sig, bool smooth, bool small_batch, int force)
{
    const int nt = nt_req > 0 ? nt_req : 256;
    const Variant* best = nullptr;
    for (const VariantGroup* g : g_groups)
    for (int i = 0; i < g->n; ++i) {
        const Variant& v = g->v[i];
        if (v.K != K || v.NT != nt || v.L * v.NT < maxT || v.sig != sig || v.smooth != smooth) continue;
        const int want = force >= 0 ? force : (small_batch ? v.pref_small : v.pref_big);
        const bool better = !best || v.L < best->L ||
                            (v.L == best->L && flavour_of(v) == want) ||
                            (v.L == best->L && flavour_of(*best) != want && flavour_of(v) == P1);
        if (better) best = &v;
    }
    return best;
}

int validate(const hmcg_config* cfg)
{
    if (!cfg) { set_err("cfg is NULL"); return HMCG_E_BADARG; }
    if (cfg->struct_size != (int32_t)sizeof(hmcg_config)) {
        set_err("hmcg_config.struct_size %d != %d", cfg->struct_size, (int)sizeof(hmcg_config));
        return HMCG_E_BADARG;
    }
    if (cfg->W < 1 || cfg->K < 2 || cfg->K > HMCG_MAXK || cfg->ldY < 2 || cfg->burnin < 0 || cfg->nrun < 0 ||
        cfg->H < 0 || cfg->H > HMCG_MAXH || cfg->max_T < 0 || cfg->max_T > cfg->ldY || cfg->sweep_base < 0 || cfg->sweep_count < 0 || cfg->n_samples < 0 || cfg->kappa < 0.0 || cfg->min_T < 0) {
        set_err("bad hmcg_config (W=%d K=%d ldY=%d max_T=%d burnin=%d nrun=%d H=%d)", cfg->W, cfg->K, cfg->ldY,
                cfg->max_T, cfg->burnin, cfg->nrun, cfg->H);
        return HMCG_E_BADARG;
    }
    for (int h = 0; h < cfg->H; ++h)
        if (cfg->horizons[h] < 0) { set_err("negative horizon"); return HMCG_E_BADARG; }
    if ((long long)(cfg->n_samples > 1 ? cfg->n_samples : 1) * ((long long)cfg->burnin + cfg->nrun) > 0x7fffffffLL) {
        set_err("n_samples * (burnin + nrun) exceeds 2^31 - 1 sweeps");
        return HMCG_E_BADARG;
    }
    return 0;
}

// What runs: the kernel instantiation for this call's shape, chosen once per call.
// One length bucket of a call: the windows with t_lo < T <= t_hi run on variant v (its own launch, beside the others).
struct Bucket {
    const Variant* v;
    int t_lo, t_hi;
};
struct Plan {
    const Variant* v = nullptr;    // register-resident kernel (with buckets: the longest bucket's variant)
    int nb = 0;                    // > 1: length-bucketed dispatch, longest bucket first
    Bucket b[MAXBUCKET];
    const BigVariant* bv = nullptr;
    int bigL = 0;
    bool stream = false;           // the LDS-resident kernel's HBM-streaming form (window too long for the CU's LDS)
    size_t dyn = 0;
    bool use_sig = false, use_smooth = false;
    bool needs_pif() const { return bv != nullptr && use_smooth; }   // the LDS-resident smoothing kernel streams pif through pif_final
    int NT() const { return v ? v->NT : bv->NT; }
    int L() const { return v ? v->L : bigL; }
    int NH() const { return v ? v->NH : 0; }
    // LDS-resident kernel: per-step pdfs handed from the product phase to the replay, [W][L][ceil(K/2)][NT][2] doubles
    size_t scratch_bytes(int W, int K) const { return bv ? sizeof(double) * (size_t)W * (size_t)bigL * (size_t)(2 * hmcg::big_scratch_pairs(K)) * (size_t)bv->NT : 0; }
    // streaming form: per window the observations, uniforms, state maps and states of its NT * L steps
    size_t slab_bytes() const { return stream ? hmcg::stream_slab_bytes((size_t)bv->NT * (size_t)bigL) : 0; }
    size_t stream_bytes(int W) const { return slab_bytes() * (size_t)W; }
    const void* fptr() const { return v ? reinterpret_cast<const void*>(v->fn) : reinterpret_cast<const void*>(bv->fn); }
};

// Static LDS of a kernel instantiation, asked of the runtime once per function (make_plan and fill_timing sit on the call path).
size_t static_lds_bytes(const void* fn, size_t fallback)
{
    static std::mutex mu;
    static std::vector<std::pair<const void*, size_t>> cache;
    std::lock_guard<std::mutex> lk(mu);
    for (const auto& e : cache) if (e.first == fn) return e.second;
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, fn) != hipSuccess) return fallback;      // (not cached: asked again next time)
    cache.emplace_back(fn, fa.sharedSizeBytes);
    return fa.sharedSizeBytes;
}

// The steps-per-thread classes compiled for a path at 256 threads per window, ascending.
int length_classes(int K, bool sig, bool smooth, int* Ls)
{
    int n = 0;
    for (const VariantGroup* g : g_groups)
        for (int i = 0; i < g->n; ++i) {
            const Variant& v = g->v[i];
            if (v.K != K || v.NT != 256 || v.sig != sig || v.smooth != smooth) continue;
            bool seen = false;
            for (int j = 0; j < n; ++j) seen |= Ls[j] == v.L;
            if (!seen && n < 16) Ls[n++] = v.L;
        }
    std::sort(Ls, Ls + n);
    return n;
}

int flavour_code(const char* f) { return !strcmp(f, "h") ? H : (!strcmp(f, "p2") ? P2 : P1); }

// The window lengths of a call as the host entries know them (rows idx[0..n) of T; idx == nullptr: rows 0..n-1).
struct HostLengths {
    const int32_t* T = nullptr;
    const int32_t* idx = nullptr;
    int n = 0;
    int at(int i) const { return T[idx ? (size_t)idx[i] : (size_t)i]; }
};

// Argument checks common to both entries + kernel choice.  W is the number of windows THIS device runs; minT the shortest
// of them when the caller knows it (0: unknown -- one launch sized for the longest window); hl (host entries) the lengths
// themselves: classes no window falls in are not launched.
int make_plan(const hmcg_config* cfg, const hmcg_extras* ex, int W, int cu_count, int minT, const HostLengths* hl, Plan* plan)
{
    if (ex && ex->struct_size != (int32_t)sizeof(hmcg_extras)) { set_err("hmcg_extras.struct_size mismatch"); return HMCG_E_BADARG; }
    const bool resume = (cfg->flags & HMCG_FLAG_RESUME) != 0;
    if (resume && !(ex && ex->xstate)) { set_err("HMCG_FLAG_RESUME needs extras.xstate"); return HMCG_E_BADARG; }
    const int maxT = cfg->max_T > 0 ? cfg->max_T : cfg->ldY;
    const bool use_sig = ex && (ex->sig_range != nullptr);
    const int n_samples = cfg->n_samples > 1 ? cfg->n_samples : 1;
    if (!use_sig && (n_samples > 1 || (ex && (ex->sigma_signal || ex->sigvals)))) {
        set_err("n_samples / sigma_signal / sigvals need extras.sig_range");
        return HMCG_E_BADARG;
    }
    if (!use_sig && (cfg->blend_mask != 0 || (ex && ex->end_pos))) {
        set_err("blend_mask / end_pos need extras.sig_range");
        return HMCG_E_BADARG;
    }
    if (cfg->blend_mask < 0 || (cfg->H < 31 && (cfg->blend_mask >> cfg->H) != 0)) { set_err("blend_mask has bits beyond H"); return HMCG_E_BADARG; }
    if (ex && ex->sigvals && ex->nsave_ld < 1) { set_err("sigvals needs nsave_ld >= 1"); return HMCG_E_BADARG; }
    if (!use_sig && ex && ex->sample_summary) { set_err("sample_summary needs extras.sig_range (without the signal path it is `summary`)"); return HMCG_E_BADARG; }
    if (ex && ex->corr) {
        if (use_sig || n_samples > 1 || cfg->H < 1 || cfg->nrun < 2) {
            set_err("extras.corr: base runs only (no signal path), H >= 1 (the forecast column) and nrun >= 2");
            return HMCG_E_BADARG;
        }
        if (resume || cfg->sweep_base != 0 || (cfg->sweep_count > 0 && cfg->sweep_count < cfg->burnin + cfg->nrun)) {
            set_err("extras.corr needs the whole run in one call (no RESUME / sweep_base / sweep_count)");
            return HMCG_E_BADARG;
        }
    }
    const bool use_smooth = ex && (ex->pi_smooth_mean != nullptr || ex->pi_filter_mean != nullptr || ex->pi_smooth_draws != nullptr);
    if (cfg->sweep_base > n_samples * (cfg->burnin + cfg->nrun)) { set_err("sweep_base beyond the run"); return HMCG_E_BADARG; }
    // Flavour: helper waves pay off while every window has a CU to itself; with more windows than CUs the capped
    // plain variant lets two windows share a CU instead (a helped block takes the whole register file).
    // HMCG_FLAVOUR=p1|p2|h and HMCG_HELPERS=0|1 override the table (diagnostics, tools/variant_sweep.py).
    const bool small_batch = W <= cu_count;
    int force = -1;
    if (const char* henv = diag_env("HMCG_HELPERS")) force = atoi(henv) != 0 ? H : (small_batch ? P1 : P2);
    if (const char* fenv = diag_env("HMCG_FLAVOUR")) force = flavour_code(fenv);
    Plan pl;
    pl.use_sig = use_sig; pl.use_smooth = use_smooth;
    // HMCG_FORCE_BIG=1 (diagnostics): the LDS-resident kernel also where a register-resident variant exists
    if (cfg->K < 5 && !diag_env("HMCG_FORCE_BIG")) pl.v = pick_variant(cfg->K, maxT, cfg->threads_per_window, use_sig, use_smooth, small_batch, force);
#ifndef HMCG_STAMPS
    // Length-bucketed dispatch: a batch of ragged windows (the reference's production run: 460 expanding windows of 120..579
    // months, code/run_hmm.jl:79-109) is cut by the steps-per-thread class each window needs; every class gets its own launch
    // on its own stream, all of them over the whole grid -- the blocks of the other classes' windows leave at once
    // (the class bounds in the header of KernelParams::order).  A window then runs on the variant its own length selects, whatever else the call holds:
    // its result equals that of a call with this window alone, bit for bit.  (HMCG_NO_BUCKETS=1, diagnostics: one launch
    // sized for the longest window, as before round 4.)
    if (pl.v && pl.v->NT == 256 && cfg->threads_per_window == 0 && minT > 0 && minT < maxT && !diag_env("HMCG_NO_BUCKETS")) {
        int Ls[16];
        const int nL = length_classes(cfg->K, use_sig, use_smooth, Ls);
        int lo = 0, hi = 0;
        while (lo < nL && 256 * Ls[lo] < minT) ++lo;
        while (hi < nL && 256 * Ls[hi] < maxT) ++hi;
        if (hi < nL && lo < hi) {
            // classes that are launched, longest first: all of them on the device entry (it does not see T); on the host
            // entries only those a window falls in (the longest one always: it reports, and flags T > max_T)
            int keep[16], nk = 0;
            for (int c = hi; c >= lo; --c) {
                bool any = !hl || c == hi;
                const int c_lo = c == 0 ? 0 : 256 * Ls[c - 1], c_hi = 256 * Ls[c];
                for (int i = 0; hl && i < hl->n && !any; ++i) any = hl->at(i) > c_lo && hl->at(i) <= c_hi;
                if (any) keep[nk++] = c;
            }
            nk = std::min(nk, MAXBUCKET);          // (more classes than slots: the last slot's class takes every shorter window too)
            const char* bf = diag_env("HMCG_BUCKET_FLAVOURS");          // "h,p2,p2": longest bucket first (diagnostics)
            for (int j = 0; j < nk; ++j) {
                int f = force;
                if (bf && *bf) {
                    char tok[8] = "";
                    size_t n = strcspn(bf, ",");
                    memcpy(tok, bf, std::min(n, sizeof tok - 1));
                    f = flavour_code(tok);
                    bf += n + (bf[n] == ',' ? 1 : 0);
                }
                // bucket j: windows longer than the next kept class holds, up to what this class holds (a skipped class is
                // empty, so every window still runs on the smallest class that covers it)
                const int t_hi = j == 0 ? INT32_MAX : 256 * Ls[keep[j]], t_lo = j == nk - 1 ? INT32_MIN : 256 * Ls[keep[j + 1]];
                pl.b[pl.nb++] = Bucket{pick_variant(cfg->K, 256 * Ls[keep[j]], 0, use_sig, use_smooth, small_batch, f), t_lo, t_hi};
            }
            if (pl.nb == 1) pl.nb = 0;             // one class after all: a plain single launch
            pl.v = pl.b[0].v;
        }
    }
#endif
    if (!pl.v) {                                       // large K, or a window too long for the register-resident variants
        const BigVariant* tab = use_sig ? (use_smooth ? g_big_sigsmooth_variants : g_big_sig_variants) : (use_smooth ? g_big_smooth_variants : g_big_variants);
        const int ntab = use_sig ? (use_smooth ? g_n_big_sigsmooth_variants : g_n_big_sig_variants) : (use_smooth ? g_n_big_smooth_variants : g_n_big_variants);
        for (int i = 0; i < ntab; ++i) if (tab[i].K == cfg->K) pl.bv = &tab[i];
        if (pl.bv) {
            pl.bigL = (maxT + pl.bv->NT - 1) / pl.bv->NT;
            pl.dyn = (size_t)pl.bv->NT * pl.bigL * (8 + 8 + 4 + 1) + 16;
            // dynamic + static LDS of the instantiation must fit the CU's 160 KiB
            const size_t stat = static_lds_bytes(reinterpret_cast<const void*>(pl.bv->fn), 48 * 1024);
            if (cfg->threads_per_window != 0 && cfg->threads_per_window != pl.bv->NT) pl.bv = nullptr;
            else if (pl.dyn + stat > 160 * 1024 || diag_env("HMCG_FORCE_STREAM")) {
                // too long for the LDS: the same kernel with its per-step arrays in an HBM scratch (HMCG_FORCE_STREAM: tests)
                pl.bv = nullptr;
                const BigVariant* stab = use_sig ? (use_smooth ? g_big_sigsmooth_stream_variants : g_big_sig_stream_variants)
                                                 : (use_smooth ? g_big_smooth_stream_variants : g_big_stream_variants);
                const int nstab = use_sig ? (use_smooth ? g_n_big_sigsmooth_stream_variants : g_n_big_sig_stream_variants)
                                          : (use_smooth ? g_n_big_smooth_stream_variants : g_n_big_stream_variants);
                for (int i = 0; i < nstab; ++i) if (stab[i].K == cfg->K) pl.bv = &stab[i];
                pl.stream = pl.bv != nullptr;
                pl.dyn = 16;
            }
        }
    }
    if (!pl.v && !pl.bv) {
        set_err("no kernel for K=%d max_T=%d threads_per_window=%d", cfg->K, maxT, cfg->threads_per_window);
        return HMCG_E_UNSUPPORTED;
    }
    *plan = pl;
    return 0;
}

// Kernel parameters common to every launch of a call; per-launch fields (sweep range, resume, output window) are
// filled by the caller.  All pointers are device pointers.
hmcg::KernelParams base_params(const hmcg_config* cfg, int W, const double* dY, const int32_t* dT, const double* dyreal,
                               int32_t* dstatus, const hmcg_extras* dex, bool use_sig)
{
    hmcg::KernelParams p{};
    const int n_samples = cfg->n_samples > 1 ? cfg->n_samples : 1;
    p.Y = dY; p.T = dT; p.yreal = dyreal;
    p.ldY = cfg->ldY; p.W = W; p.H = cfg->H;
    p.per_sample = cfg->burnin + cfg->nrun;
    p.burnin_s = cfg->burnin; p.nrun_s = cfg->nrun; p.n_samples = n_samples; p.nd = n_samples * cfg->nrun;
    if (p.per_sample < 1) p.per_sample = 1;
    p.kappa = cfg->kappa;
    for (int h = 0; h < HMCG_MAXH; ++h) p.horizons[h] = h < cfg->H ? cfg->horizons[h] : 0;
    p.seed_lo = (uint32_t)cfg->seed; p.seed_hi = (uint32_t)(cfg->seed >> 32); p.window_base = cfg->window_base;
    p.alpha = cfg->alpha > 0.0 ? cfg->alpha : 1.0;
    p.nu = cfg->nu > 0.0 ? cfg->nu : 1.0;
    p.status = dstatus;
    if (dex) {
        p.pi_smooth_mean = dex->pi_smooth_mean; p.pi_filter_mean = dex->pi_filter_mean; p.pi_smooth_draws = dex->pi_smooth_draws;
        p.sig_range = dex->sig_range; p.save_range = dex->save_range; p.sigma_signal = dex->sigma_signal;
        p.sigvals = dex->sigvals; p.nsave_ld = dex->nsave_ld;
        p.x_init = dex->x_init; p.x_final = dex->x_final; p.pif_final = dex->pif_final; p.xstate = dex->xstate;
        p.sumacc = dex->sumacc; p.window_ids = dex->window_ids;
        if (use_sig) { p.end_pos = dex->end_pos; p.blend_mask = cfg->blend_mask; p.sample_summary = dex->sample_summary; }
    }
    return p;
}

// ---- the bucketed dispatch's window lists (KernelParams::order): per bucket a block of 4 + W words (n, t_lo, t_hi, -, ids),
// then two words: the number of class changes between neighbouring windows, and the ticket of the last block ----
constexpr int ORD_HDR = 4;
struct BucketRanges { int nb; int compact; int lo[MAXBUCKET], hi[MAXBUCKET]; };
__global__ void bucket_lists_kernel(const int32_t* T, int W, BucketRanges r, int32_t* ord)
{
    const size_t stride = (size_t)W + ORD_HDR;
    int32_t* meta = ord + (size_t)MAXBUCKET * stride;
    const int w = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (w < W) {
        auto bucket_of = [&](int t) { int k = 0; for (int b = 0; b < r.nb; ++b) if (t > r.lo[b] && t <= r.hi[b]) k = b; return k; };
        const int mine = bucket_of(T[w]);           // exactly one bucket: the first reaches INT32_MAX, the last starts at INT32_MIN
        int32_t* list = ord + (size_t)mine * stride;
        const int pos = atomicAdd(&list[0], 1);
        list[ORD_HDR + pos] = w;
        if (w > 0 && bucket_of(T[w - 1]) != mine) atomicAdd(&meta[0], 1);
    }
    // the last block to get here closes the lists: ranges into the headers; and when every class is one run of the caller's
    // windows (changes == non-empty classes - 1) there is nothing to compact -- n = -1, block b keeps window b (1.7 % faster
    // than the compacted launches on the sorted production batch, measured).  r.compact == 0 (diagnostics): never compact.
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(&meta[1], 1) == (int)gridDim.x - 1) {
        __threadfence();
        int runs = 0;
        for (int b = 0; b < r.nb; ++b) runs += atomicAdd(&ord[(size_t)b * stride], 0) > 0 ? 1 : 0;
        const bool sorted = atomicAdd(&meta[0], 0) == runs - 1 || !r.compact;
        for (int b = 0; b < r.nb; ++b) {
            int32_t* list = ord + (size_t)b * stride;
            if (sorted) list[0] = -1;
            list[1] = r.lo[b]; list[2] = r.hi[b];
        }
    }
}
size_t bucket_list_bytes(int W) { return sizeof(int32_t) * ((size_t)MAXBUCKET * ((size_t)W + ORD_HDR) + 2); }
// Enqueues the list construction on `stream` (T on the device): one memset, one kernel.  The order inside a list is whatever
// the atomics give: every window's result is independent of the block that runs it.
int build_bucket_lists(const Plan& pl, const int32_t* dT, int W, int32_t* ord, hipStream_t stream)
{
    BucketRanges r{};
    r.nb = pl.nb;
    r.compact = diag_env("HMCG_NO_BUCKET_LISTS") ? 0 : 1;
    for (int b = 0; b < pl.nb; ++b) { r.lo[b] = pl.b[b].t_lo; r.hi[b] = pl.b[b].t_hi; }
    HIP_TRY(hipMemsetAsync(ord, 0, bucket_list_bytes(W), stream));
    hipLaunchKernelGGL(bucket_lists_kernel, dim3((unsigned)((W + 255) / 256)), dim3(256), 0, stream, dT, W, r, ord);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ord: the lists built by build_bucket_lists for this plan and these windows (bucketed plans), or null
int launch_kernel(DeviceCtx& c, const Plan& pl, const hmcg::KernelParams& p, hipStream_t stream, const int32_t* ord)
{
    if (pl.nb > 1) {
        // fork: every bucket's launch waits for what precedes this call on `stream`; the longest bucket runs on `stream`
        // itself, the others on the context's bucket streams; join: `stream` waits for all of them
        HIP_TRY(hipEventRecord(c.ev_fork, stream));
        for (int b = 0; b < pl.nb; ++b) {
            hipStream_t bs = b == 0 ? stream : c.bstream[b - 1];
            if (b > 0) HIP_TRY(hipStreamWaitEvent(bs, c.ev_fork, 0));
            hmcg::KernelParams q = p;
            q.order = ord + (size_t)b * ((size_t)p.W + ORD_HDR);
            const Variant* v = pl.b[b].v;
            hipLaunchKernelGGL(v->fn, dim3((unsigned)p.W), dim3((unsigned)(v->NT + 64 * v->NH)), 0, bs, q);
            HIP_TRY(hipGetLastError());
            if (b > 0) HIP_TRY(hipEventRecord(c.ev_join[b - 1], bs));
        }
        for (int b = 1; b < pl.nb; ++b) HIP_TRY(hipStreamWaitEvent(stream, c.ev_join[b - 1], 0));
        return 0;
    }
    if (pl.v) hipLaunchKernelGGL(pl.v->fn, dim3((unsigned)p.W), dim3((unsigned)(pl.v->NT + 64 * pl.v->NH)), 0, stream, p);
    else hipLaunchKernelGGL(pl.bv->fn, dim3((unsigned)p.W), dim3((unsigned)pl.bv->NT), pl.dyn, stream, p, pl.bigL);
    HIP_TRY(hipGetLastError());
    return 0;
}

void fill_timing(hmcg_timing* t, const Plan& pl, const DeviceCtx& c, double kernel_ms, int launches, double call_ms, int windows)
{
    if (!t) return;
    t->kernel_ms = kernel_ms;
    t->launches = launches;
    t->threads_per_window = pl.NT();
    t->steps_per_thread = pl.L();
    t->helper_waves = pl.NH();
    t->device = c.device;
    t->call_ms = call_ms;
    t->windows = windows;
    t->occupancy = pl.v ? pl.v->occ : 0;
    t->buckets = pl.nb > 1 ? pl.nb : 1;
    t->streaming = pl.stream ? 1 : 0;
    t->lds_bytes = (int32_t)(static_lds_bytes(pl.fptr(), 0) + pl.dyn);
}

#ifdef HMCG_STAMPS
int print_stamps(const hmcg::KernelParams& p, const Plan& pl, unsigned long long* ddbg, size_t ndbg, hipStream_t stream)
{
    static const char* names[HMCG_NSTAMP] = {"Ba wait", "param draws | shadow jobs", "Bb wait", "theta+ux+pdfs", "local product",
        "wave scan", "Bc wait", "prefix+replay+last", "Bd wait", "maps+compose", "map scan", "Be wait", "apply", "publish stats", "  (shadow: outputs)", "  (shadow: prep)", "  (param: counts+row sums)", "  (param: shapes)", "  (param: gamma)", "unused"};
    const int nwv = pl.NT() / 64 + pl.NH();
    std::vector<unsigned long long> h(ndbg);
    HIP_TRY(hipStreamSynchronize(stream));
    // HMCG_STAMPS_AFTER=n: stay silent for the first n launches (tools/stamps.py warms the chip up for >= 2 s first)
    static int launches_seen = 0;
    static const int print_after = diag_env("HMCG_STAMPS_AFTER") ? atoi(diag_env("HMCG_STAMPS_AFTER")) : 0;
    if (launches_seen++ < print_after) return 0;
    HIP_TRY(hipMemcpy(h.data(), ddbg, ndbg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    const int nsw = p.sweep_end - p.sweep_begin;
    {
        // in-kernel clock of every wave: d(s_memtime) / d(s_memrealtime) * 100 MHz around the sweep loop
        std::vector<double> clk;
        double ticks = 0;
        for (int w = 0; w < p.W; ++w)
            for (int wv = 0; wv < nwv; ++wv) {
                const unsigned long long* o = &h[((size_t)w * nwv + wv) * HMCG_NSTAMP_ALL];
                if (o[HMCG_NSTAMP + 1]) { clk.push_back(100.0 * (double)o[HMCG_NSTAMP] / (double)o[HMCG_NSTAMP + 1]); ticks += (double)o[HMCG_NSTAMP]; }
            }
        std::sort(clk.begin(), clk.end());
        if (!clk.empty())
            fprintf(stderr, "[clock] launch %d: in-kernel clock MHz min %.0f median %.0f max %.0f (s_memtime / s_memrealtime x 100 MHz); "
                            "%.0f ticks per sweep (mean over waves)\n", launches_seen - 1, clk.front(), clk[clk.size() / 2], clk.back(),
                    ticks / (double)clk.size() / (nsw > 0 ? nsw : 1));
    }
    fprintf(stderr, "[stamps] K=%d L=%d NT=%d W=%d sweeps=%d: mean cycles per sweep by wave (s_memtime ticks)\n",
            pl.v ? pl.v->K : pl.bv->K, pl.L(), pl.NT(), p.W, nsw);
    fprintf(stderr, "%-24s", "phase");
    for (int wv = 0; wv < nwv; ++wv) fprintf(stderr, "   wave%-2d", wv);
    fprintf(stderr, "\n");
    std::vector<double> tot(nwv, 0.0);
    for (int i = 0; i < HMCG_NSTAMP; ++i) {
        fprintf(stderr, "%-24s", names[i]);
        for (int wv = 0; wv < nwv; ++wv) {
            double acc = 0;
            for (int w = 0; w < p.W; ++w) acc += (double)h[((size_t)w * nwv + wv) * HMCG_NSTAMP_ALL + i];
            acc /= (double)p.W * (nsw > 0 ? nsw : 1);
            tot[wv] += acc;
            fprintf(stderr, " %8.0f", acc);
        }
        fprintf(stderr, "\n");
    }
    fprintf(stderr, "%-24s", "total");
    for (int wv = 0; wv < nwv; ++wv) fprintf(stderr, " %8.0f", tot[wv]);
    fprintf(stderr, "\n");
    return 0;
}
#endif

// ---- device-resident entry: one launch over caller-owned HBM buffers -----------------------------------------
int launch_device(DeviceCtx& c, const hmcg_config* cfg, const double* dY, const int32_t* dT, const double* dyreal, double* dmu,
                  double* dsig2, double* dA, double* dpi_end, double* dfcast, double* dsummary, int32_t* dstatus,
                  const hmcg_extras* ex, hipStream_t stream, hmcg_timing* timing)
{
    if (!dY || !dT || !dstatus) { set_err("Y, T and status are required"); return HMCG_E_BADARG; }
    Plan pl;
    int rc = make_plan(cfg, ex, cfg->W, c.cu_count, cfg->min_T, nullptr, &pl);
    if (rc) return rc;
    if (pl.needs_pif() && !(ex && ex->pif_final)) {
        set_err("pi_smooth_mean / pi_filter_mean for K >= 5 or windows beyond the register-resident kernels need extras.pif_final "
                "([W][ldY][K]: the running sweep's filtered probabilities pass through it)");
        return HMCG_E_BADARG;
    }
    if (ex && ex->corr && (!dmu || !dsig2 || !dA || !dpi_end || !dfcast)) {
        set_err("extras.corr on the device entry needs all five per-draw outputs");
        return HMCG_E_BADARG;
    }
    const bool resume = (cfg->flags & HMCG_FLAG_RESUME) != 0;
    hmcg::KernelParams p = base_params(cfg, cfg->W, dY, dT, dyreal, dstatus, ex, pl.use_sig);
    const int total_sweeps = p.n_samples * (cfg->burnin + cfg->nrun);
    p.sweep_begin = cfg->sweep_base;
    p.sweep_end = total_sweeps;
    if (cfg->sweep_count > 0 && cfg->sweep_base + cfg->sweep_count < p.sweep_end) p.sweep_end = cfg->sweep_base + cfg->sweep_count;
    p.resume = resume ? 1 : 0;
    p.final_launch = (p.sweep_end == total_sweeps) ? 1 : 0;
    p.mu = dmu; p.sig2 = dsig2; p.A = dA; p.pi_end = dpi_end; p.fcast = dfcast; p.summary = dsummary;
    p.nd_ld = p.nd; p.draw_off = 0;

    if (!resume) HIP_TRY(hipMemsetAsync(dstatus, 0, sizeof(int32_t) * (size_t)cfg->W, stream));
    // The pdf scratch and the moment tables belong to the device context, not to the call: an enqueue-only call on another
    // stream may still be using them, so this launch is ordered behind their last use.
    const bool use_lists = pl.nb > 1;
    const bool uses_scratch = pl.bv != nullptr || (ex && ex->corr) || use_lists;
    if (uses_scratch) HIP_TRY(hipStreamWaitEvent(stream, c.ev_scr, 0));
    int32_t* ord = nullptr;
    if (use_lists) {
        if (c.ord.cap < bucket_list_bytes(cfg->W)) HIP_TRY(hipStreamSynchronize(stream));        // growing it frees the old one
        if (c.ord.ensure(bucket_list_bytes(cfg->W))) { set_err("workspace allocation failed (%zu B device)", bucket_list_bytes(cfg->W)); return HMCG_E_NOMEM; }
        ord = reinterpret_cast<int32_t*>(c.ord.base);
    }
    if (pl.bv) {
        HIP_TRY(hipFuncSetAttribute(pl.fptr(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.dyn));
        const size_t fbytes = (pl.scratch_bytes(cfg->W, cfg->K) + 255) & ~(size_t)255;
        const size_t sbytes = fbytes + pl.stream_bytes(cfg->W);
        if (c.scr.cap < sbytes) HIP_TRY(hipStreamSynchronize(stream));        // growing the scratch frees the old one
        if (c.scr.ensure(sbytes)) { set_err("workspace allocation failed (%zu B device)", sbytes); return HMCG_E_NOMEM; }
        p.fscr = reinterpret_cast<double*>(c.scr.base);
        if (pl.stream) { p.sscr = reinterpret_cast<uint8_t*>(c.scr.base + fbytes); p.stream_stride = (int64_t)pl.slab_bytes(); }
    }
    if (timing) HIP_TRY(hipEventRecord(c.ev0, stream));
#ifdef HMCG_STAMPS
    const size_t ndbg = (size_t)cfg->W * (pl.NT() / 64 + pl.NH()) * HMCG_NSTAMP_ALL;
    unsigned long long* ddbg = nullptr;
    HIP_TRY(hipMalloc((void**)&ddbg, ndbg * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ddbg, 0, ndbg * sizeof(unsigned long long), stream));
    p.dbg = ddbg;
#endif
    if (ord) { rc = build_bucket_lists(pl, dT, cfg->W, ord, stream); if (rc) return rc; }
    rc = launch_kernel(c, pl, p, stream, ord);
    if (rc) return rc;
    if (ex && ex->corr) {
        // correlations of the rounded draws (calccorr): one pass over the draw arrays while they are in HBM
        const size_t mbytes = sizeof(double) * (size_t)cfg->W * hmcg_host::moments_stride(cfg->K);
        if (c.mom.cap < mbytes) HIP_TRY(hipStreamSynchronize(stream));        // growing the table frees the old one
        if (c.mom.ensure(mbytes)) { set_err("workspace allocation failed (%zu B device)", mbytes); return HMCG_E_NOMEM; }
        hmcg_host::MomentsArgs ma{dmu, dsig2, dpi_end, dA, dfcast, reinterpret_cast<double*>(c.mom.base), cfg->nrun, cfg->nrun, cfg->W, cfg->K, cfg->H, true};
        HIP_TRY(hmcg_host::launch_moments(ma, stream));
        HIP_TRY(hmcg_host::launch_corr_finalize(reinterpret_cast<double*>(c.mom.base), ex->corr, cfg->W, cfg->K, stream));
    }
    if (uses_scratch) HIP_TRY(hipEventRecord(c.ev_scr, stream));
#ifdef HMCG_STAMPS
    rc = print_stamps(p, pl, ddbg, ndbg, stream);
    (void)hipFree(ddbg);
    if (rc) return rc;
#endif
    if (timing) {
        HIP_TRY(hipEventRecord(c.ev1, stream));
        HIP_TRY(hipEventSynchronize(c.ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, c.ev0, c.ev1));
        fill_timing(timing, pl, c, ms, 1, 0.0, cfg->W);
    }
    return 0;
}

// ---- host entry on one device --------------------------------------------------------------------------------

struct HostArrays {
    const double* Y; const int32_t* T; const double* yreal;
    double* mu; double* sig2; double* A; double* pi_end; double* fcast; double* summary; int32_t* status;
    const hmcg_extras* ex;
};

// Runs the n windows idx[0..n) (rows of the caller's arrays; idx == nullptr: rows 0..n-1) on device context c.
// Caller holds c.mu and has made c.device current.
int run_host_on_device(DeviceCtx& c, const hmcg_config* cfg, const int32_t* idx, int n, const HostArrays& h, hmcg_timing* timing)
{
    const auto t_call = std::chrono::steady_clock::now();
    // HMCG_TRACE=1 (diagnostics): host-side timeline of the call on stderr -- where the wall time beyond the kernels goes
    static const bool trace_on = diag_env("HMCG_TRACE") != nullptr;
    std::vector<std::pair<const char*, double>> trace;
    auto mark = [&](const char* what) {
        if (trace_on) trace.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    };
    // HMCG_FAIL_DEVICE=id (diagnostics): the host entry fails on that device id before it touches anything -- lets a test
    // see hmcg_estimate_batch_multi report one worker's error while the others complete
    if (const char* fe = diag_env("HMCG_FAIL_DEVICE")) {
        if (atoi(fe) == c.device) { set_err("injected failure (HMCG_FAIL_DEVICE=%d)", c.device); return HMCG_E_NOMEM; }
    }
    Plan pl;
    int minT = 0;                                                  // the shortest valid window of this device's share
    if (h.T) {
        for (int i = 0; i < n; ++i) {
            const int t = h.T[idx ? (size_t)idx[i] : (size_t)i];
            if (t >= 2 && (minT == 0 || t < minT)) minT = t;
        }
    }
    const HostLengths hl{h.T, idx, n};
    int rc = make_plan(cfg, h.ex, n, c.cu_count, minT, h.T ? &hl : nullptr, &pl);
    if (rc) return rc;
    if (!h.Y || !h.T) { set_err("Y and T are required"); return HMCG_E_BADARG; }
    const hmcg_extras* ex = h.ex;
    const size_t K = (size_t)cfg->K, ld = (size_t)cfg->ldY, H = (size_t)cfg->H, N = (size_t)n;
    const int n_samples = cfg->n_samples > 1 ? cfg->n_samples : 1;
    const long long nd_total = (long long)n_samples * cfg->nrun;       // kept draws per window over the whole run
    const size_t NS = 3 * K + K * K + 2 * H;
    const bool resume_in = (cfg->flags & HMCG_FLAG_RESUME) != 0;
    constexpr int32_t ST_SKIPPED = HMCG_ST_NONFINITE | HMCG_ST_BAD_T | HMCG_ST_BAD_RANGE;
    auto row = [&](int i) -> size_t { return idx ? (size_t)idx[i] : (size_t)i; };

    // per-draw output columns of one window, in the order they sit in a chunk buffer
    struct Col { double* host; size_t ncol; size_t off; };
    // (the sixth group, extras.pi_smooth_draws, is K * ldY columns wide: samples.pib[Nrun, N, D] of every window)
    Col cols[6] = { {h.mu, K, 0}, {h.sig2, K, 0}, {h.A, K * K, 0}, {h.pi_end, K, 0}, {h.fcast, 2 * H, 0},
                    {ex ? ex->pi_smooth_draws : nullptr, K * ld, 0} };
    size_t ncols = 0;
    const bool want_corr = ex && ex->corr;          // needs every draw column on the device, wanted by the caller or not
    bool copy_out = false;
    for (Col& cc : cols) {
        if (!(cc.host || (want_corr && &cc != &cols[5])) || nd_total == 0) cc.ncol = 0;
        if (cc.host && cc.ncol) copy_out = true;
        cc.off = ncols; ncols += cc.ncol;
    }
    const bool stream_draws = ncols > 0;
    const size_t NCC = (size_t)hmcg_host::corr_columns(cfg->K), mom_stride = hmcg_host::moments_stride(cfg->K);

    const int total_sweeps = n_samples * (cfg->burnin + cfg->nrun);
    int sb = cfg->sweep_base, se = total_sweeps;
    if (cfg->sweep_count > 0 && sb + cfg->sweep_count < se) se = sb + cfg->sweep_count;
    const int per = std::max(1, cfg->burnin + cfg->nrun);
    // chunk capacity: the ring of RING chunk buffers stays within ~1 GiB of device memory (and as much pinned memory)
    long long cap = nd_total > 0 ? nd_total : 1;
    if (stream_draws) {
        const long long budget = (1LL << 30) / RING / (long long)(8 * ncols * N);
        cap = std::max(1LL, std::min(cap, budget));
    }
    if (const char* cenv = diag_env("HMCG_CHUNK_DRAWS")) { const long long v = atoll(cenv); if (v > 0) cap = std::min(cap, v); }
    const bool one_chunk_env = diag_env("HMCG_NO_CHUNKS") != nullptr;           // diagnostics: one launch, as the device entry
    std::vector<Chunk> chunks = plan_chunks(sb, se, per, cfg->burnin, cfg->nrun, one_chunk_env ? (1LL << 40) : cap, stream_draws && !one_chunk_env,
                                            diag_env("HMCG_CHUNK_FLOOR_DIV"), diag_env("HMCG_CHUNK_KEEP"));
    long long chunk_max = 0;
    for (const Chunk& ch : chunks) chunk_max = std::max(chunk_max, ch.d1 - ch.d0);
    const bool chunked = chunks.size() > 1;

    // ---- layouts ----
    // Both arenas open with the same INPUT block (Y, T, ids, yreal, the optional per-window inputs: same order, same sizes), so
    // one H2D carries it; the device arena follows with the blocks a fresh (non-resumed) call needs zeroed, contiguous, so one
    // memset clears them; status and summary sit next to each other on both sides, so one D2H brings them back.  (Every
    // separate copy or memset is a node on the stream ahead of the first kernel: ten of them cost more than the 2 MB of Y.)
    Layout LD, LP;
    const bool need_ckpt = chunked || resume_in || (ex && (ex->xstate || ex->sumacc)) || se < total_sweeps;
    const bool want_xi = ex && ex->x_init, want_xf = ex && ex->x_final;
    const bool user_pif = ex && ex->pif_final, want_pif = user_pif || pl.needs_pif();     // scratch of the smoothing kernel
    const bool want_sm = ex && ex->pi_smooth_mean, want_fm = ex && ex->pi_filter_mean;
    const bool want_sv = ex && ex->sigvals && ex->nsave_ld > 0;
    const size_t nsv = want_sv ? (size_t)n_samples * (size_t)ex->nsave_ld : 0;
    const bool want_ss = ex && ex->sample_summary;
    const size_t nss = want_ss ? (size_t)n_samples * NS : 0;
    const bool has_yr = h.yreal && H, has_sr = ex && ex->sig_range, has_svr = ex && ex->save_range, has_ep = ex && ex->end_pos,
               has_ssg = ex && ex->sigma_signal;
    size_t o_dY, o_dT, o_dwid, o_dyr = 0, o_dxi = 0, o_dsr = 0, o_dsvr = 0, o_dep = 0, o_dss = 0;
    size_t o_pY, o_pT, o_pwid, o_pyr = 0, o_pxi = 0, o_psr = 0, o_psvr = 0, o_pep = 0, o_pss = 0;
    auto both = [&](size_t bytes, size_t& od, size_t& op) { od = LD.add(bytes); op = LP.add(bytes); };
    both(8 * N * ld, o_dY, o_pY); both(4 * N, o_dT, o_pT); both(4 * N, o_dwid, o_pwid);
    if (has_yr) both(8 * N * H, o_dyr, o_pyr);
    if (want_xi) both(4 * N * ld, o_dxi, o_pxi);
    if (has_sr) both(8 * N, o_dsr, o_psr);
    if (has_svr) both(8 * N, o_dsvr, o_psvr);
    if (has_ep) both(4 * N, o_dep, o_pep);
    if (has_ssg) both(8 * N, o_dss, o_pss);
    const size_t input_bytes = LD.total;                       // == LP.total: the block [0, input_bytes) of either arena
    // device: the zeroed block
    const size_t zero_begin = LD.total;
    const size_t o_dst = LD.add(4 * N);
    const size_t o_dsum = h.summary ? LD.add(8 * N * NS) : 0;
    const size_t o_dxs = need_ckpt ? LD.add(N * ld) : 0, o_dacc = need_ckpt ? LD.add(8 * N * (NS + K)) : 0;
    const size_t o_dxf = want_xf ? LD.add(4 * N * ld) : 0;
    const size_t o_dsv = want_sv ? LD.add(8 * N * nsv) : 0;
    const size_t o_dss2 = want_ss ? LD.add(8 * N * nss) : 0;
    const size_t o_dpif = want_pif ? LD.add(8 * N * ld * K) : 0;
    const size_t o_dsm = want_sm ? LD.add(8 * N * ld * K) : 0, o_dfm = want_fm ? LD.add(8 * N * ld * K) : 0;
    const size_t zero_bytes = LD.total - zero_begin;
    const size_t chunk_bytes = 8 * ncols * N * (size_t)chunk_max;
    size_t o_dchunk[RING] = {}, o_pchunk[RING] = {};
    const int nring = stream_draws ? (int)std::min<size_t>(RING, chunks.size()) : 0;
    for (int r = 0; r < nring; ++r) o_dchunk[r] = LD.add(chunk_bytes);
    const size_t o_dmom = want_corr ? LD.add(8 * N * mom_stride) : 0, o_dcorr = want_corr ? LD.add(8 * N * NCC * NCC) : 0;
    const bool use_lists = pl.nb > 1;
    const size_t o_dord = use_lists ? LD.add(bucket_list_bytes((int)N)) : 0;
    const size_t o_dfs = pl.bv ? LD.add(pl.scratch_bytes(n, cfg->K)) : 0;
    const size_t o_dstr = pl.stream ? LD.add(pl.stream_bytes(n)) : 0;
    // pinned staging beyond the input block: small outputs (status | summary adjacent, as on the device), chunk ring, extras
    const size_t o_pst = LP.add(4 * N);
    const size_t o_psum = h.summary ? LP.add(8 * N * NS) : 0;
    const size_t o_pst0 = LP.add(4 * N);      // status words as they stand after the first launch: which windows were skipped
    for (int r = 0; r < nring; ++r) o_pchunk[r] = LP.add(chunk_bytes);
    const size_t o_pxs = need_ckpt ? LP.add(N * ld) : 0, o_pacc = need_ckpt ? LP.add(8 * N * (NS + K)) : 0;
    const size_t o_pxf = want_xf ? LP.add(4 * N * ld) : 0;
    const size_t o_ppif = user_pif ? LP.add(8 * N * ld * K) : 0;
    const size_t o_psm = want_sm ? LP.add(8 * N * ld * K) : 0, o_pfm = want_fm ? LP.add(8 * N * ld * K) : 0;
    const size_t o_psv = want_sv ? LP.add(8 * N * nsv) : 0;
    const size_t o_pss2 = want_ss ? LP.add(8 * N * nss) : 0;
    const size_t o_pcorr = want_corr ? LP.add(8 * N * NCC * NCC) : 0;
    if (c.dev.ensure(LD.total) || c.pin.ensure(LP.total)) {
        set_err("workspace allocation failed (%zu B device, %zu B pinned)", LD.total, LP.total);
        return HMCG_E_NOMEM;
    }
    char* D = c.dev.base;
    char* P = c.pin.base;
    hipStream_t s = c.stream;
#define DP(T_, off) reinterpret_cast<T_*>(D + (off))
#define PP(T_, off) reinterpret_cast<T_*>(P + (off))

    // ---- pack the inputs (rows idx[i] of the caller's arrays) into pinned staging, one H2D each ----
    // (Y is the bulk, 2 MB at the headline shape: its first half sets out while the second half is packed)
#define H2D(doff, poff, bytes) HIP_TRY(hipMemcpyAsync(D + (doff), P + (poff), (bytes), hipMemcpyHostToDevice, s))
    const int n_early = ((size_t)8 * ld * n >= ((size_t)1 << 20) && o_pY == 0 && o_dY == 0 && !diag_env("HMCG_NO_EARLY_H2D")) ? n / 2 : 0;
    for (int i = 0; i < n; ++i) {
        const size_t g = row(i);
        if (i == n_early && n_early > 0) { mark("half packed"); H2D(0, 0, (size_t)8 * ld * n_early); mark("first half sent"); }
        memcpy(PP(double, o_pY) + (size_t)i * ld, h.Y + g * ld, 8 * ld);
        PP(int32_t, o_pT)[i] = h.T[g];
        PP(uint32_t, o_pwid)[i] = (ex && ex->window_ids) ? ex->window_ids[g] : cfg->window_base + (uint32_t)g;
        if (h.yreal && H) memcpy(PP(double, o_pyr) + (size_t)i * H, h.yreal + g * H, 8 * H);
        // (a RESUME call carries the caller's status words on; whether a window is skipped is for THIS call's kernel to say)
        if (resume_in) PP(int32_t, o_pst)[i] = h.status ? (h.status[g] & ~ST_SKIPPED) : 0;
        if (want_xi) memcpy(PP(int32_t, o_pxi) + (size_t)i * ld, ex->x_init + g * ld, 4 * ld);
        if (ex && ex->sig_range) { PP(int32_t, o_psr)[2 * i] = ex->sig_range[2 * g]; PP(int32_t, o_psr)[2 * i + 1] = ex->sig_range[2 * g + 1]; }
        if (ex && ex->save_range) { PP(int32_t, o_psvr)[2 * i] = ex->save_range[2 * g]; PP(int32_t, o_psvr)[2 * i + 1] = ex->save_range[2 * g + 1]; }
        if (ex && ex->end_pos) PP(int32_t, o_pep)[i] = ex->end_pos[g];
        if (ex && ex->sigma_signal) PP(double, o_pss)[i] = ex->sigma_signal[g];
        if (resume_in) {
            memcpy(PP(uint8_t, o_pxs) + (size_t)i * ld, ex->xstate + g * ld, ld);
            if (ex->sumacc) memcpy(PP(double, o_pacc) + (size_t)i * (NS + K), ex->sumacc + g * (NS + K), 8 * (NS + K));
            if (want_sm) memcpy(PP(double, o_psm) + (size_t)i * ld * K, ex->pi_smooth_mean + g * ld * K, 8 * ld * K);
            if (want_fm) memcpy(PP(double, o_pfm) + (size_t)i * ld * K, ex->pi_filter_mean + g * ld * K, 8 * ld * K);
            if (want_ss) memcpy(PP(double, o_pss2) + (size_t)i * nss, ex->sample_summary + g * nss, 8 * nss);
        }
    }
    mark("packed");
    {
        const size_t sent = (size_t)8 * ld * n_early;          // Y, T, ids, yreal and the optional per-window inputs
        H2D(sent, sent, input_bytes - sent);
    }
    if (resume_in) {
        H2D(o_dst, o_pst, 4 * N);
        H2D(o_dxs, o_pxs, N * ld);
        if (ex->sumacc) H2D(o_dacc, o_pacc, 8 * N * (NS + K)); else HIP_TRY(hipMemsetAsync(D + o_dacc, 0, 8 * N * (NS + K), s));
        if (want_sm) H2D(o_dsm, o_psm, 8 * N * ld * K);
        if (want_fm) H2D(o_dfm, o_pfm, 8 * N * ld * K);
        if (want_ss) H2D(o_dss2, o_pss2, 8 * N * nss);
        // outputs a skipped window never writes read as zero
        if (h.summary) HIP_TRY(hipMemsetAsync(D + o_dsum, 0, 8 * N * NS, s));
        if (want_xf) HIP_TRY(hipMemsetAsync(D + o_dxf, 0, 4 * N * ld, s));
        if (want_pif) HIP_TRY(hipMemsetAsync(D + o_dpif, 0, 8 * N * ld * K, s));
        if (want_sv) HIP_TRY(hipMemsetAsync(D + o_dsv, 0, 8 * N * nsv, s));
    } else {
        // status, summary, the checkpoint blocks (they live in the recycled arena: a skipped window writes none of them and
        // must not hand the caller an earlier call's bytes), x_final, sigvals, the per-sample summaries, pif, the running
        // smoothed / filtered sums: one memset
        HIP_TRY(hipMemsetAsync(D + zero_begin, 0, zero_bytes, s));
    }
#undef H2D
    hmcg_extras dex{};
    dex.struct_size = (int32_t)sizeof(hmcg_extras);
    dex.window_ids = DP(uint32_t, o_dwid);
    if (want_xi) dex.x_init = DP(int32_t, o_dxi);
    if (want_xf) dex.x_final = DP(int32_t, o_dxf);
    if (want_pif) dex.pif_final = DP(double, o_dpif);
    if (need_ckpt) { dex.xstate = DP(uint8_t, o_dxs); dex.sumacc = DP(double, o_dacc); }
    if (want_sm) dex.pi_smooth_mean = DP(double, o_dsm);
    if (want_fm) dex.pi_filter_mean = DP(double, o_dfm);
    if (ex && ex->sig_range) dex.sig_range = DP(int32_t, o_dsr);
    if (ex && ex->save_range) dex.save_range = DP(int32_t, o_dsvr);
    if (ex && ex->end_pos) dex.end_pos = DP(int32_t, o_dep);
    if (ex && ex->sigma_signal) dex.sigma_signal = DP(double, o_dss);
    if (want_sv) { dex.sigvals = DP(double, o_dsv); dex.nsave_ld = ex->nsave_ld; }
    if (want_ss) dex.sample_summary = DP(double, o_dss2);

    hmcg::KernelParams base = base_params(cfg, n, DP(double, o_dY), DP(int32_t, o_dT), (h.yreal && H) ? DP(double, o_dyr) : nullptr,
                                          DP(int32_t, o_dst), &dex, pl.use_sig);
    base.summary = h.summary ? DP(double, o_dsum) : nullptr;
    if (pl.bv) base.fscr = DP(double, o_dfs);
    if (pl.stream) { base.sscr = DP(uint8_t, o_dstr); base.stream_stride = (int64_t)pl.slab_bytes(); }
    if (pl.bv) HIP_TRY(hipFuncSetAttribute(pl.fptr(), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.dyn));

    // ---- the chunk pipeline ----
    // kernel c -> evk[c % RING] -> SDMA copy of its chunk buffer into pinned staging (copy stream) -> evc[c % RING]
    // -> host scatter into the caller's arrays.  A chunk buffer is [array][window][column][draw] with the chunk's own
    // draw count as leading dimension (one contiguous block: a plain 1-D copy, which the SDMA engines carry without
    // touching the CUs -- a helped sweep kernel leaves no registers for a blit kernel to run beside it).  Kernel
    // c + RING reuses both the device and the pinned buffer of chunk c: the host enqueues it only after it has waited
    // for copy c and scattered chunk c.
    if (use_lists) { rc = build_bucket_lists(pl, DP(int32_t, o_dT), n, DP(int32_t, o_dord), s); if (rc) return rc; }
    mark("inputs enqueued");
    const int nch = (int)chunks.size();
    double kernel_ms = 0.0;
    // per-chunk timing events, two per chunk: around the sweep kernel(s) alone -- the copy-out of the last chunk rides the
    // same stream behind its kernel and is not kernel time.  (Created only when timing is requested; released on every path.)
    struct TimingEvents {
        std::vector<hipEvent_t> ev;
        ~TimingEvents() { for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e); }
    } tev;
    if (timing) {
        tev.ev.assign(2 * (size_t)nch, nullptr);
        for (auto& e : tev.ev) HIP_TRY(hipEventCreate(&e));
    }
    auto scatter = [&](int cidx) {
        const Chunk& ch = chunks[cidx];
        const size_t ndc = (size_t)(ch.d1 - ch.d0);
        const double* src = PP(double, o_pchunk[cidx % RING]);
        if (!copy_out || ndc == 0) return;
        const std::function<void(int, int)> part = [&](int pi, int np) {          // windows [i0, i1) of this chunk
            const int i0 = (int)((long long)n * pi / np), i1 = (int)((long long)n * (pi + 1) / np);
            for (int i = i0; i < i1; ++i) {
                const size_t g = row(i);
                // a skipped window produced nothing -- its block of the (recycled) chunk buffer holds an earlier call's bytes:
                // the contract (hmcg.h) says its outputs read zero
                const bool skipped = (PP(int32_t, o_pst0)[i] & ST_SKIPPED) != 0;
                for (const Col& cc : cols) {
                    if (!cc.host) continue;
                    if (skipped) {
                        for (size_t q = 0; q < cc.ncol; ++q) memset(cc.host + (size_t)nd_total * (q + cc.ncol * g) + (size_t)ch.d0, 0, 8 * ndc);
                        continue;
                    }
                    const bool per_step = &cc == &cols[5];          // pi_smooth_draws: column = k * ldY + t; the kernel writes t < T[w] only
                    const size_t Tg = per_step ? (size_t)std::max(0, std::min((int)ld, (int)h.T[g])) : 0;
                    for (size_t q = 0; q < cc.ncol; ++q) {
                        double* dst = cc.host + (size_t)nd_total * (q + cc.ncol * g) + (size_t)ch.d0;
                        if (per_step && (q % ld) >= Tg) memset(dst, 0, 8 * ndc);              // beyond the window: reads zero
                        else memcpy(dst, src + ndc * (cc.off * N + q + cc.ncol * (size_t)i), 8 * ndc);
                    }
                }
            }
        };
        if (8 * ncols * N * ndc < ((size_t)1 << 20)) part(0, 1);                  // small chunks: not worth a hand-off
        else c.pool.run(part);
        mark("scattered");
    };
    // The copy of chunk c waits on the copy stream for the kernel's event (device-side hand-off); the host blocks on the copy's
    // event and scatters.  (HMCG_HOST_CHAIN, diagnostics: the host itself watches for the end of kernel c and then issues the
    // copy, polling the two events in turn -- same timeline, one spinning core; kept because it shows in the trace WHEN each
    // kernel was seen to end, which is how the late first copy of `profiles/r04/trace_host_entry_skip_words.txt` was found.)
    const bool device_chain = diag_env("HMCG_HOST_CHAIN") == nullptr;
    std::vector<char> self_issued((size_t)nch, 0);
    auto enqueue_kernel = [&](int cidx) -> int {
        const Chunk& ch = chunks[cidx];
        const int slot = cidx % RING;
        hmcg::KernelParams p = base;
        p.sweep_begin = ch.s0; p.sweep_end = ch.s1;
        p.resume = (resume_in || cidx > 0) ? 1 : 0;
        p.final_launch = (ch.s1 == total_sweeps) ? 1 : 0;
        const size_t ndc = (size_t)(ch.d1 - ch.d0);
        p.nd_ld = (int32_t)std::max<size_t>(ndc, 1); p.draw_off = (int32_t)ch.d0;
        // The LAST chunk's draws (1/32 of the run, 1.3 MB at the headline shape) are written by the kernel straight into the
        // pinned staging buffer: host memory the device addresses directly, complete at the end of the kernel -- there is no
        // copy behind the last kernel (it cost ~0.1 ms of the call's tail: nothing left to hide it behind).  Only the last:
        // a kernel that writes across the link runs 11 % slower (measured with every chunk direct).
        const bool direct_tail = stream_draws && copy_out && (cidx == nch - 1 || diag_env("HMCG_DIRECT_ALL")) && !want_corr && nch > 1 &&
                                 diag_env("HMCG_NO_TAIL_COPY") == nullptr && diag_env("HMCG_NO_DIRECT_TAIL") == nullptr;
        if (stream_draws) {
            double* cb = direct_tail ? PP(double, o_pchunk[slot]) : DP(double, o_dchunk[slot]);
            p.mu = cols[0].ncol ? cb + ndc * cols[0].off * N : nullptr;
            p.sig2 = cols[1].ncol ? cb + ndc * cols[1].off * N : nullptr;
            p.A = cols[2].ncol ? cb + ndc * cols[2].off * N : nullptr;
            p.pi_end = cols[3].ncol ? cb + ndc * cols[3].off * N : nullptr;
            p.fcast = cols[4].ncol ? cb + ndc * cols[4].off * N : nullptr;
            p.pi_smooth_draws = cols[5].ncol ? cb + ndc * cols[5].off * N : nullptr;
            // (a skipped window writes nothing into its block: the scatter zeroes its rows of the caller's arrays instead of
            //  copying them -- no memset node per chunk on the stream)
        }
        // skips are decided in the first launch's prologue, which notes them in the host's skip words as well (pinned,
        // zeroed here): the scatter knows which windows' blocks hold nothing once that kernel has ended
        if (cidx == 0 && stream_draws && copy_out) {
            memset(P + o_pst0, 0, 4 * N);
            p.skip_host = PP(int32_t, o_pst0);
        }
        if (timing) HIP_TRY(hipEventRecord(tev.ev[2 * (size_t)cidx], s));
        const int lrc = launch_kernel(c, pl, p, s, use_lists ? DP(int32_t, o_dord) : nullptr);
        if (lrc) return lrc;
        if (timing) HIP_TRY(hipEventRecord(tev.ev[2 * (size_t)cidx + 1], s));
        if (stream_draws) {
            // the last chunk's copy-out hides behind nothing: it goes on the compute stream itself, right behind its kernel
            // (or is no copy at all: direct_tail)
            const bool tail_copy = cidx == nch - 1 && !want_corr && diag_env("HMCG_NO_TAIL_COPY") == nullptr;
            if (tail_copy || direct_tail || device_chain) {
                hipStream_t cs = tail_copy ? s : c.copy;
                if (!tail_copy) {
                    HIP_TRY(hipEventRecord(c.evk[slot], s));
                    HIP_TRY(hipStreamWaitEvent(c.copy, c.evk[slot], 0));
                }
                if (ndc > 0 && copy_out && !direct_tail)
                    HIP_TRY(hipMemcpyAsync(P + o_pchunk[slot], D + o_dchunk[slot], 8 * ncols * N * ndc, hipMemcpyDeviceToHost, cs));
                HIP_TRY(hipEventRecord(c.evc[slot], cs));
                self_issued[(size_t)cidx] = 1;
            } else {
                HIP_TRY(hipEventRecord(c.evk[slot], s));
            }
            if (want_corr && ndc > 0) {
                // second moments of the chunk's rounded draws, in HBM, beside the chunk's copy-out (calccorr)
                hmcg_host::MomentsArgs ma{p.mu, p.sig2, p.pi_end, p.A, p.fcast, DP(double, o_dmom), (long long)ndc, (long long)ndc,
                                          n, cfg->K, cfg->H, ch.d0 == 0};
                HIP_TRY(hmcg_host::launch_moments(ma, s));
            }
        }
        return 0;
    };
    auto issue_copy = [&](int cidx) -> int {          // kernel cidx is known to be complete: no device-side wait
        const Chunk& ch = chunks[cidx];
        const int slot = cidx % RING;
        const size_t ndc = (size_t)(ch.d1 - ch.d0);
        // (an SDMA copy: 54 GB/s beside a running sweep kernel.  Nothing small may go ahead of it on this stream: a copy of
        //  a few KB is a shader copy in the HIP runtime and waits for a free CU, i.e. for the end of the NEXT sweep kernel)
        if (ndc > 0 && copy_out)
            HIP_TRY(hipMemcpyAsync(P + o_pchunk[slot], D + o_dchunk[slot], 8 * ncols * N * ndc, hipMemcpyDeviceToHost, c.copy));
        HIP_TRY(hipEventRecord(c.evc[slot], c.copy));
        return 0;
    };
    auto after_last_kernel = [&]() -> int {
        mark("kernels enqueued");
        if (want_corr) HIP_TRY(hmcg_host::launch_corr_finalize(DP(double, o_dmom), DP(double, o_dcorr), n, cfg->K, s));
        // ---- small outputs and one-off extras: D2H on the compute stream (after the last kernel) ----
#define D2H(poff, doff, bytes) HIP_TRY(hipMemcpyAsync(P + (poff), D + (doff), (bytes), hipMemcpyDeviceToHost, s))
        D2H(o_pst, o_dst, (h.summary ? o_dsum + 8 * N * NS : o_dst + 4 * N) - o_dst);        // status | summary, adjacent on both sides
        if (want_xf) D2H(o_pxf, o_dxf, 4 * N * ld);
        if (user_pif) D2H(o_ppif, o_dpif, 8 * N * ld * K);
        if (want_sm) D2H(o_psm, o_dsm, 8 * N * ld * K);
        if (want_fm) D2H(o_pfm, o_dfm, 8 * N * ld * K);
        if (want_sv) D2H(o_psv, o_dsv, 8 * N * nsv);
        if (want_ss) D2H(o_pss2, o_dss2, 8 * N * nss);
        if (want_corr) D2H(o_pcorr, o_dcorr, 8 * N * NCC * NCC);
        if (ex && ex->xstate) D2H(o_pxs, o_dxs, N * ld);
        if (ex && ex->sumacc) D2H(o_pacc, o_dacc, 8 * N * (NS + K));
#undef D2H
        return 0;
    };
    auto done = [&](hipEvent_t e, bool& yes) -> int {
        const hipError_t q = hipEventQuery(e);
        if (q == hipSuccess) { yes = true; return 0; }
        yes = false;
        if (q == hipErrorNotReady) return 0;
        set_err("hipEventQuery failed: %s (%s:%d)", hipGetErrorString(q), __FILE__, __LINE__);
        return (int)q;
    };
    {
        int enq = 0, iss = 0, sca = 0;
        while (enq < nch || (stream_draws && sca < nch)) {
            bool progress = false;
            while (enq < nch && (!stream_draws || enq < sca + RING)) {
                rc = enqueue_kernel(enq);
                if (rc) return rc;
                if (++enq == nch) { rc = after_last_kernel(); if (rc) return rc; }
                progress = true;
            }
            if (!stream_draws) break;
            if (iss < enq) {
                bool yes = self_issued[(size_t)iss] != 0;
                if (!yes) {
                    rc = done(c.evk[iss % RING], yes);
                    if (rc) return rc;
                    if (yes) { mark("kernel seen"); rc = issue_copy(iss); if (rc) return rc; }
                }
                if (yes) { ++iss; progress = true; }
            }
            if (sca < iss) {
                bool yes = false;
                rc = done(c.evc[sca % RING], yes);
                if (rc) return rc;
                if (yes) { mark("copy landed"); scatter(sca); ++sca; progress = true; }
            }
            if (!progress) {
                // nothing but the oldest outstanding copy to wait for: block on it; otherwise (host chain) keep polling
                if (sca < iss && iss == enq) HIP_TRY(hipEventSynchronize(c.evc[sca % RING]));
                else __builtin_ia32_pause();
            }
        }
    }
    mark("chunks scattered");
    HIP_TRY(hipStreamSynchronize(s));
    mark("stream idle");
    for (int i = 0; i < n; ++i) {
        const size_t g = row(i);
        // (a skipped window's per-draw rows were zeroed by the scatter, chunk by chunk; its correlation matrix here)
        if (want_corr && (PP(int32_t, o_pst)[i] & ST_SKIPPED)) memset(PP(double, o_pcorr) + (size_t)i * NCC * NCC, 0, 8 * NCC * NCC);
        if (h.status) h.status[g] = PP(int32_t, o_pst)[i];
        if (h.summary) memcpy(h.summary + g * NS, PP(double, o_psum) + (size_t)i * NS, 8 * NS);
        if (want_xf) memcpy(ex->x_final + g * ld, PP(int32_t, o_pxf) + (size_t)i * ld, 4 * ld);
        if (user_pif) memcpy(ex->pif_final + g * ld * K, PP(double, o_ppif) + (size_t)i * ld * K, 8 * ld * K);
        if (want_sm) memcpy(ex->pi_smooth_mean + g * ld * K, PP(double, o_psm) + (size_t)i * ld * K, 8 * ld * K);
        if (want_fm) memcpy(ex->pi_filter_mean + g * ld * K, PP(double, o_pfm) + (size_t)i * ld * K, 8 * ld * K);
        if (want_sv) memcpy(ex->sigvals + g * nsv, PP(double, o_psv) + (size_t)i * nsv, 8 * nsv);
        if (want_ss) memcpy(ex->sample_summary + g * nss, PP(double, o_pss2) + (size_t)i * nss, 8 * nss);
        if (want_corr) memcpy(ex->corr + g * NCC * NCC, PP(double, o_pcorr) + (size_t)i * NCC * NCC, 8 * NCC * NCC);
        if (ex && ex->xstate) memcpy(ex->xstate + g * ld, PP(uint8_t, o_pxs) + (size_t)i * ld, ld);
        if (ex && ex->sumacc) memcpy(ex->sumacc + g * (NS + K), PP(double, o_pacc) + (size_t)i * (NS + K), 8 * (NS + K));
    }
#undef DP
#undef PP
    mark("small outputs copied");
    if (trace_on) {
        fprintf(stderr, "[trace] device %d, %d windows, %d chunks:", c.device, n, nch);
        for (const auto& t : trace) fprintf(stderr, " %s %.3f |", t.first, t.second);
        if (timing) {
            fprintf(stderr, " chunk kernels (ms @ start after the first one's start):");
            for (int cidx = 0; cidx < nch; ++cidx) {
                float ms = 0.f, at = 0.f;
                (void)hipEventElapsedTime(&ms, tev.ev[2 * (size_t)cidx], tev.ev[2 * (size_t)cidx + 1]);
                (void)hipEventElapsedTime(&at, tev.ev[0], tev.ev[2 * (size_t)cidx]);
                fprintf(stderr, " %.3f@%.3f", ms, at);
            }
        }
        fprintf(stderr, "\n");
    }
    if (timing) {
        for (int cidx = 0; cidx < nch; ++cidx) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, tev.ev[2 * (size_t)cidx], tev.ev[2 * (size_t)cidx + 1]));
            kernel_ms += ms;       // the chunk's sweep kernel(s); a wait for a ring slot falls before the first event
        }
        const double call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
        fill_timing(timing, pl, c, kernel_ms, nch, call_ms, n);
    }
    return 0;
}

}  // namespace

extern "C" {

int hmcg_version(void) { return HMCG_VERSION; }

int hmcg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    if (n > 0 && virtual_devices() > 0) return virtual_devices();
    return n;
}

const char* hmcg_last_error(void) { return g_err; }

void hmcg_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_init_mu);
    for (DeviceCtx& c : g_ctx) {
        std::lock_guard<std::mutex> lc(c.mu);
        destroy_context(c);
    }
}

int hmcg_estimate_batch_device(const hmcg_config* cfg, const double* dY, const int32_t* dT, const double* dyreal,
                               double* dmu, double* dsig2, double* dA, double* dpi_end, double* dfcast,
                               double* dsummary, int32_t* dstatus, const hmcg_extras* dextras, void* stream,
                               hmcg_timing* timing)
{
    g_err[0] = 0;
    int rc = validate(cfg);
    if (rc) return rc;
    DeviceCtx* c = nullptr;
    rc = get_context(cfg->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->phys));
    hipStream_t s = stream ? (hipStream_t)stream : c->stream;
    return launch_device(*c, cfg, dY, dT, dyreal, dmu, dsig2, dA, dpi_end, dfcast, dsummary, dstatus, dextras, s, timing);
}

int hmcg_estimate_batch(const hmcg_config* cfg, const double* Y, const int32_t* T, const double* yreal, double* mu,
                        double* sig2, double* A, double* pi_end, double* fcast, double* summary, int32_t* status,
                        const hmcg_extras* extras, hmcg_timing* timing)
{
    g_err[0] = 0;
    int rc = validate(cfg);
    if (rc) return rc;
    DeviceCtx* c = nullptr;
    rc = get_context(cfg->device, &c);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(c->mu);
    HIP_TRY(hipSetDevice(c->phys));
    const HostArrays h{Y, T, yreal, mu, sig2, A, pi_end, fcast, summary, status, extras};
    return run_host_on_device(*c, cfg, nullptr, cfg->W, h, timing);
}

int hmcg_estimate_batch_multi(const hmcg_config* cfg, int32_t n_devices, const int32_t* device_ids, const double* Y,
                              const int32_t* T, const double* yreal, double* mu, double* sig2, double* A, double* pi_end,
                              double* fcast, double* summary, int32_t* status, const hmcg_extras* extras, hmcg_timing* timing)
{
    g_err[0] = 0;
    int rc = validate(cfg);
    if (rc) return rc;
    if (!Y || !T) { set_err("Y and T are required"); return HMCG_E_BADARG; }
    if (n_devices < 1 || n_devices > HMCG_MAXDEV) { set_err("n_devices %d out of range (1..%d)", n_devices, HMCG_MAXDEV); return HMCG_E_BADARG; }
    std::vector<int32_t> devs((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) devs[(size_t)i] = device_ids ? device_ids[i] : i;
    for (int i = 0; i < n_devices; ++i)
        for (int j = 0; j < i; ++j)
            if (devs[(size_t)i] == devs[(size_t)j]) { set_err("device %d listed twice", devs[(size_t)i]); return HMCG_E_BADARG; }
    const int G = std::min<int>(n_devices, cfg->W);                 // never more devices than windows
    const auto parts = partition_windows(T, cfg->W, G);
    const HostArrays h{Y, T, yreal, mu, sig2, A, pi_end, fcast, summary, status, extras};
    std::vector<int> rcs((size_t)G, 0);
    std::vector<std::string> errs((size_t)G);
    if (timing) memset(timing, 0, sizeof(hmcg_timing) * (size_t)n_devices);
    auto worker = [&](int r) {
        g_err[0] = 0;
        DeviceCtx* c = nullptr;
        int rr = get_context(devs[(size_t)r], &c);
        if (!rr) {
            std::lock_guard<std::mutex> lk(c->mu);
            hipError_t e = hipSetDevice(c->phys);
            if (e != hipSuccess) { set_err("hipSetDevice(%d) failed: %s", c->phys, hipGetErrorString(e)); rr = (int)e; }
            else rr = run_host_on_device(*c, cfg, parts[(size_t)r].data(), (int)parts[(size_t)r].size(), h, timing ? timing + r : nullptr);
        }
        rcs[(size_t)r] = rr;
        errs[(size_t)r] = g_err;           // thread-local message of this worker
    };
    if (G == 1) {
        worker(0);
    } else {
        std::vector<std::thread> th;
        th.reserve((size_t)G);
        for (int r = 0; r < G; ++r) th.emplace_back(worker, r);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < G; ++r)
        if (rcs[(size_t)r]) {
            set_err("device %d: %s", devs[(size_t)r], errs[(size_t)r].c_str());
            return rcs[(size_t)r];
        }
    return 0;
}

}  // extern "C"
