// hmcg.hip -- host side of libhmcgibbs.so (C ABI in include/hmcg.h).
// Owns the lazily created per-process context (stream, events), validates
// arguments, picks the kernel instantiation for (K, max_T, threads_per_window)
// and launches the persistent per-window sweep kernel.  No CPU compute path
// exists here: without a HIP device every compute entry returns HMCG_E_NODEVICE.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "variants.hpp"

namespace {

thread_local char g_err[512] = "";
std::mutex g_mu;

void set_err(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess) {                                                        \
            set_err("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                            \
        }                                                                              \
    } while (0)

struct Context {
    int device = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    int cu_count = 0;
};
Context g_ctx;

int ensure_context(int device)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_err("no HIP device available (libhmcgibbs has no CPU fallback)");
        return HMCG_E_NODEVICE;
    }
    if (device < 0 || device >= n) {
        set_err("device %d out of range (count %d)", device, n);
        return HMCG_E_BADARG;
    }
    HIP_TRY(hipSetDevice(device));
    if (g_ctx.device != device) {
        if (g_ctx.stream) { (void)hipStreamDestroy(g_ctx.stream); (void)hipEventDestroy(g_ctx.ev0); (void)hipEventDestroy(g_ctx.ev1); }
        g_ctx = Context{};
        HIP_TRY(hipStreamCreateWithFlags(&g_ctx.stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreate(&g_ctx.ev0));
        HIP_TRY(hipEventCreate(&g_ctx.ev1));
        HIP_TRY(hipDeviceGetAttribute(&g_ctx.cu_count, hipDeviceAttributeMultiprocessorCount, device));
        g_ctx.device = device;
    }
    return 0;
}

using namespace hmcg_host;
int flavour_of(const Variant& v) { return v.NH > 0 ? H : (v.occ == 2 ? P2 : P1); }
const VariantGroup* const g_groups[] = { &g_group_k2, &g_group_k3, &g_group_k4, &g_group_sig, &g_group_smooth };

constexpr size_t BIG_MAX_DYN_LDS = 144 * 1024;     // leaves room for the kernel's static LDS within 160 KiB

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
        cfg->H < 0 || cfg->H > HMCG_MAXH || cfg->max_T < 0 || cfg->max_T > cfg->ldY || cfg->sweep_base < 0 || cfg->sweep_count < 0 || cfg->n_samples < 0 || cfg->kappa < 0.0) {
        set_err("bad hmcg_config (W=%d K=%d ldY=%d max_T=%d burnin=%d nrun=%d H=%d)", cfg->W, cfg->K, cfg->ldY,
                cfg->max_T, cfg->burnin, cfg->nrun, cfg->H);
        return HMCG_E_BADARG;
    }
    for (int h = 0; h < cfg->H; ++h)
        if (cfg->horizons[h] < 0) { set_err("negative horizon"); return HMCG_E_BADARG; }
    return 0;
}

int launch_device(const hmcg_config* cfg, const double* dY, const int32_t* dT, const double* dyreal, double* dmu,
                  double* dsig2, double* dA, double* dpi_end, double* dfcast, double* dsummary, int32_t* dstatus,
                  const hmcg_extras* ex, hipStream_t stream, hmcg_timing* timing)
{
    if (!dY || !dT || !dstatus) { set_err("Y, T and status are required"); return HMCG_E_BADARG; }
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
    const bool use_smooth_req = ex && (ex->pi_smooth_mean != nullptr || ex->pi_filter_mean != nullptr);
    if ((use_sig || use_smooth_req) && cfg->K >= 5) { set_err("signal path and smoothed / filtered means: K <= 4 only"); return HMCG_E_UNSUPPORTED; }
    const Variant* v = nullptr;
    const BigVariant* bv = nullptr;
    int bigL = 0;
    size_t dyn = 0;
    const bool use_smooth = use_smooth_req;
    if (use_smooth && use_sig) { set_err("pi_smooth_mean / pi_filter_mean are not available on the signal path"); return HMCG_E_UNSUPPORTED; }
    // Flavour: helper waves pay off while every window has a CU to itself; with more windows than CUs the capped
    // plain variant lets two windows share a CU instead (a helped block takes the whole register file).
    // HMCG_FLAVOUR=p1|p2|h and HMCG_HELPERS=0|1 override the table (diagnostics, tools/variant_sweep.py).
    const bool small_batch = cfg->W <= g_ctx.cu_count;
    int force = -1;
    if (const char* henv = getenv("HMCG_HELPERS")) force = atoi(henv) != 0 ? H : (small_batch ? P1 : P2);
    if (const char* fenv = getenv("HMCG_FLAVOUR")) force = !strcmp(fenv, "h") ? H : (!strcmp(fenv, "p2") ? P2 : P1);
    if (cfg->K < 5) v = pick_variant(cfg->K, maxT, cfg->threads_per_window, use_sig, use_smooth, small_batch, force);
    if (!v && !use_sig && !use_smooth) {            // large K, or a window too long for the register-resident variants
        for (int i = 0; i < g_n_big_variants; ++i) if (g_big_variants[i].K == cfg->K) bv = &g_big_variants[i];
        if (bv) {
            bigL = (maxT + bv->NT - 1) / bv->NT;
            dyn = (size_t)bv->NT * bigL * (8 + 8 + 4 + 1) + 16;
            if (dyn > BIG_MAX_DYN_LDS || (cfg->threads_per_window != 0 && cfg->threads_per_window != bv->NT)) bv = nullptr;
        }
    }
    if (!v && !bv) {
        set_err("no kernel for K=%d max_T=%d threads_per_window=%d", cfg->K, maxT, cfg->threads_per_window);
        return HMCG_E_UNSUPPORTED;
    }
    const int vNT = v ? v->NT : bv->NT, vL = v ? v->L : bigL, vK = cfg->K;
    (void)vK;
    hmcg::KernelParams p{};
    p.Y = dY; p.T = dT; p.yreal = dyreal;
    p.ldY = cfg->ldY; p.W = cfg->W; p.H = cfg->H;
    p.sweep_begin = cfg->sweep_base;
    p.per_sample = cfg->burnin + cfg->nrun;
    p.burnin_s = cfg->burnin; p.nrun_s = cfg->nrun; p.n_samples = n_samples; p.nd = n_samples * cfg->nrun;
    if (p.per_sample < 1) p.per_sample = 1;
    const int total_sweeps = n_samples * (cfg->burnin + cfg->nrun);
    p.sweep_end = total_sweeps;
    if (cfg->sweep_count > 0 && cfg->sweep_base + cfg->sweep_count < p.sweep_end) p.sweep_end = cfg->sweep_base + cfg->sweep_count;
    p.resume = resume ? 1 : 0;
    p.final_launch = (p.sweep_end == total_sweeps) ? 1 : 0;
    p.kappa = cfg->kappa;
    if (ex) { p.pi_smooth_mean = ex->pi_smooth_mean; p.pi_filter_mean = ex->pi_filter_mean; }
    if (ex) { p.sig_range = ex->sig_range; p.save_range = ex->save_range; p.sigma_signal = ex->sigma_signal; p.sigvals = ex->sigvals; p.nsave_ld = ex->nsave_ld; }
    if (use_sig) { p.end_pos = ex->end_pos; p.blend_mask = cfg->blend_mask; }
    for (int h = 0; h < HMCG_MAXH; ++h) p.horizons[h] = h < cfg->H ? cfg->horizons[h] : 0;
    p.seed_lo = (uint32_t)cfg->seed; p.seed_hi = (uint32_t)(cfg->seed >> 32); p.window_base = cfg->window_base;
    p.alpha = cfg->alpha > 0.0 ? cfg->alpha : 1.0;
    p.nu = cfg->nu > 0.0 ? cfg->nu : 1.0;
    p.mu = dmu; p.sig2 = dsig2; p.A = dA; p.pi_end = dpi_end; p.fcast = dfcast; p.summary = dsummary;
    p.status = dstatus;
    if (ex) { p.x_init = ex->x_init; p.x_final = ex->x_final; p.pif_final = ex->pif_final; p.xstate = ex->xstate; p.sumacc = ex->sumacc; p.window_ids = ex->window_ids; }
    if (p.sweep_end < p.sweep_begin) { set_err("sweep_base beyond the run"); return HMCG_E_BADARG; }
    if (p.sigvals && p.nsave_ld < 1) { set_err("sigvals needs nsave_ld >= 1"); return HMCG_E_BADARG; }

    if (!resume) HIP_TRY(hipMemsetAsync(dstatus, 0, sizeof(int32_t) * (size_t)cfg->W, stream));
    if (bv) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bv->fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    if (timing) HIP_TRY(hipEventRecord(g_ctx.ev0, stream));
#ifdef HMCG_STAMPS
    const int nwv = vNT / 64 + (v ? v->NH : 0);
    const size_t ndbg = (size_t)cfg->W * nwv * HMCG_NSTAMP;
    unsigned long long* ddbg = nullptr;
    HIP_TRY(hipMalloc((void**)&ddbg, ndbg * sizeof(unsigned long long)));
    HIP_TRY(hipMemsetAsync(ddbg, 0, ndbg * sizeof(unsigned long long), stream));
    p.dbg = ddbg;
#endif
    if (v) hipLaunchKernelGGL(v->fn, dim3((unsigned)cfg->W), dim3((unsigned)(v->NT + 64 * v->NH)), 0, stream, p);
    else hipLaunchKernelGGL(bv->fn, dim3((unsigned)cfg->W), dim3((unsigned)bv->NT), dyn, stream, p, bigL);
    HIP_TRY(hipGetLastError());
#ifdef HMCG_STAMPS
    {
        static const char* names[HMCG_NSTAMP] = {"Ba wait", "param draws | shadow jobs", "Bb wait", "theta+ux+pdfs", "local product",
            "wave scan", "Bc wait", "prefix+replay+last", "Bd wait", "maps+compose", "map scan", "Be wait", "apply", "publish stats", "  (shadow: outputs)", "  (shadow: prep)", "  (stats: accumulate+N ballots)", "  (stats: wave sums)", "  (stats: pair ballots)", "unused"};
        std::vector<unsigned long long> h(ndbg);
        HIP_TRY(hipStreamSynchronize(stream));
        HIP_TRY(hipMemcpy(h.data(), ddbg, ndbg * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        (void)hipFree(ddbg);
        const int nsw = p.sweep_end - p.sweep_begin;
        fprintf(stderr, "[stamps] K=%d L=%d NT=%d W=%d sweeps=%d: mean cycles per sweep by wave (s_memtime ticks)\n", vK, vL, vNT, cfg->W, nsw);
        fprintf(stderr, "%-24s", "phase");
        for (int wv = 0; wv < nwv; ++wv) fprintf(stderr, "   wave%-2d", wv);
        fprintf(stderr, "\n");
        std::vector<double> tot(nwv, 0.0);
        for (int i = 0; i < HMCG_NSTAMP; ++i) {
            fprintf(stderr, "%-24s", names[i]);
            for (int wv = 0; wv < nwv; ++wv) {
                double acc = 0;
                for (int w = 0; w < cfg->W; ++w) acc += (double)h[((size_t)w * nwv + wv) * HMCG_NSTAMP + i];
                acc /= (double)cfg->W * (nsw > 0 ? nsw : 1);
                tot[wv] += acc;
                fprintf(stderr, " %8.0f", acc);
            }
            fprintf(stderr, "\n");
        }
        fprintf(stderr, "%-24s", "total");
        for (int wv = 0; wv < nwv; ++wv) fprintf(stderr, " %8.0f", tot[wv]);
        fprintf(stderr, "\n");
    }
#endif
    if (timing) {
        HIP_TRY(hipEventRecord(g_ctx.ev1, stream));
        HIP_TRY(hipEventSynchronize(g_ctx.ev1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, g_ctx.ev0, g_ctx.ev1));
        timing->kernel_ms = ms;
        timing->launches = 1;
        timing->threads_per_window = vNT;
        timing->steps_per_thread = vL;
        timing->helper_waves = v ? v->NH : 0;
        timing->reserved = 0;
        hipFuncAttributes fa{};
        const void* fptr = v ? reinterpret_cast<const void*>(v->fn) : reinterpret_cast<const void*>(bv->fn);
        if (hipFuncGetAttributes(&fa, fptr) == hipSuccess)
            timing->lds_bytes = (int32_t)(fa.sharedSizeBytes + dyn);
    }
    return 0;
}

template <typename T>
struct DevBuf {
    T* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) { return (int)hipMalloc((void**)&p, n * sizeof(T)); }
};

}  // namespace

extern "C" {

int hmcg_version(void) { return HMCG_VERSION; }

int hmcg_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* hmcg_last_error(void) { return g_err; }

void hmcg_shutdown(void)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_ctx.stream) {
        (void)hipStreamSynchronize(g_ctx.stream);
        (void)hipStreamDestroy(g_ctx.stream);
        (void)hipEventDestroy(g_ctx.ev0);
        (void)hipEventDestroy(g_ctx.ev1);
    }
    g_ctx = Context{};
}

int hmcg_estimate_batch_device(const hmcg_config* cfg, const double* dY, const int32_t* dT, const double* dyreal,
                               double* dmu, double* dsig2, double* dA, double* dpi_end, double* dfcast,
                               double* dsummary, int32_t* dstatus, const hmcg_extras* dextras, void* stream,
                               hmcg_timing* timing)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_err[0] = 0;
    int rc = validate(cfg);
    if (rc) return rc;
    rc = ensure_context(cfg->device);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : g_ctx.stream;
    return launch_device(cfg, dY, dT, dyreal, dmu, dsig2, dA, dpi_end, dfcast, dsummary, dstatus, dextras, s, timing);
}

int hmcg_estimate_batch(const hmcg_config* cfg, const double* Y, const int32_t* T, const double* yreal, double* mu,
                        double* sig2, double* A, double* pi_end, double* fcast, double* summary, int32_t* status,
                        const hmcg_extras* extras, hmcg_timing* timing)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_err[0] = 0;
    int rc = validate(cfg);
    if (rc) return rc;
    if (!Y || !T) { set_err("Y and T are required"); return HMCG_E_BADARG; }
    rc = ensure_context(cfg->device);
    if (rc) return rc;
    const size_t W = (size_t)cfg->W, K = (size_t)cfg->K, ld = (size_t)cfg->ldY, H = (size_t)cfg->H;
    const size_t nsmp = cfg->n_samples > 1 ? (size_t)cfg->n_samples : 1;
    const size_t nrun = nsmp * (size_t)cfg->nrun;        // kept draws per window
    const size_t NS = 3 * K + K * K + 2 * H;
    hipStream_t s = g_ctx.stream;
    DevBuf<double> dY, dyr, dmu, dsig, dA, dpe, dfc, dsum, dpif, dacc;
    DevBuf<int32_t> dT, dst, dxi, dxf;
    DevBuf<uint8_t> dxs;
    DevBuf<uint32_t> dwid;
    DevBuf<int32_t> dsr, dsv, dep;
    DevBuf<double> dss, dsvals, dsm, dfm;
#define ALLOC(buf, n) do { if ((buf).alloc(n) != 0) { set_err("hipMalloc of %zu elements failed", (size_t)(n)); return HMCG_E_NOMEM; } } while (0)
    ALLOC(dY, W * ld); ALLOC(dT, W); ALLOC(dst, W);
    HIP_TRY(hipMemcpyAsync(dY.p, Y, sizeof(double) * W * ld, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(dT.p, T, sizeof(int32_t) * W, hipMemcpyHostToDevice, s));
    if (yreal && H) { ALLOC(dyr, W * H); HIP_TRY(hipMemcpyAsync(dyr.p, yreal, sizeof(double) * W * H, hipMemcpyHostToDevice, s)); }
    // outputs start zeroed so that skipped windows (status HMCG_ST_NONFINITE / HMCG_ST_BAD_T) read as 0
#define ALLOC0(buf, n) do { ALLOC(buf, n); HIP_TRY(hipMemsetAsync((buf).p, 0, sizeof(double) * (size_t)(n), s)); } while (0)
    if (mu && nrun) ALLOC0(dmu, W * K * nrun);
    if (sig2 && nrun) ALLOC0(dsig, W * K * nrun);
    if (A && nrun) ALLOC0(dA, W * K * K * nrun);
    if (pi_end && nrun) ALLOC0(dpe, W * K * nrun);
    if (fcast && nrun && H) ALLOC0(dfc, W * 2 * H * nrun);
    if (summary) ALLOC0(dsum, W * NS);
#undef ALLOC0
    hmcg_extras dex{};
    dex.struct_size = (int32_t)sizeof(hmcg_extras);
    const bool resume = (cfg->flags & HMCG_FLAG_RESUME) != 0;
    if (extras) {
        if (extras->struct_size != (int32_t)sizeof(hmcg_extras)) { set_err("hmcg_extras.struct_size mismatch"); return HMCG_E_BADARG; }
        if (extras->x_init) { ALLOC(dxi, W * ld); HIP_TRY(hipMemcpyAsync(dxi.p, extras->x_init, sizeof(int32_t) * W * ld, hipMemcpyHostToDevice, s)); dex.x_init = dxi.p; }
        if (extras->x_final) { ALLOC(dxf, W * ld); HIP_TRY(hipMemsetAsync(dxf.p, 0, sizeof(int32_t) * W * ld, s)); dex.x_final = dxf.p; }
        if (extras->pif_final) { ALLOC(dpif, W * ld * K); HIP_TRY(hipMemsetAsync(dpif.p, 0, sizeof(double) * W * ld * K, s)); dex.pif_final = dpif.p; }
        if (extras->xstate) {
            ALLOC(dxs, W * ld);
            if (resume) HIP_TRY(hipMemcpyAsync(dxs.p, extras->xstate, W * ld, hipMemcpyHostToDevice, s));
            else HIP_TRY(hipMemsetAsync(dxs.p, 0, W * ld, s));
            dex.xstate = dxs.p;
        }
        if (extras->sig_range) { ALLOC(dsr, 2 * W); HIP_TRY(hipMemcpyAsync(dsr.p, extras->sig_range, sizeof(int32_t) * 2 * W, hipMemcpyHostToDevice, s)); dex.sig_range = dsr.p; }
        if (extras->save_range) { ALLOC(dsv, 2 * W); HIP_TRY(hipMemcpyAsync(dsv.p, extras->save_range, sizeof(int32_t) * 2 * W, hipMemcpyHostToDevice, s)); dex.save_range = dsv.p; }
        if (extras->end_pos) { ALLOC(dep, W); HIP_TRY(hipMemcpyAsync(dep.p, extras->end_pos, sizeof(int32_t) * W, hipMemcpyHostToDevice, s)); dex.end_pos = dep.p; }
        if (extras->sigma_signal) { ALLOC(dss, W); HIP_TRY(hipMemcpyAsync(dss.p, extras->sigma_signal, sizeof(double) * W, hipMemcpyHostToDevice, s)); dex.sigma_signal = dss.p; }
        if (extras->sigvals && extras->nsave_ld > 0) {
            ALLOC(dsvals, W * nsmp * (size_t)extras->nsave_ld);
            HIP_TRY(hipMemsetAsync(dsvals.p, 0, sizeof(double) * W * nsmp * (size_t)extras->nsave_ld, s));
            dex.sigvals = dsvals.p; dex.nsave_ld = extras->nsave_ld;
        }
        if (extras->pi_smooth_mean) {
            ALLOC(dsm, W * ld * K);
            if (resume) HIP_TRY(hipMemcpyAsync(dsm.p, extras->pi_smooth_mean, sizeof(double) * W * ld * K, hipMemcpyHostToDevice, s));
            else HIP_TRY(hipMemsetAsync(dsm.p, 0, sizeof(double) * W * ld * K, s));
            dex.pi_smooth_mean = dsm.p;
        }
        if (extras->pi_filter_mean) {
            ALLOC(dfm, W * ld * K);
            if (resume) HIP_TRY(hipMemcpyAsync(dfm.p, extras->pi_filter_mean, sizeof(double) * W * ld * K, hipMemcpyHostToDevice, s));
            else HIP_TRY(hipMemsetAsync(dfm.p, 0, sizeof(double) * W * ld * K, s));
            dex.pi_filter_mean = dfm.p;
        }
        if (extras->window_ids) { ALLOC(dwid, W); HIP_TRY(hipMemcpyAsync(dwid.p, extras->window_ids, sizeof(uint32_t) * W, hipMemcpyHostToDevice, s)); dex.window_ids = dwid.p; }
        if (extras->sumacc) {
            ALLOC(dacc, W * (NS + K));
            if (resume) HIP_TRY(hipMemcpyAsync(dacc.p, extras->sumacc, sizeof(double) * W * (NS + K), hipMemcpyHostToDevice, s));
            dex.sumacc = dacc.p;
        }
    }
    if (resume && status) HIP_TRY(hipMemcpyAsync(dst.p, status, sizeof(int32_t) * W, hipMemcpyHostToDevice, s));
    else if (resume) HIP_TRY(hipMemsetAsync(dst.p, 0, sizeof(int32_t) * W, s));
#undef ALLOC
    rc = launch_device(cfg, dY.p, dT.p, dyr.p, dmu.p, dsig.p, dA.p, dpe.p, dfc.p, dsum.p, dst.p, &dex, s, timing);
    if (rc) return rc;
#define D2H(dst_, src_, n) do { if ((dst_) && (src_)) HIP_TRY(hipMemcpyAsync((dst_), (src_), (n), hipMemcpyDeviceToHost, s)); } while (0)
    D2H(mu, dmu.p, sizeof(double) * W * K * nrun);
    D2H(sig2, dsig.p, sizeof(double) * W * K * nrun);
    D2H(A, dA.p, sizeof(double) * W * K * K * nrun);
    D2H(pi_end, dpe.p, sizeof(double) * W * K * nrun);
    D2H(fcast, dfc.p, sizeof(double) * W * 2 * H * nrun);
    D2H(summary, dsum.p, sizeof(double) * W * NS);
    D2H(status, dst.p, sizeof(int32_t) * W);
    if (extras) {
        D2H(extras->sigvals, dsvals.p, sizeof(double) * W * nsmp * (size_t)(extras->nsave_ld > 0 ? extras->nsave_ld : 0));
        D2H(extras->pi_smooth_mean, dsm.p, sizeof(double) * W * ld * K);
        D2H(extras->pi_filter_mean, dfm.p, sizeof(double) * W * ld * K);
        D2H(extras->x_final, dxf.p, sizeof(int32_t) * W * ld);
        D2H(extras->pif_final, dpif.p, sizeof(double) * W * ld * K);
        D2H(extras->xstate, dxs.p, W * ld);
        D2H(extras->sumacc, dacc.p, sizeof(double) * W * (NS + K));
    }
#undef D2H
    HIP_TRY(hipStreamSynchronize(s));
    return 0;
}

}  // extern "C"
