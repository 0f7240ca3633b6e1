// crl_api.hip -- the C ABI of include/crl.h: context, HBM allocations, launches.
//
// Host-side only; every entry point cites the reference interface it replaces in
// include/crl.h.  No torch types cross this boundary.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

#include "crl_internal.h"
#include "pong_device.h"

namespace crl {
void pong_gray_print_ticks();  // CRL_GRAY_DEBUG & 128
void launch_pong_gray_templates(const GrayParams &p, const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_rows, int band_chunks, uint8_t *band, uint8_t *rest,
                                hipStream_t st);
void launch_pong_raster_gray_ex(const GrayParams &p, const uint8_t *rest, int zero_row0, int zero_row1,
                                const uint8_t *x_first, const uint8_t *x_last, const uint8_t *y_first,
                                const uint8_t *y_last, int band_chunks, const uint8_t *tab_blob, const GrayTabOfs &tofs,
                                hipStream_t st);
}  // namespace crl

using namespace crl;

// Error text: per calling thread (crl_last_error: also covers crl_create, which has no context yet) and per context
// (crl_ctx_last_error).  Every entry point that takes a context names it first (CRL_ENTER); crl_fail then files the
// message under that context as well.
static thread_local std::string g_err;
static thread_local std::string *g_ctx_err = nullptr;

// entry points that take no crl_ctx (crl_policy_*, crl_frame_stack_update): their failures are the thread's, not those of
// whichever context the thread touched last (which another thread may have destroyed since)
void crl_fail_no_ctx(void) { g_ctx_err = nullptr; }

int crl_fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    if (g_ctx_err) *g_ctx_err = buf;
    return code;
}
#define fail crl_fail

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(CRL_EHIP, "%s: %s", #expr, hipGetErrorString(e_));       \
    } while (0)

struct AreaTab {
    std::vector<int32_t> ofs, si;  // ofs[d] .. ofs[d+1]: taps of output index d
    std::vector<float> alpha;
};

// cv2.resize INTER_AREA weight table for one axis (OpenCV computeResizeAreaTab; SURVEY
// C.3): table math in f64, weights stored as f32.
static AreaTab area_table(int ssize, int dsize) {
    AreaTab t;
    const double scale = 1.0 / ((double)dsize / (double)ssize);
    for (int d = 0; d < dsize; d++) {
        t.ofs.push_back((int32_t)t.si.size());
        const double f1 = d * scale, f2 = f1 + scale;
        const double cell = std::min(scale, ssize - f1);
        int s1 = (int)ceil(f1), s2 = (int)floor(f2);
        s2 = std::min(s2, ssize - 1);
        s1 = std::min(s1, s2);
        if (s1 - f1 > 1e-3) t.si.push_back(s1 - 1), t.alpha.push_back((float)((s1 - f1) / cell));
        for (int s = s1; s < s2; s++) t.si.push_back(s), t.alpha.push_back((float)(1.0 / cell));
        if (f2 - s2 > 1e-3) t.si.push_back(s2), t.alpha.push_back((float)(std::min(std::min(f2 - s2, 1.0), cell) / cell));
    }
    t.ofs.push_back((int32_t)t.si.size());
    return t;
}


struct crl_ctx {
    std::string err;  // text of this context's most recent failing call (crl_ctx_last_error)
    crl_opts o;
    int64_t n;
    PongSoA s{};
    ServeSrc src{};
    std::vector<void *> allocs;
    // raw mode
    uint8_t *atlas_rgb = nullptr;
    int ink_row0 = 0, ink_row1 = 0;
    // gray mode
    uint8_t *atlas_gray = nullptr, *band = nullptr, *rest = nullptr;
    uint8_t *x_first = nullptr, *x_last = nullptr, *y_first = nullptr, *y_last = nullptr;
    int32_t *xofs = nullptr, *yofs = nullptr, *xsi = nullptr, *ysi = nullptr;
    float *xalpha = nullptr, *yalpha = nullptr;
    int band_rows = 0, band_chunks = 0, zero_row0 = 0, zero_row1 = 0;
    uint8_t *tab_blob = nullptr;
    GrayTabOfs tofs{};
    uint8_t *tile_hdr = nullptr;  // [n * views * K] 64-byte tile headers of the address-linear gray writer
    float *f32_top = nullptr, *f32_bot = nullptr;  // CRL_OBS_F32_REF: court-without-objects tables (GrayParams)
    int f32_bot0 = 0, f32_xtaps = 0, f32_ytaps = 0, f32_map_row0 = 0, f32_map_rows = 0;
    // replay
    double *ru = nullptr;
    uint8_t *rbx = nullptr, *rby = nullptr;
    crl_timer tm;
    // crl_terminal_observation_dev: gathered descriptors (grows on demand, never shrinks)
    uint64_t *gather = nullptr;
    int64_t gather_cap = 0;
    int64_t *idx_dev = nullptr;  // staging for the host-index entry point
    int64_t idx_cap = 0;
    // action-containment flag (base_pong_env.py:42): host-mapped, written by the step kernel
    int32_t *bad_action_host = nullptr, *bad_action_dev = nullptr;
    std::vector<uint8_t> atlas_host;
    hipEvent_t flags_ev = nullptr;  // crl_set_flags_event: the caller's event, recorded behind the kernel that writes rewards and done flags
    crl_car_ctx *car = nullptr;  // set for CRL_ENV_CAR_DOUBLE contexts (everything above unused then)
};

template <class T>
static int dev_alloc(crl_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(count * sizeof(T), 16)));
    c->allocs.push_back(q);
    *p = (T *)q;
    return CRL_OK;
}

template <class T>
static int dev_upload(crl_ctx *c, T **p, const std::vector<T> &v, size_t pad_to = 0) {
    size_t cnt = std::max(v.size(), pad_to);
    int rc = dev_alloc(c, p, cnt);
    if (rc) return rc;
    HIP_TRY(hipMemset(*p, 0, std::max<size_t>(cnt * sizeof(T), 16)));
    HIP_TRY(hipMemcpy(*p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return CRL_OK;
}

// timing events: no system-scope fence when they complete (hip_runtime_api.h: "for events that are only being used to measure
// timing"): they bracket kernels on the launch stream and must not flush the caches in front of them
static const unsigned kTimerEvFlags = hipEventDefault | (getenv("CRL_EVENT_SYSTEM_FENCE") ? 0u : (unsigned)hipEventDisableSystemFence);

void crl_timer_begin(crl_timer *t, int which, hipStream_t st) {
    if (!t || !t->on) return;
    crl_event_pair p;
    if (!t->pool.empty()) {  // events are recycled by crl_kernel_time_ms: none is created inside a timed loop after the first pass
        p = t->pool.back();
        t->pool.pop_back();
    } else {
        hipEventCreateWithFlags(&p.a, kTimerEvFlags), hipEventCreateWithFlags(&p.b, kTimerEvFlags);
    }
    hipEventRecord(p.a, st);
    t->ev[which].push_back(p);
}
void crl_timer_end(crl_timer *t, int which, hipStream_t st) {
    if (!t || !t->on) return;
    hipEventRecord(t->ev[which].back().b, st);
}
static void begin_timed(crl_ctx *c, int which, hipStream_t st) { crl_timer_begin(&c->tm, which, st); }
static void end_timed(crl_ctx *c, int which, hipStream_t st) { crl_timer_end(&c->tm, which, st); }

static PongMode pong_mode(const crl_ctx *c) {
    PongMode m;
    m.wrapped = c->o.obs_mode == CRL_OBS_GRAY_RESIZED;
    m.single = c->o.env_kind == CRL_ENV_PONG_SINGLE;
    m.replicate = (c->o.flags & CRL_FLAG_STACK_REPLICATE) != 0;
    return m;
}
static int pong_views(const crl_ctx *c) { return c->o.env_kind == CRL_ENV_PONG_SINGLE ? 1 : 2; }

static int setup_gray(crl_ctx *c) {
    const int R = c->o.resized_dim;
    AreaTab xt = area_table(CRL_PONG_W, R), yt = area_table(CRL_PONG_H, R);
    std::vector<uint8_t> xf(CRL_PONG_W, 255), xl(CRL_PONG_W, 0), yf(CRL_PONG_H, 255), yl(CRL_PONG_H, 0);
    int band_rows = 0;
    for (int d = 0; d < R; d++) {
        for (int k = xt.ofs[d]; k < xt.ofs[d + 1]; k++) {
            xf[xt.si[k]] = std::min<uint8_t>(xf[xt.si[k]], (uint8_t)d), xl[xt.si[k]] = std::max<uint8_t>(xl[xt.si[k]], (uint8_t)d);
        }
        for (int k = yt.ofs[d]; k < yt.ofs[d + 1]; k++) {
            yf[yt.si[k]] = std::min<uint8_t>(yf[yt.si[k]], (uint8_t)d), yl[yt.si[k]] = std::max<uint8_t>(yl[yt.si[k]], (uint8_t)d);
            if (yt.si[k] < CRL_PONG_TOP) band_rows = d + 1;
        }
    }
    for (int i = 0; i < CRL_PONG_W; i++)
        if (xf[i] == 255) return fail(CRL_ESTATE, "source col %d feeds no output col", i);
    for (int i = 0; i < CRL_PONG_H; i++)
        if (yf[i] == 255) return fail(CRL_ESTATE, "source row %d feeds no output row", i);
    c->band_rows = band_rows;
    c->band_chunks = (band_rows * R + 15) / 16;
    // dense tables for the LDS-resident fast evaluator
    {
        GrayTabOfs o{};
        std::vector<uint8_t> blob;
        auto put = [&](const void *src, size_t bytes) {
            size_t at = (blob.size() + 15) & ~size_t(15);
            blob.resize(at + bytes);
            memcpy(blob.data() + at, src, bytes);
            return (int)at;
        };
        bool ok = true;
        int max_taps = 0;
        auto dense = [&](const AreaTab &t, std::vector<float> &a, std::vector<uint8_t> &s0, std::vector<uint8_t> &cnt) {
            a.assign((size_t)5 * R, 0.f), s0.assign(R, 0), cnt.assign(R, 0);
            for (int d = 0; d < R; d++) {
                const int k0 = t.ofs[d], k1 = t.ofs[d + 1];
                if (k1 - k0 > 5 || k1 <= k0) { ok = false; continue; }
                s0[d] = (uint8_t)t.si[k0], cnt[d] = (uint8_t)(k1 - k0);
                max_taps = std::max(max_taps, k1 - k0);
                for (int k = k0; k < k1; k++) {
                    if (t.si[k] != t.si[k0] + (k - k0)) ok = false;
                    a[(size_t)(k - k0) * R + d] = t.alpha[k];
                }
            }
        };
        std::vector<float> xa, ya;
        std::vector<uint8_t> xs0, xn, ys0, yn;
        dense(xt, xa, xs0, xn), dense(yt, ya, ys0, yn);
        // court rectangles touch output rows >= y_first[TOP]; their taps above the court must be ink-free
        for (int d = yf[CRL_PONG_TOP]; d < R; d++)
            for (int k = yt.ofs[d]; k < yt.ofs[d + 1]; k++)
                if (yt.si[k] < CRL_PONG_TOP && yt.si[k] < c->ink_row1) ok = false;
        // the x weights are stored pre-multiplied by 255 (the only non-zero source value): same single f32 multiply
        for (float &a : xa) a = 255.0f * a;
        o.xa = o.xa255 = put(xa.data(), xa.size() * 4), o.ya = put(ya.data(), ya.size() * 4);
        o.xs0 = put(xs0.data(), R), o.xn = put(xn.data(), R), o.ys0 = put(ys0.data(), R), o.yn = put(yn.data(), R);
        o.xf = put(xf.data(), xf.size()), o.xl = put(xl.data(), xl.size());
        o.yf = put(yf.data(), yf.size()), o.yl = put(yl.data(), yl.size());
        {
            // separable evaluator: 255*alpha, and the static bits of the per-row / per-col words
            // (bit 5t+2 / 5t+3: tap t lies in the left / right bat's columns; bit 5t+4: rows: tap t
            // is a white source row, cols: tap t exists)
            std::vector<uint32_t> rs(R, 0), cs(R, 0);
            for (int d = 0; d < R; d++) {
                for (int j = 0; j < yn[d] && j < 5; j++) {
                    const int r = ys0[d] + j;
                    if (r < CRL_PONG_TOP || r >= CRL_PONG_BOTTOM) rs[d] |= 16u << (5 * j);
                }
                for (int k = 0; k < xn[d] && k < 5; k++) {
                    const int cc = xs0[d] + k;
                    cs[d] |= 16u << (5 * k);
                    if (cc >= CRL_PONG_BATL_X && cc < CRL_PONG_BATL_X + CRL_PONG_BAT_W) cs[d] |= 4u << (5 * k);
                    if (cc >= CRL_PONG_BATR_X && cc < CRL_PONG_BATR_X + CRL_PONG_BAT_W) cs[d] |= 8u << (5 * k);
                }
            }
            o.rowstatic = put(rs.data(), rs.size() * 4), o.colstatic = put(cs.data(), cs.size() * 4);
        }
        blob.resize((blob.size() + 15) & ~size_t(15));
        o.total = (int)blob.size();
        o.fast_ok = ok ? 1 : 0;
        o.max_taps = max_taps;
        if (o.total > 6144) return fail(CRL_ESTATE, "tap tables (%d B) exceed the LDS budget", o.total);
        {
            std::vector<int32_t> b32;
            for (auto *v : {&xf, &xl, &yf, &yl})
                for (uint8_t e : *v) b32.push_back(e);
            o.box32 = put(b32.data(), b32.size() * 4);
            blob.resize((blob.size() + 15) & ~size_t(15));
        }
        int rc2 = dev_upload(c, &c->tab_blob, blob);
        if (rc2) return rc2;
        c->tofs = o;
    }
    const int chunks = (R * R + 15) / 16;
    int rc;
    if ((rc = dev_upload(c, &c->xofs, xt.ofs))) return rc;
    if ((rc = dev_upload(c, &c->yofs, yt.ofs))) return rc;
    if ((rc = dev_upload(c, &c->xsi, xt.si))) return rc;
    if ((rc = dev_upload(c, &c->ysi, yt.si))) return rc;
    if ((rc = dev_upload(c, &c->xalpha, xt.alpha))) return rc;
    if ((rc = dev_upload(c, &c->yalpha, yt.alpha))) return rc;
    if ((rc = dev_upload(c, &c->x_first, xf))) return rc;
    if ((rc = dev_upload(c, &c->x_last, xl))) return rc;
    if ((rc = dev_upload(c, &c->y_first, yf))) return rc;
    if ((rc = dev_upload(c, &c->y_last, yl))) return rc;
    if ((rc = dev_upload(c, &c->atlas_gray, c->atlas_host))) return rc;
    if ((rc = dev_alloc(c, &c->band, (size_t)3 * 484 * 2 * c->band_chunks * 16))) return rc;
    if ((rc = dev_alloc(c, &c->rest, (size_t)chunks * 16))) return rc;
    if ((rc = dev_alloc(c, &c->tile_hdr, (size_t)c->n * pong_views(c) * c->o.frame_stack * (64 + 8 + 2)))) return rc;
    HIP_TRY(hipMemset(c->rest, 0, (size_t)chunks * 16));
    GrayParams p{};
    p.R = R, p.K = c->o.frame_stack, p.atlas_gray = c->atlas_gray;
    p.xofs = c->xofs, p.yofs = c->yofs, p.xsi = c->xsi, p.ysi = c->ysi, p.xalpha = c->xalpha, p.yalpha = c->yalpha;
    launch_pong_gray_templates(p, c->x_first, c->x_last, c->y_first, c->y_last, c->band_rows, c->band_chunks, c->band,
                               c->rest, nullptr);
    HIP_TRY(hipGetLastError());
    if (c->o.obs_dtype == CRL_OBS_F32_REF) {  // the unrounded float32 path's tables: 484 score pairs x 3 kinds x 2 views x {unrounded, rounded}
        c->f32_bot0 = yf[CRL_PONG_BOTTOM], c->f32_xtaps = (int)xt.si.size(), c->f32_ytaps = (int)yt.si.size();
        c->f32_map_row0 = yf[CRL_PONG_TOP], c->f32_map_rows = yl[CRL_PONG_BOTTOM - 1] - yf[CRL_PONG_TOP] + 1;
        p.band_rows = c->band_rows, p.f32_bot0 = c->f32_bot0;
        // x 3 "kinds": both kept frames with this score pair | the left / the right score one higher in one of them (a point scored between them)
        if ((rc = dev_alloc(c, &c->f32_top, (size_t)484 * 3 * 4 * c->band_rows * R))) return rc;
        if ((rc = dev_alloc(c, &c->f32_bot, (size_t)2 * (R - c->f32_bot0) * R))) return rc;
        launch_pong_gray_f32ref_tables(p, c->f32_top, c->f32_bot, nullptr);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipDeviceSynchronize());
    // longest run of all-zero template rows below the score rows = the empty court
    std::vector<uint8_t> rest((size_t)R * R);
    HIP_TRY(hipMemcpy(rest.data(), c->rest, rest.size(), hipMemcpyDeviceToHost));
    int best0 = 0, best1 = 0, run0 = -1;
    for (int r = band_rows; r <= R; r++) {
        bool zero = r < R;
        if (zero)
            for (int x = 0; x < R; x++)
                if (rest[(size_t)r * R + x]) { zero = false; break; }
        if (zero && run0 < 0) run0 = r;
        if (!zero && run0 >= 0) {
            if (r - run0 > best1 - best0) best0 = run0, best1 = r;
            run0 = -1;
        }
    }
    c->zero_row0 = best0, c->zero_row1 = best1;
    return CRL_OK;
}

extern "C" {

#define CRL_ENTER(c) g_ctx_err = (c) ? &const_cast<crl_ctx *>(c)->err : nullptr
const char *crl_last_error(void) { return g_err.c_str(); }
const char *crl_ctx_last_error(const crl_ctx *c) { return c ? c->err.c_str() : g_err.c_str(); }
const char *crl_version(void) { return "crl-hip 0.1 (gfx950)"; }

int crl_create(const crl_opts *opts, const uint8_t *score_atlas_host, crl_ctx **out) {
    g_ctx_err = nullptr;
    if (!opts || !out) return fail(CRL_EINVAL, "null argument");
    const bool is_car = opts->env_kind == CRL_ENV_CAR_DOUBLE || opts->env_kind == CRL_ENV_CAR_SINGLE;
    if (!score_atlas_host && !is_car) return fail(CRL_EINVAL, "null argument");
    if (opts->env_kind != CRL_ENV_PONG_DOUBLE && opts->env_kind != CRL_ENV_CAR_DOUBLE && opts->env_kind != CRL_ENV_PONG_SINGLE &&
        opts->env_kind != CRL_ENV_CAR_SINGLE)
        return fail(CRL_EINVAL, "unknown env_kind %d", opts->env_kind);
    if (opts->num_envs <= 0) return fail(CRL_EINVAL, "num_envs must be positive");
    if (opts->reserved != 0) return fail(CRL_EINVAL, "crl_opts.reserved must be 0 (zero-initialise the struct)");
    if (opts->env_kind == CRL_ENV_CAR_DOUBLE || opts->env_kind == CRL_ENV_CAR_SINGLE) {
        if (opts->frame_stack < 0 || opts->frame_stack > 8) return fail(CRL_EINVAL, "frame_stack must be 1..8");
        if (opts->action_repeat < 0 || opts->action_repeat > 16) return fail(CRL_EINVAL, "action_repeat %d out of range (0..16)", opts->action_repeat);
        if (opts->done_policy != CRL_CAR_DONE_ANY && opts->done_policy != CRL_CAR_DONE_CAR0)
            return fail(CRL_EINVAL, "unknown done_policy %d", opts->done_policy);
        if (opts->done_policy == CRL_CAR_DONE_CAR0 && opts->env_kind != CRL_ENV_CAR_DOUBLE)
            return fail(CRL_EINVAL, "CRL_CAR_DONE_CAR0 needs a two-car context");
        if (opts->obs_dtype != CRL_OBS_U8) return fail(CRL_EINVAL, "CarRacing observations are uint8");
        int nd = 0;
        HIP_TRY(hipGetDeviceCount(&nd));
        if (opts->device < 0 || opts->device >= nd) return fail(CRL_EINVAL, "device %d of %d", opts->device, nd);
        HIP_TRY(hipSetDevice(opts->device));
        crl_ctx *cc = new crl_ctx();
        cc->o = *opts, cc->n = opts->num_envs;
        int rc = crl_car_create(opts, reinterpret_cast<const uint32_t *>(score_atlas_host), &cc->car);
        if (rc) { delete cc; return rc; }
        *out = cc;
        return CRL_OK;
    }
    if (opts->action_repeat != 0 || opts->done_policy != 0)
        return fail(CRL_EINVAL, "action_repeat / done_policy are CarRacing options (must be 0 for Pong)");
    if (opts->obs_dtype != CRL_OBS_U8 && !((opts->obs_dtype == CRL_OBS_F32 || opts->obs_dtype == CRL_OBS_F32_REF) && opts->obs_mode == CRL_OBS_GRAY_RESIZED))
        return fail(CRL_EINVAL, "obs_dtype %d unsupported for this obs_mode", opts->obs_dtype);
    if (opts->obs_mode == CRL_OBS_GRAY_RESIZED) {
        if (opts->resized_dim < 8 || opts->resized_dim > 84 || (opts->resized_dim * opts->resized_dim) % 4)
            return fail(CRL_EINVAL, "resized_dim %d unsupported (8..84, R*R %% 4 == 0)", opts->resized_dim);
        if (opts->frame_stack < 1 || opts->frame_stack > 4) return fail(CRL_EINVAL, "frame_stack must be 1..4");
    } else if (opts->obs_mode != CRL_OBS_RAW_RGB) {
        return fail(CRL_EINVAL, "unknown obs_mode %d", opts->obs_mode);
    }
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (opts->device < 0 || opts->device >= ndev) return fail(CRL_EINVAL, "device %d of %d", opts->device, ndev);
    HIP_TRY(hipSetDevice(opts->device));
    crl_ctx *c = new crl_ctx();
    c->o = *opts;
    const int64_t n = c->n = opts->num_envs;
    c->atlas_host.assign(score_atlas_host, score_atlas_host + CRL_PONG_ATLAS_BYTES);
    int rc = 0;
#define A(field, count) if (!rc) rc = dev_alloc(c, &c->s.field, (size_t)(count))
    A(speed_x, n); A(speed_y, n); A(ball_x, n); A(ball_y, n); A(bat_l, n); A(bat_r, n); A(score_l, n); A(score_r, n);
    A(rounds, n); A(steps, n); A(wrap_steps, n); A(serve_ctr, n); A(keep, 2 * n); A(ring, 8 * n); A(obs_frames, 2 * n);
    A(term_frames, 2 * n); A(real_reward, 2 * n); A(num_steps, n);
#undef A
    if (rc) { crl_destroy(c); return rc; }
    // zero state, BLANK kept frames (MaxAndSkipEnv._obs_buffer starts as zeros)
    {
        std::vector<uint64_t> blank((size_t)8 * n, kBlankFrame);
        hipError_t e = hipMemset(c->s.serve_ctr, 0, n * 4);
        if (e == hipSuccess) e = hipMemset(c->s.wrap_steps, 0, n * 4);
        if (e == hipSuccess) e = hipMemset(c->s.num_steps, 0, n * 4);
        if (e == hipSuccess) e = hipMemset(c->s.real_reward, 0, n * 8);
        if (e == hipSuccess) e = hipMemcpy(c->s.keep, blank.data(), 2 * n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->s.ring, blank.data(), 8 * n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->s.obs_frames, blank.data(), 2 * n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(c->s.term_frames, blank.data(), 2 * n * 8, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipHostMalloc((void **)&c->bad_action_host, 64, hipHostMallocMapped);
        if (e == hipSuccess) {
            *c->bad_action_host = 0;
            e = hipHostGetDevicePointer((void **)&c->bad_action_dev, c->bad_action_host, 0);
        }
        if (e != hipSuccess) {
            crl_destroy(c);
            return fail(CRL_EHIP, "create: state initialisation: %s", hipGetErrorString(e));
        }
        c->s.bad_action = c->bad_action_dev;
    }
    c->src.seed = opts->seed, c->src.env_id_base = opts->env_id_base;
    // score band: ink rows and the RGB-expanded copy used by the raw writer
    int r0 = CRL_PONG_TOP, r1 = 0;
    for (int sp = 0; sp < 484; sp++)
        for (int r = 0; r < CRL_PONG_TOP; r++)
            for (int x = 0; x < CRL_PONG_W; x++)
                if (c->atlas_host[((size_t)sp * CRL_PONG_TOP + r) * CRL_PONG_W + x] != 255) r0 = std::min(r0, r), r1 = std::max(r1, r + 1);
    if (r1 <= r0) r0 = r1 = 0;
    c->ink_row0 = r0, c->ink_row1 = r1;
    if (opts->obs_mode == CRL_OBS_RAW_RGB) {
        std::vector<uint8_t> rgb((size_t)CRL_PONG_ATLAS_BYTES * 3);
        for (size_t i = 0; i < (size_t)CRL_PONG_ATLAS_BYTES; i++) rgb[3 * i] = rgb[3 * i + 1] = rgb[3 * i + 2] = c->atlas_host[i];
        rc = dev_upload(c, &c->atlas_rgb, rgb);
    } else {
        rc = setup_gray(c);
    }
    if (rc) { crl_destroy(c); return rc; }
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) { crl_destroy(c); return fail(CRL_EHIP, "create: %s", hipGetErrorString(e)); }
    *out = c;
    return CRL_OK;
}

void crl_destroy(crl_ctx *c) {
    g_ctx_err = nullptr;
    if (!c) return;
    hipSetDevice(c->o.device);
    hipDeviceSynchronize();
#ifdef CRL_ABLATION
    if (getenv("CRL_GRAY_DEBUG") && (atoi(getenv("CRL_GRAY_DEBUG")) & 128)) crl::pong_gray_print_ticks();
#endif
    if (c->car) crl_car_destroy(c->car);
    for (void *p : c->allocs) hipFree(p);
    for (int w = 0; w < kTimerSlots; w++)
        for (auto &p : c->tm.ev[w]) hipEventDestroy(p.a), hipEventDestroy(p.b);
    for (auto &p : c->tm.pool) hipEventDestroy(p.a), hipEventDestroy(p.b);
    if (c->ru) hipFree(c->ru);
    if (c->rbx) hipFree(c->rbx);
    if (c->rby) hipFree(c->rby);
    if (c->gather) hipFree(c->gather);
    if (c->idx_dev) hipFree(c->idx_dev);
    if (c->bad_action_host) hipHostFree(c->bad_action_host);
    delete c;
}

int crl_seed(crl_ctx *c, uint64_t seed) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    if (c->car) crl_car_seed(c->car, seed);
    c->src.seed = seed;
    return CRL_OK;
}

// crl_stack_desc -> what the gray launch needs; refuses what the fused draw cannot do (the caller then keeps crl_frame_stack_update)
static int stack_of(const crl_ctx *c, const crl_stack_desc *sd, const uint8_t *obs_dev, GrayStack *out) {
    *out = GrayStack{};
    if (!sd) return CRL_OK;
    if (c->car || c->o.obs_mode != CRL_OBS_GRAY_RESIZED) return fail(CRL_ESTATE, "a fused frame stack needs a GRAY_RESIZED Pong context");
    if (c->o.flags & CRL_FLAG_STACK_REPLICATE)
        return fail(CRL_ESTATE, "a fused frame stack follows FrameStackTensor's zero-on-done history; this context keeps the FrameStack wrapper's (CRL_FLAG_STACK_REPLICATE)");
    if (!sd->stack_dev || sd->reserved != 0) return fail(CRL_EINVAL, "crl_stack_desc: null stack / reserved must be 0");
    if (sd->planes < 1 || sd->planes > 4) return fail(CRL_EINVAL, "crl_stack_desc.planes %d: the context keeps the descriptors of the last 4 planes", sd->planes);
    if (sd->agent < 0 || sd->agent >= pong_views(c)) return fail(CRL_EINVAL, "crl_stack_desc.agent %d of %d", sd->agent, pong_views(c));
    if (sd->valid_planes < 0) return fail(CRL_EINVAL, "crl_stack_desc.valid_planes %d", sd->valid_planes);
    const int want_f32 = c->o.obs_dtype != CRL_OBS_U8;
    if (sd->dtype != CRL_OBS_U8 && sd->dtype != CRL_OBS_F32) return fail(CRL_EINVAL, "crl_stack_desc.dtype %d (CRL_OBS_U8 or CRL_OBS_F32)", sd->dtype);
    if (want_f32 && sd->dtype != CRL_OBS_F32) return fail(CRL_EINVAL, "a float32 context's stack is float32");
    if (sd->alias_newest) {
        if (c->o.frame_stack != 1) return fail(CRL_EINVAL, "alias_newest needs a context with frame_stack 1 (the observation IS the stack's newest plane)");
        if ((sd->dtype == CRL_OBS_F32) != (want_f32 != 0)) return fail(CRL_EINVAL, "alias_newest needs the stack and the observation in one element type");
    }
    if ((uintptr_t)sd->stack_dev % 16) return fail(CRL_EINVAL, "crl_stack_desc.stack_dev must be 16-byte aligned");
    (void)obs_dev;
    out->out = reinterpret_cast<uint8_t *>(sd->stack_dev), out->k = sd->planes, out->view = sd->agent, out->f32 = sd->dtype == CRL_OBS_F32;
    out->valid = std::min(sd->valid_planes, sd->planes), out->alias = sd->alias_newest ? 1 : 0;
    return CRL_OK;
}

static int draw_obs(crl_ctx *c, uint8_t *obs_dev, hipStream_t st, const GrayStack *sk = nullptr) {
    if (!obs_dev && !(sk && sk->out)) return CRL_OK;
    begin_timed(c, 1, st);
    if (c->o.obs_mode == CRL_OBS_RAW_RGB) {
        launch_pong_raster_raw(c->s.obs_frames, c->n, c->atlas_rgb, c->ink_row0, c->ink_row1, obs_dev, pong_views(c), st);
    } else {
        GrayParams p{};
        p.ring = c->s.ring, p.n = c->n, p.R = c->o.resized_dim, p.K = c->o.frame_stack, p.views = pong_views(c);
        p.atlas_gray = c->atlas_gray, p.band = c->band, p.band_rows = c->band_rows;
        p.xofs = c->xofs, p.yofs = c->yofs, p.xsi = c->xsi, p.ysi = c->ysi, p.xalpha = c->xalpha, p.yalpha = c->yalpha;
        p.obs = obs_dev, p.obs_f32 = c->o.obs_dtype, p.hdr = c->tile_hdr;
        p.f32_top = c->f32_top, p.f32_bot = c->f32_bot, p.f32_bot0 = c->f32_bot0, p.f32_xtaps = c->f32_xtaps, p.f32_ytaps = c->f32_ytaps;
        p.f32_map_row0 = c->f32_map_row0, p.f32_map_rows = c->f32_map_rows;
        if (sk) p.stack = *sk;
        launch_pong_raster_gray_ex(p, c->rest, c->zero_row0, c->zero_row1, c->x_first, c->x_last, c->y_first, c->y_last,
                                   c->band_chunks, c->tab_blob, c->tofs, st);
    }
    end_timed(c, 1, st);
    HIP_TRY(hipGetLastError());
    return CRL_OK;
}

// The step kernel raises *bad_action_host (host-mapped memory) when it meets an action outside {0, 1, 2, 999}; reading
// it here costs no synchronisation -- a set flag belongs to an EARLIER step.
static int pending_action_error(crl_ctx *c, bool clear) {
    if (!c->bad_action_host) return CRL_OK;
    const int32_t v = *(volatile int32_t *)c->bad_action_host;
    if (!v) return CRL_OK;
    if (clear) *(volatile int32_t *)c->bad_action_host = 0;
    return fail(CRL_EACTION, "a Pong action outside {0, 1, 2, %d} was passed to an earlier crl_step (first seen: %d); "
                "the reference asserts action_space.contains(action) (pong/base_pong_env.py:42)", CRL_PONG_CHEAT, v - 1);
}

int crl_check(crl_ctx *c, void *stream) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return pending_action_error(c, true);
}

int crl_reset(crl_ctx *c, uint8_t *obs_dev, void *stream) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    hipStream_t st = (hipStream_t)stream;
    if (c->car) return crl_car_reset(c->car, obs_dev, st);
    if (int rc = pending_action_error(c, true)) return rc;  // reported ONCE, by the first call that sees it: the next call proceeds
    launch_pong_reset(c->s, c->src, c->n, pong_mode(c), st);
    HIP_TRY(hipGetLastError());
    return draw_obs(c, obs_dev, st);
}

int crl_render(crl_ctx *c, uint8_t *obs_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || !obs_dev) return fail(CRL_EINVAL, "null argument");
    hipStream_t st = (hipStream_t)stream;
    if (c->car) return crl_car_render(c->car, obs_dev, st);
    return draw_obs(c, obs_dev, st);
}

int crl_step(crl_ctx *c, const void *actions_void, uint8_t *obs_dev, float *rew_dev, uint8_t *done_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || !actions_void) return fail(CRL_EINVAL, "null ctx/actions");
    hipStream_t st = (hipStream_t)stream;
    if (c->car) return crl_car_step(c->car, (const float *)actions_void, obs_dev, rew_dev, done_dev, st, &c->tm);
    if (int rc = pending_action_error(c, true)) return rc;  // (reported once; this call did no work)
    const int32_t *actions_dev = (const int32_t *)actions_void;
    begin_timed(c, 0, st);
    launch_pong_dynamics(c->s, c->src, actions_dev, c->n, pong_mode(c), rew_dev, done_dev, st);
    end_timed(c, 0, st);
    HIP_TRY(hipGetLastError());
    if (c->flags_ev) HIP_TRY(hipEventRecord(c->flags_ev, st));
    return draw_obs(c, obs_dev, st);
}

int crl_step_stack(crl_ctx *c, const void *actions_void, uint8_t *obs_dev, float *rew_dev, uint8_t *done_dev, const crl_stack_desc *stack,
                   void *stream) {
    CRL_ENTER(c);
    if (!c || !actions_void) return fail(CRL_EINVAL, "null ctx/actions");
    if (!stack) return crl_step(c, actions_void, obs_dev, rew_dev, done_dev, stream);
    GrayStack sk;
    if (int rc = stack_of(c, stack, obs_dev, &sk)) return rc;  // (before anything is stepped: a refused call has done no work)
    hipStream_t st = (hipStream_t)stream;
    if (int rc = pending_action_error(c, true)) return rc;
    begin_timed(c, 0, st);
    launch_pong_dynamics(c->s, c->src, (const int32_t *)actions_void, c->n, pong_mode(c), rew_dev, done_dev, st);
    end_timed(c, 0, st);
    HIP_TRY(hipGetLastError());
    if (c->flags_ev) HIP_TRY(hipEventRecord(c->flags_ev, st));
    return draw_obs(c, obs_dev, st, &sk);
}

int crl_set_flags_event(crl_ctx *c, void *event) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    if (c->car) return fail(CRL_ESTATE, "crl_set_flags_event is a Pong entry point");
    c->flags_ev = (hipEvent_t)event;
    return CRL_OK;
}

int crl_draw_stack(crl_ctx *c, uint8_t *obs_dev, const crl_stack_desc *stack, void *stream) {
    CRL_ENTER(c);
    if (!c || !stack) return fail(CRL_EINVAL, "null argument");
    GrayStack sk;
    if (int rc = stack_of(c, stack, obs_dev, &sk)) return rc;
    return draw_obs(c, obs_dev, (hipStream_t)stream, &sk);
}

int crl_info(crl_ctx *c, const float **real_reward_dev, const int32_t **num_steps_dev) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    if (c->car) return fail(CRL_ESTATE, "crl_info is a Pong entry point");
    if (real_reward_dev) *real_reward_dev = c->s.real_reward;
    if (num_steps_dev) *num_steps_dev = c->s.num_steps;
    return CRL_OK;
}

int crl_copy_info(crl_ctx *c, float *rr_out, int32_t *ns_out, void *stream) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    if (c->car) return fail(CRL_ESTATE, "crl_copy_info is a Pong entry point");
    hipStream_t st = (hipStream_t)stream;
    if (rr_out) HIP_TRY(hipMemcpyAsync(rr_out, c->s.real_reward, (size_t)c->n * 8, hipMemcpyDeviceToDevice, st));
    if (ns_out) HIP_TRY(hipMemcpyAsync(ns_out, c->s.num_steps, (size_t)c->n * 4, hipMemcpyDeviceToDevice, st));
    return CRL_OK;
}

int crl_car_info(crl_ctx *c, const uint8_t **done_car_dev, const int32_t **num_steps_dev) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    if (done_car_dev) *done_car_dev = crl_car_done_flags(c->car);
    if (num_steps_dev) *num_steps_dev = crl_car_info_steps(c->car);
    return CRL_OK;
}

int crl_car_copy_info(crl_ctx *c, uint8_t *done_car_out, int32_t *num_steps_out, int32_t *elapsed_out, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = crl_car_players(c->car);
    if (done_car_out) HIP_TRY(hipMemcpyAsync(done_car_out, crl_car_done_flags(c->car), (size_t)(c->n * P), hipMemcpyDeviceToDevice, st));
    if (num_steps_out) HIP_TRY(hipMemcpyAsync(num_steps_out, crl_car_info_steps(c->car), (size_t)c->n * 4, hipMemcpyDeviceToDevice, st));
    if (elapsed_out) HIP_TRY(hipMemcpyAsync(elapsed_out, crl_car_info_elapsed(c->car), (size_t)c->n * 4, hipMemcpyDeviceToDevice, st));
    return CRL_OK;
}

int64_t crl_obs_bytes_per_env(const crl_ctx *c) {
    CRL_ENTER(c);
    if (!c) return 0;
    if (c->car) return crl_car_obs_bytes(c->car);
    if (c->o.obs_mode == CRL_OBS_RAW_RGB) return pong_views(c) * (int64_t)CRL_PONG_FRAME_BYTES;
    return pong_views(c) * (int64_t)c->o.frame_stack * c->o.resized_dim * c->o.resized_dim * (c->o.obs_dtype != CRL_OBS_U8 ? 4 : 1);
}

// Draws `m` frame pairs that already sit in device memory as a single-plane ring ([8][m], planes 0..2 unused).
static int render_ring(crl_ctx *c, const uint64_t *ring_dev, int64_t m, uint8_t *out_dev, hipStream_t st) {
    if (m == 0) return CRL_OK;
    if (c->o.obs_mode == CRL_OBS_RAW_RGB) {
        launch_pong_raster_raw(ring_dev + 6 * m, m, c->atlas_rgb, c->ink_row0, c->ink_row1, out_dev, pong_views(c), st);
    } else {
        GrayParams p{};
        p.ring = ring_dev, p.n = m, p.R = c->o.resized_dim, p.K = 1, p.views = pong_views(c);
        p.atlas_gray = c->atlas_gray, p.band = c->band, p.band_rows = c->band_rows;
        p.xofs = c->xofs, p.yofs = c->yofs, p.xsi = c->xsi, p.ysi = c->ysi, p.xalpha = c->xalpha, p.yalpha = c->yalpha;
        p.obs = out_dev, p.obs_f32 = c->o.obs_dtype;
        p.f32_top = c->f32_top, p.f32_bot = c->f32_bot, p.f32_bot0 = c->f32_bot0, p.f32_xtaps = c->f32_xtaps, p.f32_ytaps = c->f32_ytaps;
        p.f32_map_row0 = c->f32_map_row0, p.f32_map_rows = c->f32_map_rows;
        launch_pong_raster_gray_ex(p, c->rest, c->zero_row0, c->zero_row1, c->x_first, c->x_last, c->y_first, c->y_last,
                                   c->band_chunks, c->tab_blob, c->tofs, st);
    }
    HIP_TRY(hipGetLastError());
    return CRL_OK;
}

static int ensure_gather(crl_ctx *c, int64_t m) {
    if (m <= c->gather_cap) return CRL_OK;
    HIP_TRY(hipDeviceSynchronize());  // growing is rare: earlier launches may still read the old scratch
    if (c->gather) hipFree(c->gather), c->gather = nullptr, c->gather_cap = 0;
    const int64_t cap = std::max<int64_t>(m, 1024);
    HIP_TRY(hipMalloc((void **)&c->gather, (size_t)8 * cap * 8));
    c->gather_cap = cap;
    return CRL_OK;
}

// Renders `count` frame pairs given on the host (crl_render_raw).
static int render_pairs(crl_ctx *c, const std::vector<uint64_t> &f0, const std::vector<uint64_t> &f1, uint8_t *out_dev,
                        hipStream_t st) {
    const int64_t m = (int64_t)f0.size();
    if (m == 0) return CRL_OK;
    if (int rc = ensure_gather(c, m)) return rc;
    std::vector<uint64_t> ring((size_t)8 * m, kBlankFrame);
    std::copy(f0.begin(), f0.end(), ring.begin() + 6 * m);
    std::copy(f1.begin(), f1.end(), ring.begin() + 7 * m);
    HIP_TRY(hipMemcpyAsync(c->gather, ring.data(), ring.size() * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // `ring` is a local
    return render_ring(c, c->gather, m, out_dev, st);
}

int crl_terminal_observation_dev(crl_ctx *c, const int64_t *env_idx_dev, int64_t count, uint8_t *out_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || (count > 0 && (!env_idx_dev || !out_dev))) return fail(CRL_EINVAL, "null argument");
    if (count <= 0) return CRL_OK;
    hipStream_t st = (hipStream_t)stream;
    if (c->car) {
        // CarRacing keeps the finished envs' last frames (players, 96, 96): one gather kernel
        const int64_t tile = (int64_t)crl_car_players(c->car) * 96 * 96;
        crl::launch_car_gather_frames(crl_car_terminal_frames(c->car), env_idx_dev, count, c->n, tile, out_dev, st);
        HIP_TRY(hipGetLastError());
        return CRL_OK;
    }
    if (int rc = ensure_gather(c, count)) return rc;
    launch_pong_gather_frames(c->s.term_frames, env_idx_dev, count, c->n, c->gather, st);
    return render_ring(c, c->gather, count, out_dev, st);
}

int crl_terminal_observation(crl_ctx *c, const int64_t *env_idx_host, int64_t count, uint8_t *out_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || (count > 0 && (!env_idx_host || !out_dev))) return fail(CRL_EINVAL, "null argument");
    if (count <= 0) return CRL_OK;
    for (int64_t k = 0; k < count; k++)
        if (env_idx_host[k] < 0 || env_idx_host[k] >= c->n) return fail(CRL_EINVAL, "env index %lld out of range", (long long)env_idx_host[k]);
    hipStream_t st = (hipStream_t)stream;
    if (count > c->idx_cap) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->idx_dev) hipFree(c->idx_dev), c->idx_dev = nullptr, c->idx_cap = 0;
        const int64_t cap = std::max<int64_t>(count, 1024);
        HIP_TRY(hipMalloc((void **)&c->idx_dev, (size_t)cap * 8));
        c->idx_cap = cap;
    }
    HIP_TRY(hipMemcpyAsync(c->idx_dev, env_idx_host, (size_t)count * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));  // the caller's index array may be reused on return
    return crl_terminal_observation_dev(c, c->idx_dev, count, out_dev, stream);
}

// The frame descriptors the CURRENT observation was drawn from, in the ring layout uint64 [8][N] (plane p of the stack, kept
// frame s -> row 2p + s; raw contexts and K = 1 use the newest plane, rows 6 and 7): 64 bytes per env instead of its pixels.
int crl_obs_descriptors(crl_ctx *c, crl_pong_frame *desc_out_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || !desc_out_dev) return fail(CRL_EINVAL, "null argument");
    if (c->car) return fail(CRL_ESTATE, "crl_obs_descriptors is a Pong entry point");
    hipStream_t st = (hipStream_t)stream;
    uint64_t *out = reinterpret_cast<uint64_t *>(desc_out_dev);
    if (c->o.obs_mode == CRL_OBS_RAW_RGB) {
        HIP_TRY(hipMemsetAsync(out, 0xFF, (size_t)6 * c->n * 8, st));  // (blank planes: never read)
        HIP_TRY(hipMemcpyAsync(out + 6 * c->n, c->s.obs_frames, (size_t)2 * c->n * 8, hipMemcpyDeviceToDevice, st));
    } else {
        HIP_TRY(hipMemcpyAsync(out, c->s.ring, (size_t)8 * c->n * 8, hipMemcpyDeviceToDevice, st));
    }
    return CRL_OK;
}

// Draws `count` observations from descriptors in that layout (uint64 [8][count]) -- any envs, e.g. another shard's: out_dev as
// crl_step's obs_dev for `count` envs.  A frame is a pure function of its descriptors, so a rank that holds every shard's
// descriptors holds every shard's observation (BASELINE config #5 without moving pixels).
int crl_render_frames_dev(crl_ctx *c, const crl_pong_frame *desc_dev, int64_t count, uint8_t *out_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || (count > 0 && (!desc_dev || !out_dev))) return fail(CRL_EINVAL, "null argument");
    if (c->car) return fail(CRL_ESTATE, "crl_render_frames_dev is a Pong entry point");
    if (count <= 0) return CRL_OK;
    hipStream_t st = (hipStream_t)stream;
    const uint64_t *ring = reinterpret_cast<const uint64_t *>(desc_dev);
    if (c->o.obs_mode == CRL_OBS_RAW_RGB) {
        launch_pong_raster_raw(ring + 6 * count, count, c->atlas_rgb, c->ink_row0, c->ink_row1, out_dev, pong_views(c), st);
    } else {
        GrayParams p{};
        p.ring = ring, p.n = count, p.R = c->o.resized_dim, p.K = c->o.frame_stack, p.views = pong_views(c);
        p.atlas_gray = c->atlas_gray, p.band = c->band, p.band_rows = c->band_rows;
        p.xofs = c->xofs, p.yofs = c->yofs, p.xsi = c->xsi, p.ysi = c->ysi, p.xalpha = c->xalpha, p.yalpha = c->yalpha;
        p.obs = out_dev, p.obs_f32 = c->o.obs_dtype;
        p.f32_top = c->f32_top, p.f32_bot = c->f32_bot, p.f32_bot0 = c->f32_bot0, p.f32_xtaps = c->f32_xtaps, p.f32_ytaps = c->f32_ytaps;
        p.f32_map_row0 = c->f32_map_row0, p.f32_map_rows = c->f32_map_rows;
        launch_pong_raster_gray_ex(p, c->rest, c->zero_row0, c->zero_row1, c->x_first, c->x_last, c->y_first, c->y_last,
                                   c->band_chunks, c->tab_blob, c->tofs, st);
    }
    HIP_TRY(hipGetLastError());
    return CRL_OK;
}

int crl_render_raw(crl_ctx *c, const crl_pong_frame *frames_host, int64_t count, uint8_t *out_dev, void *stream) {
    CRL_ENTER(c);
    if (!c || !frames_host || !out_dev) return fail(CRL_EINVAL, "null argument");
    if (c->car) return fail(CRL_ESTATE, "crl_render_raw is a Pong entry point");
    if (c->o.obs_mode != CRL_OBS_RAW_RGB) return fail(CRL_ESTATE, "crl_render_raw needs a RAW_RGB context");
    std::vector<uint64_t> f(count);
    memcpy(f.data(), frames_host, (size_t)count * 8);
    return render_pairs(c, f, f, out_dev, (hipStream_t)stream);
}

}  // extern "C"

// ---- state exchange: device SoA <-> host AoS
template <class T>
static int d2h(std::vector<T> &v, const T *dev, int64_t first, int64_t count, hipStream_t st) {
    v.resize(count);
    HIP_TRY(hipMemcpyAsync(v.data(), dev + first, count * sizeof(T), hipMemcpyDeviceToHost, st));
    return CRL_OK;
}
template <class T>
static int h2d(const std::vector<T> &v, T *dev, int64_t first, hipStream_t st) {
    HIP_TRY(hipMemcpyAsync(dev + first, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, st));
    return CRL_OK;
}

static crl_pong_frame to_frame(uint64_t u) {
    crl_pong_frame f;
    memcpy(&f, &u, 8);
    return f;
}
static uint64_t from_frame(const crl_pong_frame &f) {
    uint64_t u;
    memcpy(&u, &f, 8);
    return u;
}

extern "C" {

int crl_get_state(crl_ctx *c, crl_pong_env_state *out, int64_t first, int64_t count, void *stream) {
    CRL_ENTER(c);
    if (!c || !out || first < 0 || count < 0 || first + count > c->n) return fail(CRL_EINVAL, "bad range");
    if (c->car) return fail(CRL_ESTATE, "use crl_car_get_state for CarRacing contexts");
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> sx, sy;
    std::vector<int32_t> bx, by, bl, br, sl, sr, ro, stp, ws;
    std::vector<uint32_t> sc;
    std::vector<uint64_t> keep[2], ring[8];
    int rc = 0;
    rc |= d2h(sx, c->s.speed_x, first, count, st), rc |= d2h(sy, c->s.speed_y, first, count, st);
    rc |= d2h(bx, c->s.ball_x, first, count, st), rc |= d2h(by, c->s.ball_y, first, count, st);
    rc |= d2h(bl, c->s.bat_l, first, count, st), rc |= d2h(br, c->s.bat_r, first, count, st);
    rc |= d2h(sl, c->s.score_l, first, count, st), rc |= d2h(sr, c->s.score_r, first, count, st);
    rc |= d2h(ro, c->s.rounds, first, count, st), rc |= d2h(stp, c->s.steps, first, count, st);
    rc |= d2h(ws, c->s.wrap_steps, first, count, st), rc |= d2h(sc, c->s.serve_ctr, first, count, st);
    for (int k = 0; k < 2; k++) rc |= d2h(keep[k], c->s.keep + k * c->n, first, count, st);
    for (int k = 0; k < 8; k++) rc |= d2h(ring[k], c->s.ring + k * c->n, first, count, st);
    if (rc) return CRL_EHIP;
    HIP_TRY(hipStreamSynchronize(st));
    for (int64_t i = 0; i < count; i++) {
        crl_pong_env_state &e = out[i];
        e.speed_x = sx[i], e.speed_y = sy[i], e.ball_x = bx[i], e.ball_y = by[i], e.bat_l_y = bl[i], e.bat_r_y = br[i];
        e.score_l = sl[i], e.score_r = sr[i], e.num_rounds = ro[i], e.num_steps = stp[i];
        e.serve_ctr = sc[i], e.wrap_steps = ws[i];
        e.keep[0] = to_frame(keep[0][i]), e.keep[1] = to_frame(keep[1][i]);
        // exchange format: the 3 most recent planes, oldest first = ring planes 1..3
        for (int h = 0; h < 3; h++) e.hist[h][0] = to_frame(ring[2 * (h + 1)][i]), e.hist[h][1] = to_frame(ring[2 * (h + 1) + 1][i]);
    }
    return CRL_OK;
}

int crl_set_state(crl_ctx *c, const crl_pong_env_state *in, int64_t first, int64_t count, void *stream) {
    CRL_ENTER(c);
    if (!c || !in || first < 0 || count < 0 || first + count > c->n) return fail(CRL_EINVAL, "bad range");
    if (c->car) return fail(CRL_ESTATE, "use crl_car_set_state for CarRacing contexts");
    hipStream_t st = (hipStream_t)stream;
    std::vector<double> sx(count), sy(count);
    std::vector<int32_t> bx(count), by(count), bl(count), br(count), sl(count), sr(count), ro(count), stp(count), ws(count);
    std::vector<uint32_t> sc(count);
    std::vector<uint64_t> keep[2], ring[8], of[2];
    for (auto &v : keep) v.resize(count);
    for (auto &v : ring) v.resize(count);
    for (auto &v : of) v.resize(count);
    for (int64_t i = 0; i < count; i++) {
        const crl_pong_env_state &e = in[i];
        sx[i] = e.speed_x, sy[i] = e.speed_y, bx[i] = e.ball_x, by[i] = e.ball_y, bl[i] = e.bat_l_y, br[i] = e.bat_r_y;
        sl[i] = e.score_l, sr[i] = e.score_r, ro[i] = e.num_rounds, stp[i] = e.num_steps, sc[i] = e.serve_ctr, ws[i] = e.wrap_steps;
        keep[0][i] = from_frame(e.keep[0]), keep[1][i] = from_frame(e.keep[1]);
        ring[0][i] = ring[1][i] = kBlankFrame;  // the plane that the next roll drops
        for (int h = 0; h < 3; h++) ring[2 * (h + 1)][i] = from_frame(e.hist[h][0]), ring[2 * (h + 1) + 1][i] = from_frame(e.hist[h][1]);
        of[0][i] = ring[6][i], of[1][i] = ring[7][i];
    }
    int rc = 0;
    rc |= h2d(sx, c->s.speed_x, first, st), rc |= h2d(sy, c->s.speed_y, first, st);
    rc |= h2d(bx, c->s.ball_x, first, st), rc |= h2d(by, c->s.ball_y, first, st);
    rc |= h2d(bl, c->s.bat_l, first, st), rc |= h2d(br, c->s.bat_r, first, st);
    rc |= h2d(sl, c->s.score_l, first, st), rc |= h2d(sr, c->s.score_r, first, st);
    rc |= h2d(ro, c->s.rounds, first, st), rc |= h2d(stp, c->s.steps, first, st);
    rc |= h2d(ws, c->s.wrap_steps, first, st), rc |= h2d(sc, c->s.serve_ctr, first, st);
    for (int k = 0; k < 2; k++) rc |= h2d(keep[k], c->s.keep + k * c->n, first, st);
    for (int k = 0; k < 8; k++) rc |= h2d(ring[k], c->s.ring + k * c->n, first, st);
    if (c->o.obs_mode == CRL_OBS_GRAY_RESIZED)
        for (int k = 0; k < 2; k++) rc |= h2d(of[k], c->s.obs_frames + k * c->n, first, st);
    if (rc) return CRL_EHIP;
    HIP_TRY(hipStreamSynchronize(st));
    return CRL_OK;
}

int crl_set_replay(crl_ctx *c, const double *u, const uint8_t *bx, const uint8_t *by, int64_t per_env) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    if (c->car) return fail(CRL_ESTATE, "use crl_car_set_replay for CarRacing contexts");
    HIP_TRY(hipDeviceSynchronize());
    if (c->ru) hipFree(c->ru), hipFree(c->rbx), hipFree(c->rby), c->ru = nullptr, c->rbx = c->rby = nullptr;
    c->src.ru = nullptr, c->src.rbx = c->src.rby = nullptr, c->src.per_env = 0;
    if (per_env <= 0) return CRL_OK;
    if (!u || !bx || !by) return fail(CRL_EINVAL, "null replay arrays");
    const size_t m = (size_t)c->n * per_env;
    HIP_TRY(hipMalloc((void **)&c->ru, m * 8));
    HIP_TRY(hipMalloc((void **)&c->rbx, m));
    HIP_TRY(hipMalloc((void **)&c->rby, m));
    HIP_TRY(hipMemcpy(c->ru, u, m * 8, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->rbx, bx, m, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->rby, by, m, hipMemcpyHostToDevice));
    c->src.ru = c->ru, c->src.rbx = c->rbx, c->src.rby = c->rby, c->src.per_env = per_env;
    return CRL_OK;
}

int crl_kernel_timing(crl_ctx *c, int enable) {
    CRL_ENTER(c);
    if (!c) return fail(CRL_EINVAL, "null ctx");
    c->tm.on = enable != 0;
    // the event pairs of the first few hundred timed launches are created HERE, not inside the caller's timed loop
    // (later ones come back through the pool when crl_kernel_time_ms reads them out)
    if (c->tm.on && c->tm.pool.empty() && c->tm.ev[0].empty() && c->tm.ev[1].empty()) {
        for (int i = 0; i < 512; i++) {
            crl_event_pair p;
            if (hipEventCreateWithFlags(&p.a, kTimerEvFlags) != hipSuccess) break;
            if (hipEventCreateWithFlags(&p.b, kTimerEvFlags) != hipSuccess) {
                hipEventDestroy(p.a);
                break;
            }
            c->tm.pool.push_back(p);
        }
    }
    return CRL_OK;
}

int crl_kernel_time_stats(crl_ctx *c, int which, double *total_ms, int64_t *launches, double *max_ms) {
    CRL_ENTER(c);
    if (!c || which < 0 || which >= kTimerSlots) return fail(CRL_EINVAL, "bad argument");
    crl_timer &t = c->tm;
    for (auto &p : t.ev[which]) {
        HIP_TRY(hipEventSynchronize(p.b));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        t.ms[which] += ms, t.cnt[which]++;
        t.max_ms[which] = std::max(t.max_ms[which], (double)ms);
        t.pool.push_back(p);
    }
    t.ev[which].clear();
    if (total_ms) *total_ms = t.ms[which];
    if (launches) *launches = t.cnt[which];
    if (max_ms) *max_ms = t.max_ms[which];
    t.ms[which] = 0, t.cnt[which] = 0, t.max_ms[which] = 0;
    return CRL_OK;
}

int crl_kernel_time_ms(crl_ctx *c, int which, double *total_ms, int64_t *launches) {
    return crl_kernel_time_stats(c, which, total_ms, launches, nullptr);
}

int crl_car_get_state(crl_ctx *c, crl_car_env_state *out, int64_t first, int64_t count, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car || !out) return fail(CRL_EINVAL, "not a CarRacing context / null argument");
    return crl_car_get_state_impl(c->car, out, first, count, (hipStream_t)stream);
}
int crl_car_set_state(crl_ctx *c, const crl_car_env_state *in, int64_t first, int64_t count, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car || !in) return fail(CRL_EINVAL, "not a CarRacing context / null argument");
    return crl_car_set_state_impl(c->car, in, first, count, (hipStream_t)stream);
}
int crl_car_get_track(crl_ctx *c, int64_t env, int32_t *n, float *tile_poly, float *border_poly, uint8_t *border,
                      float *start_pose, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    return crl_car_get_track_impl(c->car, env, n, tile_poly, border_poly, border, start_pose, (hipStream_t)stream);
}
int crl_car_set_track(crl_ctx *c, int64_t env, int32_t n, const double *tile_poly, const double *border_poly, const uint8_t *border,
                      const float *start_pose, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    return crl_car_set_track_impl(c->car, env, n, tile_poly, border_poly, border, start_pose, (hipStream_t)stream);
}
int crl_car_cap_hits(crl_ctx *c, int32_t *out4_host, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car || !out4_host) return fail(CRL_EINVAL, "not a CarRacing context / null argument");
    return crl_car_cap_hits_impl(c->car, out4_host, (hipStream_t)stream);
}
int crl_car_get_map(crl_ctx *c, int64_t env, uint8_t *palette_host, int32_t *overflow, void *stream) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    return crl_car_get_map_impl(c->car, env, palette_host, overflow, (hipStream_t)stream);
}
int crl_car_set_replay(crl_ctx *c, const double *u, const uint8_t *swap, int64_t attempts) {
    CRL_ENTER(c);
    if (!c || !c->car) return fail(CRL_EINVAL, "not a CarRacing context");
    return crl_car_set_replay_impl(c->car, u, swap, attempts);
}

}  // extern "C"
