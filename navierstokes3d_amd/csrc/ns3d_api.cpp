// ns3d_api.cpp — the extern "C" boundary of libns3d.so (see include/ns3d.h for the contract and the
// reference file:line each entry point replaces).  Host-only logic: argument checks, mode dispatch
// (strict / fast kernels), the context (stream, reduction scratch, ping-pong buffer) and the
// pseudo-transient loop of multi.jl:458-471 / gpu.jl:126-137.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <vector>

#include "ns3d_internal.h"

static thread_local char g_err[512] = "";

int ns3d_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// tile choices measured so far in this process (per device and grid): contexts come and go (one per driver call), the
// measurement should not be repeated
// `pinned`: the pass depth that was FORCED when the entry was measured (ns3d_set_pt_depth, or ns3d_slab_plan / box_plan pinning the
// depth their ghosts allow) — 0 for a free choice; `cus_off`: compute units left out of the launches (ns3d_reserve_cus).  Both are
// part of the key: an entry tuned under a pin or a CU mask must not answer a lookup made without it, nor the other way round
// (a forced-depth entry used to supply its depth to later unpinned lookups).
struct Tuned { int device, nx, ny, nz, nk, esize, mode, pinned, cus_off, variant, depth, variantn; };
static std::vector<Tuned> g_tuned;
static std::mutex g_tuned_mutex;

// launch + (unless NS3D_ASYNC) block like `@parallel` does
static int finish(ns3d_ctx *ctx, hipError_t e, const char *what)
{
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(NS3D_ERR_HIP, "%s launch: %s", what, hipGetErrorString(e)); }
    if (!(ctx->flags & NS3D_ASYNC)) {
        e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) { (void)hipGetLastError(); return fail(NS3D_ERR_HIP, "%s: %s", what, hipGetErrorString(e)); }
    }
    return NS3D_OK;
}

// arithmetic mode of a launch: 2 = FAST, 1 = STRICT with the exact division-by-known-divisor sequence (bit-identical
// to 0, chosen when every spacing passes recip_ok), 0 = STRICT with plain IEEE divisions
static bool recip_ok(double d)
{
    if (!(d > 0x1p-100 && d < 0x1p100)) return false;           // also rejects NaN, 0, negatives
    unsigned long long bits;
    std::memcpy(&bits, &d, sizeof bits);
    if ((bits & 0x000FFFFFFFFFFFFFull) == 0x000FFFFFFFFFFFFFull) return false; // significand all ones
    const float f = (float)d;                                     // the f32 kernels divide by (float)d
    unsigned int fb;
    std::memcpy(&fb, &f, sizeof fb);
    if (!(f > 0x1p-20f && f < 0x1p20f) || (fb & 0x007FFFFFu) == 0x007FFFFFu) return false;
    return true;
}
// d = 2^e exactly with -30 ≤ e ≤ 0: 1/d and 1/d² are exact powers of two ≥ 1 in double and float alike (ns3d_strictp)
static bool pow2_ok(double d)
{
    if (!(d >= 0x1p-30 && d <= 1.0)) return false;
    int e;
    return std::frexp(d, &e) == 0.5;
}
// 3 = STRICT on power-of-two spacings (x/d ≡ x·(1/d), same bits as 0 and 1 with plain multiplications)
static int mode_of(const ns3d_ctx *c, double dx, double dy, double dz)
{
    if (c->flags & NS3D_FAST) return 2;
    if (c->flags & NS3D_IEEE_DIV) return 0;
    if (pow2_ok(dx) && pow2_ok(dy) && pow2_ok(dz)) return 3;
    return (recip_ok(dx) && recip_ok(dy) && recip_ok(dz)) ? 1 : 0;
}
#define DISPATCHM(mode, call)                                                                               \
    ((mode) == 2 ? ns3d_fast::call : (mode) == 3 ? ns3d_strictp::call : (mode) == 1 ? ns3d_strictx::call : ns3d_strict::call)
#define DISPATCH(ctx, call) DISPATCHM(((ctx)->flags & NS3D_FAST) ? 2 : 0, call)      /* kernels without divisions */
#define DISPATCHG(ctx, dx, dy, dz, call) DISPATCHM(mode_of((ctx), (dx), (dy), (dz)), call)

extern "C" {

int ns3d_version(void) { return NS3D_VERSION; }
const char *ns3d_last_error(void) { return g_err; }

ns3d_ctx *ns3d_create(int device, int flags)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        fail(NS3D_ERR_HIP, "ns3d_create: no HIP device (%s) — libns3d has no CPU path", hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= ndev) {
        fail(NS3D_ERR_ARG, "ns3d_create: device %d out of range [0,%d)", device, ndev);
        return nullptr;
    }
    ns3d_device_guard guard(device);        // the caller's current device is put back on return
    if (guard.err != hipSuccess) {
        fail(NS3D_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(guard.err));
        return nullptr;
    }
    ns3d_ctx *c = new ns3d_ctx();
    c->device = device;
    c->flags = flags;
    c->own_stream = nullptr;
    c->stream = nullptr;
    c->pingpong = nullptr;
    c->pingpong_bytes = 0;
    c->pingpong_d = nullptr;
    c->pingpong_d_bytes = 0;
    c->pt_variant = 0;
    c->pt2_variant = 0; // temporal blocking on by default (ns3d_set_pt2_variant(ctx,-1) turns it off)
    if (const char *ev = std::getenv("NS3D_PT2_VARIANT")) c->pt2_variant = std::atoi(ev);   // experiments without an API call
    c->ptn_variant = 0;
    if (const char *ev = std::getenv("NS3D_PTN_VARIANT")) c->ptn_variant = std::atoi(ev);
    c->pt_depth = 0;
    if (const char *ev = std::getenv("NS3D_PT_DEPTH")) c->pt_depth = std::atoi(ev);
    c->autotune = 1;
    c->last_pt2 = 0;
    c->last_ptn = 0;
    c->last_depth = 0;
    c->tune_ev[0] = c->tune_ev[1] = nullptr;
    c->graph_mode = -1;
    if (const char *ev = std::getenv("NS3D_GRAPH_MODE")) c->graph_mode = std::atoi(ev);     // A/B without an API call
    if (const char *ev = std::getenv("NS3D_PT_PERSIST")) c->persist_mode = std::atoi(ev);
    c->fence = nullptr;
    c->key_dev = nullptr;
    c->key_host = nullptr;
    if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void **)&c->key_dev, 64)) != hipSuccess ||
        (e = hipHostMalloc((void **)&c->key_host, 64, hipHostMallocDefault)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->fence, hipEventDisableTiming)) != hipSuccess) {
        fail(NS3D_ERR_HIP, "ns3d_create: %s", hipGetErrorString(e));
        ns3d_destroy(c);                    // releases whatever was created so far
        return nullptr;
    }
    c->stream = c->own_stream;
    if (const char *ev = std::getenv("NS3D_RESERVE_CUS"))
        if (std::atoi(ev) > 0 && ns3d_reserve_cus(c, std::atoi(ev)) != NS3D_OK) { ns3d_destroy(c); return nullptr; }
    return c;
}

void ns3d_destroy(ns3d_ctx *c)
{
    if (!c) return;
    ns3d_device_guard guard(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->own_stream) (void)hipStreamSynchronize(c->own_stream);
    c->clear_graphs();
    if (c->persist.H) (void)hipFree(c->persist.H);
    if (c->persist.err_host) (void)hipHostFree(c->persist.err_host);
    if (c->persist.ev) (void)hipEventDestroy(c->persist.ev);
    if (c->direct_plan && c->direct_free) c->direct_free(c->direct_plan);
    if (c->fence) (void)hipEventDestroy(c->fence);
    for (int q = 0; q < 2; ++q)
        if (c->tune_ev[q]) (void)hipEventDestroy(c->tune_ev[q]);
    if (c->pingpong) (void)hipFree(c->pingpong);
    if (c->pingpong_d) (void)hipFree(c->pingpong_d);
    if (c->key_dev) (void)hipFree(c->key_dev);
    if (c->key_host) (void)hipHostFree(c->key_host);
    if (c->masked_stream) { (void)hipStreamSynchronize(c->masked_stream); (void)hipStreamDestroy(c->masked_stream); }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int ns3d_flags(const ns3d_ctx *c) { return c ? c->flags : -1; }

int ns3d_set_stream(ns3d_ctx *c, void *s)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_stream: null context");
    c->stream = (hipStream_t)s;
    return NS3D_OK;
}
int ns3d_use_own_stream(ns3d_ctx *c)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_use_own_stream: null context");
    c->stream = c->own_stream;
    return NS3D_OK;
}
void *ns3d_get_stream(ns3d_ctx *c) { return c ? (void *)c->stream : nullptr; }

// CU mask: bit i of the mask words = compute unit i as the runtime numbers them.  How those numbers map onto the 8 XCDs is not
// documented for this part, so both plausible layouts exist (NS3D_RESERVE_CUS_LAYOUT): 0 = numbers interleave the XCDs (CU i sits
// on XCD i mod 8: drop the n highest numbers), 1 = 32 consecutive numbers per XCD (drop the n/8 highest of every 32).  Measured
// (profiles/r4_cu_mask_ab.log): the interleaved reading is the one under which the masked sweep loses time in proportion to the CUs.
int ns3d_reserve_cus(ns3d_ctx *c, int n)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_reserve_cus: null context");
    ns3d_device_guard guard(c->device);
    int cus = 0;
    HIPCHK(c, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
    if (n < 0 || n > cus / 2) return fail(NS3D_ERR_ARG, "ns3d_reserve_cus: %d of %d compute units (0 … half the device)", n, cus);
    n = (n + 7) / 8 * 8;
    if (c->masked_stream) {
        HIPCHK(c, hipStreamSynchronize(c->masked_stream));
        if (c->stream == c->masked_stream) c->stream = c->own_stream;
        c->clear_graphs();
        HIPCHK(c, hipStreamDestroy(c->masked_stream));
        c->masked_stream = nullptr;
        c->reserved_cus = 0;
    }
    if (n == 0) return NS3D_OK;
    static const int layout = std::getenv("NS3D_RESERVE_CUS_LAYOUT") ? std::atoi(std::getenv("NS3D_RESERVE_CUS_LAYOUT")) : 0;
    const int words = (cus + 31) / 32;
    std::vector<uint32_t> mask((size_t)words, 0u);
    for (int i = 0; i < cus; ++i) {
        bool keep;
        if (layout == 1) keep = (i % 32) < 32 - n / 8 || cus % 32 != 0;      // n/8 off the top of every 32-CU group
        else keep = i < cus - n;                                               // the n highest numbers
        if (keep) mask[(size_t)(i / 32)] |= 1u << (i % 32);
    }
    HIPCHK(c, hipExtStreamCreateWithCUMask(&c->masked_stream, (uint32_t)words, mask.data()));
    c->reserved_cus = n;
    c->stream = c->masked_stream;
    return NS3D_OK;
}
int ns3d_reserved_cus(const ns3d_ctx *c) { return c ? c->reserved_cus : -1; }

int ns3d_sync(ns3d_ctx *c)
{
    CHECK_CTX(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return NS3D_OK;
}

int ns3d_set_pt_variant(ns3d_ctx *c, int v)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_pt_variant: null context");
    if (v < 0 || v >= 10000) return fail(NS3D_ERR_ARG, "ns3d_set_pt_variant: unknown variant %d", v);
    c->pt_variant = v;
    return NS3D_OK;
}

int ns3d_set_graph_mode(ns3d_ctx *c, int mode)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_graph_mode: null context");
    c->graph_mode = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
    return NS3D_OK;
}

int ns3d_set_autotune(ns3d_ctx *c, int on)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_autotune: null context");
    c->autotune = on ? 1 : 0;
    return NS3D_OK;
}

int ns3d_last_pt2_variant(const ns3d_ctx *c) { return c ? c->last_pt2 : -1; }
int ns3d_last_ptn_variant(const ns3d_ctx *c) { return c ? c->last_ptn : -1; }
int ns3d_last_pt_depth(const ns3d_ctx *c) { return c ? c->last_depth : -1; }
int ns3d_set_persist_mode(ns3d_ctx *c, int mode)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_persist_mode: null context");
    if (mode < -1 || mode > 1) return fail(NS3D_ERR_ARG, "ns3d_set_persist_mode: %d (-1 auto, 0 off, 1 on)", mode);
    c->persist_mode = mode;
    return NS3D_OK;
}
int ns3d_cached_graphs(const ns3d_ctx *c) { return c ? (int)c->graphs.size() : -1; }
int ns3d_arith_build(const ns3d_ctx *c, double dx, double dy, double dz) { return c ? mode_of(c, dx, dy, dz) : -1; }

int ns3d_set_ptn_variant(ns3d_ctx *c, int v)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_ptn_variant: null context");
    if (v < 0 || v >= 10000) return fail(NS3D_ERR_ARG, "ns3d_set_ptn_variant: unknown variant %d", v);
    c->ptn_variant = v;
    return NS3D_OK;
}

int ns3d_persist_faults(const ns3d_ctx *c) { return c ? (int)c->persist.faults : -1; }
int ns3d_set_pt_depth(ns3d_ctx *c, int depth)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_pt_depth: null context");
    if (depth < 0 || depth > 5) return fail(NS3D_ERR_ARG, "ns3d_set_pt_depth: depth %d (0 = automatic, 1…4; 5: float32 fields only)", depth);
    c->pt_depth = depth;
    return NS3D_OK;
}

int ns3d_set_pt2_variant(ns3d_ctx *c, int v)
{
    if (!c) return fail(NS3D_ERR_ARG, "ns3d_set_pt2_variant: null context");
    if (v >= 10000) return fail(NS3D_ERR_ARG, "ns3d_set_pt2_variant: unknown variant %d", v);
    c->pt2_variant = v;
    return NS3D_OK;
}

} // extern "C"

// read back a reduction key: 8-byte D2H into pinned memory + stream sync (the only host round trip of the PT loop)
static int fetch_key(ns3d_ctx *c, hipStream_t s, double *out)
{
    HIPCHK(c, hipMemcpyAsync(c->key_host, c->key_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    double v;
    std::memcpy(&v, c->key_host, sizeof v);
    *out = v;
    return NS3D_OK;
}

// After a synchronisation of the stream the persist launches ran on: did a bounded wait expire in any of them since the last check?
// (k_pt_persist writes its ticket into pinned host memory when one does.)  A failure turns the cooperative form off for this context:
// what made the workgroups non-resident — another context or process holding CUs — is unlikely to have gone away.
static bool persist_failed(ns3d_ctx *c)
{
    ns3d_persist_state &ps = c->persist;
    if (!ps.err_host || ps.checked == ps.ticket) return false;
    const unsigned seen = *(volatile unsigned *)ps.err_host;
    const bool failed = seen != 0u && (int)(seen - ps.checked) > 0 && (int)(seen - ps.ticket) <= 0;
    ps.checked = ps.ticket;
    if (failed) { ++ps.faults; c->persist_mode = 0; }
    return failed;
}

int ns3d_check_pt_params(const ns3d_pt_params *p, const char *fn)
{
    if (!p) return fail(NS3D_ERR_ARG, "%s: null params", fn);
    if (p->nx < 3 || p->ny < 3 || p->nz < 3)
        return fail(NS3D_ERR_ARG, "%s: grid %dx%dx%d too small for the fused PT sweep (need >= 3)", fn, p->nx, p->ny,
                    p->nz);
    if (p->bc_kind != NS3D_BC_MULTI && p->bc_kind != NS3D_BC_GPU) return fail(NS3D_ERR_ARG, "%s: bad bc_kind %d", fn, p->bc_kind);
    if (p->bc_kind == NS3D_BC_GPU && (p->z_lo_is_halo || p->z_hi_is_halo))
        return fail(NS3D_ERR_ARG, "%s: gpu.jl boundary set is single-device (no z halos)", fn);
    return NS3D_OK;
}

// Temporal blocking pays from ≈1.5 M cells on (128³: 137 000 against 113 000 Mcells·iter/s, 160³: 169 000 against 124 000;
// 127×76×76 = 0.7 M cells: 55 000 against 70 000 — there the one-thread-per-cell sweep has more parallelism);
// an explicit ns3d_set_pt2_variant(ctx, v>0) forces it, v<0 disables it.
static const long long NS3D_TWO_MIN_CELLS = 1500ll * 1000;
static bool use_two(const ns3d_ctx *c, const ns3d_pt_params *p)
{
    if (c->pt2_variant < 0 || c->pt_depth == 1) return false;
    if (c->pt2_variant > 0 || c->pt_depth >= 2) return true;
    return (long long)p->nx * p->ny * p->nz >= NS3D_TWO_MIN_CELLS;
}

template <class T>
static int ensure_pingpong(ns3d_ctx *c, const ns3d_pt_params *p, T **buf)
{
    const size_t need = (size_t)p->nx * p->ny * p->nz * sizeof(T);
    if (c->pingpong_bytes < need) {
        if (c->pingpong) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipStreamSynchronize(c->own_stream));
            c->clear_graphs();
            HIPCHK(c, hipFree(c->pingpong));
            c->pingpong = nullptr;
            c->pingpong_bytes = 0;
        }
        HIPCHK(c, hipMalloc(&c->pingpong, need));
        c->pingpong_bytes = need;
    }
    *buf = (T *)c->pingpong;
    return NS3D_OK;
}

template <class T>
static int ensure_pingpong_d(ns3d_ctx *c, const ns3d_pt_params *p, T **buf)
{
    const size_t need = (size_t)(p->nx - 2) * (p->ny - 2) * (p->nz - 2) * sizeof(T);
    if (c->pingpong_d_bytes < need) {
        if (c->pingpong_d) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipStreamSynchronize(c->own_stream));
            c->clear_graphs();
            HIPCHK(c, hipFree(c->pingpong_d));
            c->pingpong_d = nullptr;
            c->pingpong_d_bytes = 0;
        }
        HIPCHK(c, hipMalloc(&c->pingpong_d, need));
        c->pingpong_d_bytes = need;
    }
    *buf = (T *)c->pingpong_d;
    return NS3D_OK;
}

// ---- tile shape of the two-iteration sweep ------------------------------------------------------------------
// Every shape gives the same bits (tests/test_gpu_pt.py), and which one is fastest depends on how the row length
// divides into 64/128/256-wide tiles and on how many workgroups the grid yields (profiles/r1b_shapes.log: 128×8 with
// two workgroups per CU wins at 255×153×153 and 384³, 256×8 at 512³ and 1024³).  So the first automatic launch on a
// grid times the candidates on the caller's own arguments — the operation is idempotent: inputs and outputs are
// distinct buffers — and remembers the winner in the context.  Skipped (built-in choice by grid instead) while the
// stream is being captured, for launches under 1.5 M cells, after ns3d_set_autotune(ctx, 0), or with an explicit variant.
// How a pass over memory is made: PT iterations per pass (depth 2: k_pt_sweep2 with tile variant v2; 3, 4: k_pt_sweepN with
// tile variant vn).  Explicit settings (ns3d_set_pt2_variant / _ptn_variant / _pt_depth) always win.
struct Plan { int depth, v2, vn; bool known; };
// Below this the planner does not time deeper passes.  Round 4 lowered it from 8 M to 3 M cells: on the reference's own 255×153×153
// grid (6.0 M cells) three iterations per pass run 34 % faster than two in FAST mode (2800: 17.8 against 23.8 µs per iteration;
// profiles/r4_midsize.log) — the round-2 finding "stays at two" held for the exact-division STRICT build only, which is
// VALU-bound there (29.8 against 28.4) and keeps two because the planner measures head to head.  127×77×77 (0.75 M): two.
static const long long NS3D_DEEP_MIN_CELLS = 3ll * 1000 * 1000;
template <class T>
static Plan lookup_plan(const ns3d_ctx *c, int mode, const ns3d_pt_params *p, int k0, int k1)
{
    Plan pl{c->pt_depth > 0 ? c->pt_depth : 2, c->pt2_variant > 0 ? c->pt2_variant : 0, c->ptn_variant, true};
    if (sizeof(T) == 8 && pl.depth > 4) pl.depth = 4;       // the fifth level exists for float32 fields only
    const int nk = k1 - k0;
    const long long cells = (long long)p->nx * p->ny * nk;
    if (!c->autotune || cells < NS3D_TWO_MIN_CELLS) return pl;
    if (c->pt2_variant > 0 && (c->pt_depth > 0 || cells < NS3D_DEEP_MIN_CELLS)) return pl;      // nothing left to decide
    if (c->pt_depth >= 3 && c->ptn_variant > 0) return pl;                                      // deep passes, shape given
    std::lock_guard<std::mutex> lock(g_tuned_mutex);
    for (const auto &t : g_tuned)
        if (t.device == c->device && t.nx == p->nx && t.ny == p->ny && t.nz == p->nz && t.nk == nk &&
            t.esize == (int)sizeof(T) && t.mode == mode && t.pinned == c->pt_depth && t.cus_off == c->reserved_cus) {
            if (c->pt2_variant <= 0) pl.v2 = t.variant;
            if (c->pt_depth <= 0) pl.depth = t.depth;
            if (c->ptn_variant <= 0) pl.vn = t.variantn;
            return pl;
        }
    pl.known = false;
    return pl;
}
// the measurement: runs on the caller's own arguments (idempotent: inputs and outputs are distinct buffers), blocks on
// its events.  Called from ns3d_plan_pt, ns3d_pt_iterate and ns3d_pt_solve only — never from ns3d_pt_sweep2 / _sweepn, whose
// callers (z-slab schedules with an exchange in flight) must not be stalled by ≈100 extra launches.
template <class T>
static Plan tune_plan(ns3d_ctx *c, hipStream_t s, int mode, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                      const ns3d_pt_params *p, int k0, int k1)
{
    Plan pl = lookup_plan<T>(c, mode, p, k0, k1);
    pl.known = true;
    const int nk = k1 - k0;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); return pl; }
    if (cap != hipStreamCaptureStatusNone) return pl;
    for (int q = 0; q < 2; ++q)
        if (!c->tune_ev[q] && hipEventCreate(&c->tune_ev[q]) != hipSuccess) { (void)hipGetLastError(); return pl; }
    // `depth` iterations per launch; ms = time of ONE launch
    auto time_launch = [&](int depth, int v, float &ms, int timed = 3) -> bool {
        bool ok = true;
        for (int rep = 0; rep <= timed && ok; ++rep) {     // one untimed launch, then `timed` timed ones
            if (rep == 1) ok = hipEventRecord(c->tune_ev[0], s) == hipSuccess;
            const int tf = (c->reserved_cus / 8) << 8;      // compute units the launch must not count on
            hipError_t e = depth == 2 ? DISPATCHM(mode, pt_sweep2<T>(s, v, src, dst, dsrc, ddst, divV, *p, k0, k1, tf))
                                      : DISPATCHM(mode, pt_sweepn<T>(s, depth, v, src, dst, dsrc, ddst, divV, *p, k0, k1, tf));
            ok = ok && e == hipSuccess;
        }
        ok = ok && hipEventRecord(c->tune_ev[1], s) == hipSuccess && hipEventSynchronize(c->tune_ev[1]) == hipSuccess &&
             hipEventElapsedTime(&ms, c->tune_ev[0], c->tune_ev[1]) == hipSuccess;
        ms /= (float)timed;
        if (!ok) (void)hipGetLastError();
        return ok;
    };
    // ---- two iterations per pass: variant = shape*100 + kz (ns3d.h).  Stage 1: every shape with round-filling chunks (92)
    // and with short 16-plane chunks (which keep neighbouring tiles close in time: their overlaps then hit the L2); stage 2:
    // the other chunk lengths for the three best shapes.  0 = the built-in choice; it wins ties.
    float ms2 = 0.f;                                       // per launch of the chosen two-iteration variant
    if (c->pt2_variant > 0) {
        if (!time_launch(2, pl.v2, ms2, 6) || !time_launch(2, pl.v2, ms2)) return pl;
    } else {
        static const int shapes[] = {11, 8, 9, 7, 13, 19, 12};
        static const int stage1[] = {92, 16}, stage2[] = {94, 91, 98, 32};
        constexpr int NS = (int)(sizeof shapes / sizeof shapes[0]);
        int best = 0;
        float best_ms = 0.f, ms = 0.f, shape_ms[NS];
        if (!time_launch(2, 0, best_ms, 6)) return pl;     // also brings the clocks up
        if (!time_launch(2, 0, best_ms)) return pl;
        for (int q = 0; q < NS; ++q) {
            shape_ms[q] = 0.f;
            for (int kz : stage1) {
                if (!time_launch(2, shapes[q] * 100 + kz, ms)) return pl;
                if (shape_ms[q] == 0.f || ms < shape_ms[q]) shape_ms[q] = ms;
                if (ms < 0.99f * best_ms) { best_ms = ms; best = shapes[q] * 100 + kz; }
            }
        }
        for (int pick = 0; pick < 3; ++pick) {
            int q = -1;
            for (int r = 0; r < NS; ++r)
                if (shape_ms[r] > 0.f && (q < 0 || shape_ms[r] < shape_ms[q])) q = r;
            if (q < 0) break;
            shape_ms[q] = 0.f;                             // taken
            for (int kz : stage2) {
                if (!time_launch(2, shapes[q] * 100 + kz, ms)) return pl;
                if (ms < 0.99f * best_ms) { best_ms = ms; best = shapes[q] * 100 + kz; }
            }
        }
        if (best != 0) {                                   // head to head against the built-in choice, longer runs
            float ms0 = 0.f, ms1 = 0.f;
            if (!time_launch(2, 0, ms0, 6) || !time_launch(2, best, ms1, 6)) return pl;
            if (!(ms1 < 0.97f * ms0)) { best = 0; best_ms = ms0; } else best_ms = ms1;
        }
        pl.v2 = best;
        ms2 = best_ms;
    }
    // ---- three iterations per pass (k_pt_sweepN): fewer bytes per iteration, more arithmetic per byte — pays where the
    // two-iteration pass is bandwidth-bound (FAST mode, power-of-two spacings), not where it is VALU-bound.  Taken only for
    // a clear per-ITERATION gain (≥ 2 %), head to head.
    const long long cells = (long long)p->nx * p->ny * nk;
    if (c->pt_depth <= 0 && cells >= NS3D_DEEP_MIN_CELLS && nk >= 12) {
        static const int cand[] = {1100, 2800, 2300, 100, 1600, 600, 2200, 1132, 3800};   // the first one is the built-in shape: it wins near-ties
        int bestn = c->ptn_variant;
        float best3 = 0.f, ms = 0.f;
        if (c->ptn_variant > 0) { if (!time_launch(3, bestn, best3)) best3 = 0.f; }
        else
            for (int v : cand) {
                if (!time_launch(3, v, ms)) continue;      // a shape that cannot run here (LDS / tile size)
                if (best3 == 0.f || ms < 0.98f * best3) { best3 = ms; bestn = v; }
            }
        if (best3 > 0.f && best3 / 3.f < 0.99f * ms2 / 2.f) {
            float a2 = 0.f, a3 = 0.f;
            if (time_launch(2, pl.v2, a2, 6) && time_launch(3, bestn, a3, 6) && a3 / 3.f < 0.98f * a2 / 2.f) {
                pl.depth = 3;
                pl.vn = bestn;
            }
        }
        // four iterations per pass: needs three or four waves per SIMD to pay — in fp64 768-thread workgroups of 64×24 columns
        // (two rows per thread: 154 registers), in fp32 1024-thread workgroups of 64×32 columns (two rows per thread, ≤ 128
        // registers) or 768 threads on 64×48; 2391 / 2891 = ONE round of workgroups, each marching the whole z range
        if (nk >= 16) {
            // 3800 (round 4): k_pt_sweepD, the planes of P⁰ through an LDS-DMA ring — what lets fp64 fit a 1024-thread 64×32 tile;
            // ties 2800 at 512³ (180 tiles for 256 CUs), so it has to win its place by 2 % like every later candidate
            static const int cand4_f32[] = {2400, 2200, 1100, 0, 0}, cand4_f64[] = {2800, 2891, 2300, 2391, 3800};
            const int *cand4 = sizeof(T) == 4 ? cand4_f32 : cand4_f64;
            const float cur_per_it = pl.depth == 3 ? best3 / 3.f : ms2 / 2.f;
            int best4v = 0;
            float best4 = 0.f;
            for (int q4 = 0; q4 < 5; ++q4) {
                const int v = cand4[q4];
                if (v == 0) continue;
                if (c->ptn_variant > 0 && v != c->ptn_variant) continue;
                if (!time_launch(4, v, ms)) continue;
                if (best4 == 0.f || ms < 0.98f * best4) { best4 = ms; best4v = v; }
            }
            if (best4 > 0.f && best4 / 4.f < 0.98f * cur_per_it) {
                float a = 0.f, b4 = 0.f;
                const bool ok = (pl.depth == 3 ? time_launch(3, pl.vn, a, 6) : time_launch(2, pl.v2, a, 6)) && time_launch(4, best4v, b4, 6);
                if (ok && b4 / 4.f < 0.97f * a / (float)pl.depth) {
                    pl.depth = 4;
                    pl.vn = best4v;
                }
            }
            // a fifth level: fp32 only, on the 1024-thread shape (registers to spare there and nowhere else): +6 % at 512³, +9.5 % at
            // 1024³ per iteration
            if (sizeof(T) == 4 && pl.depth == 4 && nk >= 24 && (c->ptn_variant <= 0 || c->ptn_variant / 100 == 24)) {
                const int v5 = c->ptn_variant > 0 ? c->ptn_variant : 2400;
                float a4 = 0.f, b5 = 0.f;
                if (time_launch(5, v5, ms) && ms / 5.f < 0.99f * best4 / 4.f && time_launch(4, pl.vn, a4, 6) && time_launch(5, v5, b5, 6) &&
                    b5 / 5.f < 0.98f * a4 / 4.f) {
                    pl.depth = 5;
                    pl.vn = v5;
                }
            }
        }
    } else if (c->pt_depth >= 3 && c->ptn_variant <= 0 && cells >= NS3D_TWO_MIN_CELLS) {
        // (24xx: float32 only, 28xx / 23xx with 768 threads: both element types; a shape that cannot run the depth is skipped)
        static const int cand[] = {1100, 2400, 2491, 2300, 2391, 2800, 2891, 100, 1600, 600, 2200, 1132, 3800};
        float bestd = 0.f, ms = 0.f;
        for (int v : cand) {
            if (!time_launch(c->pt_depth, v, ms)) continue;
            if (bestd == 0.f || ms < bestd) { bestd = ms; pl.vn = v; }
        }
    }
    {
        std::lock_guard<std::mutex> lock(g_tuned_mutex);
        g_tuned.push_back({c->device, p->nx, p->ny, p->nz, nk, (int)sizeof(T), mode, c->pt_depth, c->reserved_cus, pl.v2, pl.depth, pl.vn});
    }
    return pl;
}
template <class T>
static Plan pick_plan(ns3d_ctx *c, hipStream_t s, int mode, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                      const ns3d_pt_params *p, int k0, int k1, bool may_tune)
{
    Plan pl = lookup_plan<T>(c, mode, p, k0, k1);
    if (!pl.known && may_tune) pl = tune_plan<T>(c, s, mode, src, dst, dsrc, ddst, divV, p, k0, k1);
    return pl;
}
// one pass of `depth` (2…4, fp32: 5) PT iterations
template <class T>
static hipError_t launch_pass(ns3d_ctx *c, hipStream_t s, int depth, const Plan &pl, const T *src, T *dst, const T *dsrc, T *ddst,
                              const T *divV, const ns3d_pt_params *p, int k0, int k1, const ns3d_tile_window *win = nullptr,
                              int more_flags = 0)
{
    const int mode = mode_of(c, p->dx, p->dy, p->dz);
    c->last_depth = depth;
    const int flags = more_flags | ((c->reserved_cus / 8) << 8);   // bits 8…: compute units the launch must not count on (in eights)
    if (depth == 2) {
        c->last_pt2 = pl.v2;
        return DISPATCHM(mode, pt_sweep2<T>(s, pl.v2, src, dst, dsrc, ddst, divV, *p, k0, k1, flags, win));
    }
    c->last_ptn = pl.vn;
    return DISPATCHM(mode, pt_sweepn<T>(s, depth, pl.vn, src, dst, dsrc, ddst, divV, *p, k0, k1, flags, win));
}
// iterations of the next pass when `rem` remain until the next residual check / the end
static int next_depth(const Plan &pl, bool blocked, int rem)
{
    if (!blocked || rem < 2) return 1;
    if (rem >= pl.depth) return (rem == pl.depth + 1 && pl.depth >= 3) ? pl.depth - 1 : pl.depth;   // 4 = 2+2, not 3+1
    return rem;                                             // 2 … depth−1 left: one pass of exactly that many
}
template <class T>
static hipError_t launch_pt2(ns3d_ctx *c, hipStream_t s, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                             const ns3d_pt_params *p, int k0, int k1, bool may_tune)
{
    const Plan pl = pick_plan<T>(c, s, mode_of(c, p->dx, p->dy, p->dz), src, dst, dsrc, ddst, divV, p, k0, k1, may_tune);
    return launch_pass<T>(c, s, 2, pl, src, dst, dsrc, ddst, divV, p, k0, k1);
}

// ---- what the multi-GPU layer enqueues on its ranks' contexts (ns3d_internal.h) ----------------------------------
template <class T>
hipError_t ns3d_enqueue_pt2(ns3d_ctx *c, hipStream_t s, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                            const ns3d_pt_params *p, int k0, int k1)
{
    return launch_pt2<T>(c, s, src, dst, dsrc, ddst, divV, p, k0, k1, false);
}
template <class T>
hipError_t ns3d_enqueue_pass(ns3d_ctx *c, hipStream_t s, int depth, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                             const ns3d_pt_params *p, int k0, int k1, int v2, int vn, const ns3d_tile_window *win, int skip_faces)
{
    Plan pl = lookup_plan<T>(c, mode_of(c, p->dx, p->dy, p->dz), p, k0, k1);
    if (v2 >= 0 && c->pt2_variant <= 0) pl.v2 = v2;        // the caller's own plan; explicit context settings still win
    if (vn >= 0 && c->ptn_variant <= 0) pl.vn = vn;
    return launch_pass<T>(c, s, depth, pl, src, dst, dsrc, ddst, divV, p, k0, k1, win, skip_faces ? NS3D_PASS_SKIP_FACES : 0);
}
template <class T>
hipError_t ns3d_enqueue_faces_region(ns3d_ctx *c, hipStream_t s, T *Pout, const ns3d_pt_params *p, const int c0[3], const int c1[3],
                                     int want_core)
{
    return DISPATCHG(c, p->dx, p->dy, p->dz, pt_faces_region<T>(s, Pout, *p, c0, c1, want_core));
}
template <class T>
int ns3d_plan_pt_internal(ns3d_ctx *c, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV, const ns3d_pt_params *p,
                          int k0, int k1)
{
    const Plan pl = pick_plan<T>(c, c->stream, mode_of(c, p->dx, p->dy, p->dz), src, dst, dsrc, ddst, divV, p, k0, k1, true);
    c->last_pt2 = pl.v2; c->last_ptn = pl.vn; c->last_depth = pl.depth;
    return pl.depth;
}
template <class T>
hipError_t ns3d_enqueue_pt1(ns3d_ctx *c, hipStream_t s, const T *src, T *dst, T *d, const T *divV, const ns3d_pt_params *p,
                            int k0, int k1)
{
    return DISPATCHG(c, p->dx, p->dy, p->dz, pt_sweep<T>(s, c->pt_variant, src, dst, d, divV, *p, k0, k1));
}
template <class T>
hipError_t ns3d_enqueue_residual_key(ns3d_ctx *c, hipStream_t s, const T *Pr, const T *divV, const ns3d_pt_params *p,
                                     unsigned long long *key_dev)
{
    return DISPATCHG(c, p->dx, p->dy, p->dz, residual_max_key<T>(s, Pr, divV, *p, key_dev));
}
template <class T>
hipError_t ns3d_enqueue_strip_inner(ns3d_ctx *c, hipStream_t s, const T *A, T *out, int sx, int sy, int sz)
{
    return DISPATCH(c, strip_inner<T>(s, A, out, sx, sy, sz));
}
template <class T>
hipError_t ns3d_enqueue_face_copy(ns3d_ctx *c, hipStream_t s, T *A, T *buf, int sx, int sy, int sz, int dim, int idx, int unpack)
{
    return DISPATCH(c, face_copy<T>(s, A, buf, sx, sy, sz, dim, idx, unpack));
}
template <class T>
hipError_t ns3d_enqueue_subbox_copy(ns3d_ctx *c, hipStream_t s, const ns3d_subbox_batch<T> &batch)
{
    return DISPATCH(c, subbox_copy<T>(s, batch));
}
template <class T>
hipError_t ns3d_enqueue_advect(ns3d_ctx *c, hipStream_t s, T *Vx, const T *Vx_o, T *Vy, const T *Vy_o, T *Vz, const T *Vz_o, T *C,
                               const T *C_o, double dt, double dx, double dy, double dz, int nx, int ny, int nz, int flags, int koff,
                               int nzg)
{
    return DISPATCHG(c, dx, dy, dz, advect<T>(s, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy, dz, nx, ny, nz, flags, koff, nzg));
}
#define NS3D_INST_INTERNAL(T)                                                                                       \
    template hipError_t ns3d_enqueue_advect<T>(ns3d_ctx *, hipStream_t, T *, const T *, T *, const T *, T *, const T *, T *, \
                                               const T *, double, double, double, double, int, int, int, int, int, int); \
    template hipError_t ns3d_enqueue_pt2<T>(ns3d_ctx *, hipStream_t, const T *, T *, const T *, T *, const T *,      \
                                            const ns3d_pt_params *, int, int);                                      \
    template hipError_t ns3d_enqueue_pass<T>(ns3d_ctx *, hipStream_t, int, const T *, T *, const T *, T *, const T *, \
                                             const ns3d_pt_params *, int, int, int, int, const ns3d_tile_window *, int); \
    template hipError_t ns3d_enqueue_faces_region<T>(ns3d_ctx *, hipStream_t, T *, const ns3d_pt_params *, const int *, const int *, int); \
    template int ns3d_plan_pt_internal<T>(ns3d_ctx *, const T *, T *, const T *, T *, const T *,                     \
                                          const ns3d_pt_params *, int, int);                                        \
    template hipError_t ns3d_enqueue_pt1<T>(ns3d_ctx *, hipStream_t, const T *, T *, T *, const T *,                 \
                                            const ns3d_pt_params *, int, int);                                      \
    template hipError_t ns3d_enqueue_residual_key<T>(ns3d_ctx *, hipStream_t, const T *, const T *,                  \
                                                     const ns3d_pt_params *, unsigned long long *);                 \
    template hipError_t ns3d_enqueue_strip_inner<T>(ns3d_ctx *, hipStream_t, const T *, T *, int, int, int);         \
    template hipError_t ns3d_enqueue_face_copy<T>(ns3d_ctx *, hipStream_t, T *, T *, int, int, int, int, int, int);  \
    template hipError_t ns3d_enqueue_subbox_copy<T>(ns3d_ctx *, hipStream_t, const ns3d_subbox_batch<T> &);
NS3D_INST_INTERNAL(double)
NS3D_INST_INTERNAL(float)
#undef NS3D_INST_INTERNAL

template <class T>
static bool use_persist(const ns3d_ctx *c, const ns3d_pt_params *p);
template <class T>
static hipError_t enqueue_iters(ns3d_ctx *c, hipStream_t s, int n, bool two, T *&src, T *&dst, T *&dsrc, T *&ddst,
                                const T *divV, const ns3d_pt_params *p, bool may_tune);

// n_iters fused sweeps, result left in Pr (one D2D copy when n_iters is odd).  With z halos the scratch
// buffer's halo planes are seeded from Pr first (a sweep never writes them).
template <class T>
static int pt_iterate_impl(ns3d_ctx *c, T *Pr, T *D, const T *divV, const ns3d_pt_params *p, int n_iters)
{
    if (n_iters <= 0) return NS3D_OK;
    T *other = nullptr;
    int rc = ensure_pingpong<T>(c, p, &other);
    if (rc) return rc;
    const size_t plane = (size_t)p->nx * p->ny;
    if (p->z_lo_is_halo) HIPCHK(c, hipMemcpyAsync(other, Pr, plane * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    if (p->z_hi_is_halo)
        HIPCHK(c, hipMemcpyAsync(other + plane * (p->nz - 1), Pr + plane * (p->nz - 1), plane * sizeof(T),
                                 hipMemcpyDeviceToDevice, c->stream));
    T *src = Pr, *dst = other;
    T *dsrc = D, *ddst = nullptr;
    const bool two = use_two(c, p) && !p->z_lo_is_halo && !p->z_hi_is_halo && n_iters >= 2;
    if (two && (rc = ensure_pingpong_d<T>(c, p, &ddst))) return rc;
    const unsigned ticket0 = c->persist.ticket;
    hipError_t e = enqueue_iters<T>(c, c->stream, n_iters, two, src, dst, dsrc, ddst, divV, p, true);
    if (e != hipSuccess) return fail(NS3D_ERR_HIP, "pt_sweep launch: %s", hipGetErrorString(e));
    if (c->persist.ticket != ticket0) {
        // the block ran as ONE cooperative launch: before its result replaces the caller's arrays, make sure no hand-over timed out
        // (a stream synchronisation: these are launch-bound grids, the wait is the block itself); if one did, the inputs are
        // untouched — redo the block by launches (the cooperative form is off for this context from here on)
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (persist_failed(c)) {
            src = Pr; dst = other; dsrc = D; ddst = nullptr;
            if (two && (rc = ensure_pingpong_d<T>(c, p, &ddst))) return rc;
            e = enqueue_iters<T>(c, c->stream, n_iters, two, src, dst, dsrc, ddst, divV, p, true);
            if (e != hipSuccess) return fail(NS3D_ERR_HIP, "pt_sweep launch: %s", hipGetErrorString(e));
        }
    }
    if (dsrc != D)
        HIPCHK(c, hipMemcpyAsync(D, dsrc, (size_t)(p->nx - 2) * (p->ny - 2) * (p->nz - 2) * sizeof(T),
                                 hipMemcpyDeviceToDevice, c->stream));
    if (src != Pr)
        HIPCHK(c, hipMemcpyAsync(Pr, src, plane * p->nz * sizeof(T), hipMemcpyDeviceToDevice, c->stream));
    return NS3D_OK;
}

// NS3D_BC_FUSED=0 (read per call: tests compare the two forms): set_bc_Vel! / set_bc_Pr! rule by rule, a launch each, as rounds 1-3
static bool bc_fused_enabled()
{
    const char *v = std::getenv("NS3D_BC_FUSED");
    return !(v && *v == '0');
}

// ---- the PT loop of multi.jl:458-471 ----------------------------------------------------------------------
// enqueue exactly n iterations on stream s (two per pass where allowed) and leave the result pointers in src/dsrc
template <class T>
static hipError_t enqueue_iters(ns3d_ctx *c, hipStream_t s, int n, bool two, T *&src, T *&dst, T *&dsrc, T *&ddst,
                                const T *divV, const ns3d_pt_params *p, bool may_tune)
{
    hipError_t e = hipSuccess;
    // small grids: the whole block of n iterations in one cooperative launch (k_pt_persist), where it applies
    if (n >= 2 && use_persist<T>(c, p)) {
        // outputs in buffers of their own (dPrdτ too): a launch whose bounded waits expire — its workgroups were not all resident at
        // once: another context or process on the device — leaves the block's inputs intact, and whoever synchronises the stream
        // next (persist_failed below) redoes the block by launches
        T *dout = ddst;
        if (!dout && ensure_pingpong_d<T>(c, p, &dout) != NS3D_OK) return hipErrorOutOfMemory;
        e = DISPATCHG(c, p->dx, p->dy, p->dz, pt_persist<T>(s, src, dst, dsrc, dout, divV, *p, n, &c->persist));
        if (e == hipSuccess) {
            T *t = src; src = dst; dst = t;
            if (ddst) { t = dsrc; dsrc = ddst; ddst = t; }
            else { ddst = dsrc; dsrc = dout; }       // the caller had no second dPrdτ buffer: it has one now (copied back at the end)
            return e;
        }
        if (e != hipErrorInvalidValue) return e;     // invalid value: the form does not apply here → the ordinary path
        (void)hipGetLastError();
        e = hipSuccess;
    }
    Plan pl{2, 0, 0, true};
    if (two && n >= 2)
        pl = pick_plan<T>(c, s, mode_of(c, p->dx, p->dy, p->dz), src, dst, dsrc, ddst, divV, p, 1, p->nz - 1, may_tune);
    for (int it = 0; it < n && e == hipSuccess;) {
        const int d = next_depth(pl, two, n - it);
        if (d >= 2) {
            e = launch_pass<T>(c, s, d, pl, src, dst, dsrc, ddst, divV, p, 1, p->nz - 1);
            T *t = dsrc; dsrc = ddst; ddst = t;
        } else
            e = DISPATCHG(c, p->dx, p->dy, p->dz, pt_sweep<T>(s, c->pt_variant, src, dst, dsrc, divV, *p, 1, p->nz - 1));
        it += d;
        T *t = src; src = dst; dst = t;
    }
    return e;
}

// Field by field: the struct has four bytes of padding after owns_outlet that a C caller's stack struct or Julia's
// Ref(PtParams(…)) leaves indeterminate — a memcmp would miss the cache on every call and re-capture a graph each time.
static bool same_pt_params(const ns3d_pt_params &a, const ns3d_pt_params &b)
{
    return a.rho == b.rho && a.dt == b.dt && a.dtau == b.dtau && a.damp == b.damp && a.dx == b.dx && a.dy == b.dy && a.dz == b.dz &&
           a.nx == b.nx && a.ny == b.ny && a.nz == b.nz && a.bc_kind == b.bc_kind && a.owns_outlet == b.owns_outlet &&
           a.outlet_val == b.outlet_val && a.g == b.g && a.z_lo_is_halo == b.z_lo_is_halo && a.z_hi_is_halo == b.z_hi_is_halo;
}

// One residual-check block (nchk iterations) as a HIP graph: launch-bound grids (63×38×38: ≈2 µs of kernel per ≈5 µs
// launch) replay the whole block with one host call.  Graphs are cached per buffer state in the context.
template <class T>
static int run_block_graph(ns3d_ctx *c, hipStream_t s, int n, bool two, T *&src, T *&dst, T *&dsrc, T *&ddst,
                           const T *divV, const ns3d_pt_params *p)
{
    const int mode = mode_of(c, p->dx, p->dy, p->dz);
    for (auto &g : c->graphs)
        if (g.src == src && g.dst == dst && g.dsrc == dsrc && g.ddst == ddst && g.rhs == divV && g.n == n && g.two == two &&
            g.mode == mode && g.v1 == c->pt_variant && g.v2 == c->pt2_variant && g.vn == c->ptn_variant && g.depth == c->pt_depth &&
            g.esize == (int)sizeof(T) &&
            same_pt_params(g.p, *p)) {
            HIPCHK(c, hipGraphLaunch(g.exec, s));
            src = (T *)g.src_out; dst = (T *)g.dst_out; dsrc = (T *)g.dsrc_out; ddst = (T *)g.ddst_out;
            return NS3D_OK;
        }
    if (c->graphs.size() >= 16) c->clear_graphs();
    ns3d_ctx::BlockGraph g;
    g.src = src; g.dst = dst; g.dsrc = dsrc; g.ddst = ddst; g.rhs = divV; g.n = n; g.two = two; g.mode = mode;
    g.v1 = c->pt_variant; g.v2 = c->pt2_variant; g.vn = c->ptn_variant; g.depth = c->pt_depth; g.esize = (int)sizeof(T); g.p = *p;
    hipGraph_t graph = nullptr;
    HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipError_t e = enqueue_iters<T>(c, s, n, two, src, dst, dsrc, ddst, divV, p, false);
    hipError_t e2 = hipStreamEndCapture(s, &graph);
    if (e != hipSuccess || e2 != hipSuccess) {
        if (graph) (void)hipGraphDestroy(graph);
        return fail(NS3D_ERR_HIP, "PT block graph capture: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    }
    e = hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(NS3D_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    g.src_out = src; g.dst_out = dst; g.dsrc_out = dsrc; g.ddst_out = ddst;
    c->graphs.push_back(g);
    HIPCHK(c, hipGraphLaunch(g.exec, s));
    return NS3D_OK;
}

// k_pt_persist: the launch-bound regime only (a cell per thread, the grid resident for the whole block)
template <class T>
static bool use_persist(const ns3d_ctx *c, const ns3d_pt_params *p)
{
    if (c->persist_mode == 0 || p->z_lo_is_halo || p->z_hi_is_halo) return false;
    if (c->persist_mode < 0) {
        // an explicit request (graph replay, iterations per pass, a tile shape) wins over the automatic choice
        if (c->graph_mode > 0 || c->pt_depth > 0 || c->ptn_variant > 0 || c->pt2_variant != 0) return false;
        // where it was measured to win (profiles/r3_persist_ab.log): up to ≈170 000 cells, one workgroup across x
        if ((long long)p->nx * p->ny * p->nz > 170ll * 1000 || p->nx > 66) return false;
    }
    const hipError_t e = DISPATCHG(c, p->dx, p->dy, p->dz, pt_persist<T>(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, *p, 2, nullptr));
    if (e != hipSuccess) (void)hipGetLastError();
    return e == hipSuccess;                                         // the chip holds the whole grid at once
}
template <class T>
static bool use_graphs(const ns3d_ctx *c, const ns3d_pt_params *p, int nchk)
{
    if (use_persist<T>(c, p)) return false;                         // one launch per block already (and no capture of a cooperative launch)
    if (c->graph_mode == 0 || nchk < 4) return false;
    if (c->graph_mode > 0) return true;
    return (long long)p->nx * p->ny * p->nz < 3ll * 1000 * 1000;    // launch-bound regime
}

template <class T>
static int pt_solve_impl(ns3d_ctx *c, T *Pr, T *D, const T *divV, const ns3d_pt_params *p, double eps, int niter,
                         int nchk, double err_mul, double err_div, int *iters_done, double *err_hist, int max_checks,
                         int *n_checks)
{
    T *other = nullptr;
    int rc = ensure_pingpong<T>(c, p, &other);
    if (rc) return rc;
    const size_t plane = (size_t)p->nx * p->ny;
    T *src = Pr, *dst = other;
    int checks = 0, iter = 0, done = niter;
    bool converged = false;
    const bool two = use_two(c, p) && niter >= 2;
    T *dsrc = D, *ddst = nullptr;
    if (two && (rc = ensure_pingpong_d<T>(c, p, &ddst))) return rc;
    // graphs cannot be captured on HIP's null stream: run the loop on the context's own stream, fenced by events
    const bool graphs = use_graphs<T>(c, p, nchk);
    hipStream_t s = c->stream;
    const bool fenced = graphs && (s == nullptr);
    if (fenced) {
        s = c->own_stream;
        HIPCHK(c, hipEventRecord(c->fence, c->stream));
        HIPCHK(c, hipStreamWaitEvent(s, c->fence, 0));
    }
    // settle the tile choice of the two-iteration sweep now (eagerly, into the scratch buffers): inside a stream capture
    // it could only be looked up
    if (two) (void)pick_plan<T>(c, s, mode_of(c, p->dx, p->dy, p->dz), src, dst, dsrc, ddst, divV, p, 1, p->nz - 1, true);
    // Small grids (round 4): the WHOLE loop — iterations, residual checks, the decision of :467-469 — as one launch of k_pt_persist.  The
    // checks' maxima come back through pinned host memory behind one synchronisation; a launch whose bounded waits expired left its
    // inputs intact and the loop below redoes the solve by launches.  NS3D_PERSIST_SOLVE=0: a launch per residual-check block (A/B).
    const char *wv = std::getenv("NS3D_PERSIST_SOLVE");         // read per solve: tests flip it
    const bool whole = !(wv && *wv == '0');
    if (whole && nchk > 0 && niter >= nchk && niter / nchk <= NS3D_PERSIST_MAXCHK && use_persist<T>(c, p)) {
        T *dout = ddst;
        if (!dout && (rc = ensure_pingpong_d<T>(c, p, &dout))) return rc;
        hipError_t e = DISPATCHG(c, p->dx, p->dy, p->dz,
                                 pt_persist<T>(s, src, dst, dsrc, dout, divV, *p, niter, &c->persist, nchk, eps, err_mul, err_div));
        if (e == hipSuccess) {
            HIPCHK(c, hipStreamSynchronize(s));
            if (!persist_failed(c)) {
                const unsigned long long *res = c->persist.res_host;
                done = (int)res[0];
                checks = (int)res[1];
                for (int q = 0; q < checks && q < max_checks && err_hist; ++q) {
                    double mx;
                    std::memcpy(&mx, &res[2 + q], sizeof mx);
                    err_hist[q] = mx * err_mul / err_div;
                }
                src = dst; dsrc = dout;
                iter = niter;                   // nothing left for the loop below
            }
        } else if (e != hipErrorInvalidValue)
            return fail(NS3D_ERR_HIP, "pt_persist launch: %s", hipGetErrorString(e));
        else
            (void)hipGetLastError();
    }
    while (iter < niter) {
        // iterations until the next residual check (multi.jl:464) or the end of the budget
        const int n = nchk > 0 ? std::min(nchk - iter % nchk, niter - iter) : niter - iter;
        T *const in_src = src, *const in_dst = dst, *const in_dsrc = dsrc, *const in_ddst = ddst;     // the block's inputs
        const unsigned ticket0 = c->persist.ticket;
        if (graphs && n == nchk) {
            if ((rc = run_block_graph<T>(c, s, n, two, src, dst, dsrc, ddst, divV, p))) return rc;
        } else {
            hipError_t e = enqueue_iters<T>(c, s, n, two, src, dst, dsrc, ddst, divV, p, true);
            if (e != hipSuccess) return fail(NS3D_ERR_HIP, "pt_sweep launch: %s", hipGetErrorString(e));
        }
        iter += n;
        const bool check_now = nchk > 0 && iter % nchk == 0;
        if (c->persist.ticket != ticket0 && !check_now) HIPCHK(c, hipStreamSynchronize(s));     // a last, partial block: no read-back follows
        for (int attempt = 0; attempt < 2; ++attempt) {
            double mx = 0.0;
            if (check_now) { // multi.jl:464-469
                hipError_t e = DISPATCHG(c, p->dx, p->dy, p->dz, residual_max_key<T>(s, src, divV, *p, c->key_dev));
                if (e != hipSuccess) return fail(NS3D_ERR_HIP, "residual launch: %s", hipGetErrorString(e));
                if ((rc = fetch_key(c, s, &mx))) return rc;
            }
            // the read-back synchronised the stream: if the block ran as one cooperative launch and a hand-over in it timed out, its
            // inputs are still there — redo it by launches (once; the cooperative form is off for this context afterwards)
            if (attempt == 0 && c->persist.ticket != ticket0 && persist_failed(c)) {
                src = in_src; dst = in_dst; dsrc = in_dsrc; ddst = in_ddst;
                hipError_t e = enqueue_iters<T>(c, s, n, two, src, dst, dsrc, ddst, divV, p, true);
                if (e != hipSuccess) return fail(NS3D_ERR_HIP, "pt_sweep launch: %s", hipGetErrorString(e));
                continue;
            }
            if (!check_now) break;
            const double err = mx * err_mul / err_div; // maximum(abs.(Rp))*ly^2/psc, multi.jl:466
            if (err_hist && checks < max_checks) err_hist[checks] = err;
            ++checks;
            if (eps >= 0 && (err < eps || !std::isfinite(err))) { done = iter; converged = true; }
            break;
        }
        if (converged) break;
    }
    if (src != Pr)
        HIPCHK(c, hipMemcpyAsync(Pr, src, plane * p->nz * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (dsrc != D)
        HIPCHK(c, hipMemcpyAsync(D, dsrc, (size_t)(p->nx - 2) * (p->ny - 2) * (p->nz - 2) * sizeof(T),
                                 hipMemcpyDeviceToDevice, s));
    if (fenced) {
        HIPCHK(c, hipEventRecord(c->fence, s));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->fence, 0));
    }
    if (iters_done) *iters_done = done;
    if (n_checks) *n_checks = checks;
    return NS3D_OK;
}

#define NS3D_DEFINE(T, S)                                                                                    \
    extern "C" int ns3d_update_tau_##S(ns3d_ctx *c, T *txx, T *tyy, T *tzz, T *txy, T *txz, T *tyz,          \
                                       const T *Vx, const T *Vy, const T *Vz, double mu, double dx,          \
                                       double dy, double dz, int nx, int ny, int nz)                         \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 2);       \
        return finish(c, DISPATCHG(c, dx, dy, dz, update_tau<T>(c->stream, txx, tyy, tzz, txy, txz, tyz, Vx, Vy, Vz, mu,  \
                                                   dx, dy, dz, nx, ny, nz)), "update_tau");                  \
    }                                                                                                        \
    extern "C" int ns3d_predict_V_##S(ns3d_ctx *c, T *Vx, T *Vy, T *Vz, const T *txx, const T *tyy,          \
                                      const T *tzz, const T *txy, const T *txz, const T *tyz, double rho,    \
                                      double g, double dt, double dx, double dy, double dz, int nx, int ny,  \
                                      int nz)                                                                \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz); CHECK_GRID(nx, ny, nz, 2);       \
        return finish(c, DISPATCHG(c, dx, dy, dz, predict_V<T>(c->stream, Vx, Vy, Vz, txx, tyy, tzz, txy, txz, tyz, rho,  \
                                                  g, dt, dx, dy, dz, nx, ny, nz)), "predict_V");             \
    }                                                                                                        \
    extern "C" int ns3d_predict_fused_##S(ns3d_ctx *c, T *Vx_new, T *Vy_new, T *Vz_new, const T *Vx, const T *Vy, \
                                          const T *Vz, double mu, double rho, double g, double dt, double dx, \
                                          double dy, double dz, int nx, int ny, int nz)                      \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx_new, Vy_new, Vz_new, Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 2);             \
        if (Vx_new == Vx || Vy_new == Vy || Vz_new == Vz)                                                    \
            return fail(NS3D_ERR_ARG, "ns3d_predict_fused: the predicted velocities need buffers of their own"); \
        return finish(c, DISPATCHG(c, dx, dy, dz, predict_fused<T>(c->stream, Vx_new, Vy_new, Vz_new, Vx, Vy, Vz, mu, rho, \
                                                  g, dt, dx, dy, dz, nx, ny, nz)), "predict_fused");         \
    }                                                                                                        \
    extern "C" int ns3d_set_cylinder_##S(ns3d_ctx *c, T *C, T *Vx, T *Vy, T *Vz, double a2, double b2,       \
                                         double ox, double oy, double sinb, double cosb, double xco_g,       \
                                         double yco_g, double zco_g, double lx, double ly, double lz,        \
                                         double dx, double dy, double dz, int nx, int ny, int nz)            \
    {                                                                                                        \
        (void)zco_g; (void)lz; (void)dz;                                                                     \
        CHECK_CTX(c); CHECK_PTRS(C, Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 1);                                  \
        return finish(c, DISPATCH(c, set_cylinder<T>(c->stream, C, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb,   \
                                                     0, xco_g, yco_g, lx, ly, dx, dy, nx, ny, nz)),          \
                      "set_cylinder");                                                                       \
    }                                                                                                        \
    extern "C" int ns3d_set_cylinder_local_##S(ns3d_ctx *c, T *C, T *Vx, T *Vy, T *Vz, double a2, double b2, \
                                               double ox, double oy, double sinb, double cosb, double lx,    \
                                               double ly, double lz, double dx, double dy, double dz,        \
                                               int nx, int ny, int nz)                                       \
    {                                                                                                        \
        (void)lz; (void)dz;                                                                                  \
        CHECK_CTX(c); CHECK_PTRS(C, Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 1);                                  \
        return finish(c, DISPATCH(c, set_cylinder<T>(c->stream, C, Vx, Vy, Vz, a2, b2, ox, oy, sinb, cosb,   \
                                                     1, 0.0, 0.0, lx, ly, dx, dy, nx, ny, nz)),              \
                      "set_cylinder_local");                                                                 \
    }                                                                                                        \
    extern "C" int ns3d_update_divV_##S(ns3d_ctx *c, T *divV, const T *Vx, const T *Vy, const T *Vz,         \
                                        double dx, double dy, double dz, int nx, int ny, int nz)             \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(divV, Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 1);                               \
        return finish(c, DISPATCHG(c, dx, dy, dz, update_divV<T>(c->stream, divV, Vx, Vy, Vz, dx, dy, dz, nx, ny, nz)),   \
                      "update_divV");                                                                        \
    }                                                                                                        \
    extern "C" int ns3d_update_dPrdtau_##S(ns3d_ctx *c, const T *Pr, T *dPrdtau, const T *divV, double rho,  \
                                           double dt, double dtau, double damp, double dx, double dy,        \
                                           double dz, int nx, int ny, int nz)                                \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, dPrdtau, divV); CHECK_GRID(nx, ny, nz, 3);                              \
        return finish(c, DISPATCHG(c, dx, dy, dz, update_dPrdtau<T>(c->stream, Pr, dPrdtau, divV, rho, dt, dtau, damp,    \
                                                       dx, dy, dz, nx, ny, nz)), "update_dPrdtau");          \
    }                                                                                                        \
    extern "C" int ns3d_update_Pr_##S(ns3d_ctx *c, T *Pr, const T *dPrdtau, double dtau, int nx, int ny,     \
                                      int nz)                                                                \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, dPrdtau); CHECK_GRID(nx, ny, nz, 3);                                    \
        return finish(c, DISPATCH(c, update_Pr<T>(c->stream, Pr, dPrdtau, dtau, nx, ny, nz)), "update_Pr");  \
    }                                                                                                        \
    extern "C" int ns3d_compute_res_##S(ns3d_ctx *c, T *Rp, const T *Pr, const T *divV, double rho,          \
                                        double dt, double dx, double dy, double dz, int nx, int ny, int nz)  \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Rp, Pr, divV); CHECK_GRID(nx, ny, nz, 3);                                   \
        return finish(c, DISPATCHG(c, dx, dy, dz, compute_res<T>(c->stream, Rp, Pr, divV, rho, dt, dx, dy, dz, nx, ny,    \
                                                    nz)), "compute_res");                                    \
    }                                                                                                        \
    extern "C" int ns3d_max_abs_##S(ns3d_ctx *c, const T *A, long n, double *out_host)                       \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(A, out_host);                                                               \
        if (n < 0) return fail(NS3D_ERR_ARG, "ns3d_max_abs: negative length");                               \
        hipError_t e = DISPATCH(c, max_abs_key<T>(c->stream, A, n, c->key_dev));                             \
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "max_abs launch: %s", hipGetErrorString(e));          \
        return fetch_key(c, c->stream, out_host);                                                                       \
    }                                                                                                        \
    extern "C" int ns3d_correct_V_##S(ns3d_ctx *c, T *Vx, T *Vy, T *Vz, const T *Pr, double dt, double rho,  \
                                      double dx, double dy, double dz, int nx, int ny, int nz)               \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx, Vy, Vz, Pr); CHECK_GRID(nx, ny, nz, 2);                                 \
        return finish(c, DISPATCHG(c, dx, dy, dz, correct_V<T>(c->stream, Vx, Vy, Vz, Pr, dt, rho, dx, dy, dz, nx, ny,    \
                                                  nz)), "correct_V");                                        \
    }                                                                                                        \
    static int bc_##S(ns3d_ctx *c, int which, T *A, int sx, int sy, int sz, double a, double b, int nz_arg,  \
                      const char *name)                                                                      \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(A);                                                                         \
        if (sx < 2 || sy < 2 || sz < 2) return fail(NS3D_ERR_ARG, "%s: extents %dx%dx%d too small", name,    \
                                                    sx, sy, sz);                                             \
        return finish(c, DISPATCH(c, bc_plane<T>(c->stream, which, A, sx, sy, sz, a, b, 0.0, nz_arg)), name);\
    }                                                                                                        \
    extern "C" int ns3d_bc_x_##S(ns3d_ctx *c, T *A, int sx, int sy, int sz) { return bc_##S(c, 0, A, sx, sy, sz, 0, 0, 0, "bc_x"); } \
    extern "C" int ns3d_bc_y_##S(ns3d_ctx *c, T *A, int sx, int sy, int sz) { return bc_##S(c, 1, A, sx, sy, sz, 0, 0, 0, "bc_y"); } \
    extern "C" int ns3d_bc_z_##S(ns3d_ctx *c, T *A, int sx, int sy, int sz) { return bc_##S(c, 2, A, sx, sy, sz, 0, 0, 0, "bc_z"); } \
    extern "C" int ns3d_bc_zV_##S(ns3d_ctx *c, T *A, int sx, int sy, int sz) { return bc_##S(c, 3, A, sx, sy, sz, 0, 0, 0, "bc_zV"); } \
    extern "C" int ns3d_bc_xhydstatic_##S(ns3d_ctx *c, T *A, double dz, int nz, double g, double rho,        \
                                          int sx, int sy, int sz)                                            \
    {                                                                                                        \
        return bc_##S(c, 4, A, sx, sy, sz, (double)((T)rho * (T)g), dz, nz, "bc_xhydstatic");                \
    }                                                                                                        \
    extern "C" int ns3d_bc_x_Vx_##S(ns3d_ctx *c, T *A, double V, int sx, int sy, int sz) { return bc_##S(c, 5, A, sx, sy, sz, V, 0, 0, "bc_x_Vx"); } \
    extern "C" int ns3d_bc_x_Pr_##S(ns3d_ctx *c, T *A, double v, int sx, int sy, int sz) { return bc_##S(c, 6, A, sx, sy, sz, v, 0, 0, "bc_x_Pr"); } \
    extern "C" int ns3d_copy_##S(ns3d_ctx *c, T *dst, const T *src, long n)                                  \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(dst, src);                                                                  \
        if (n < 0) return fail(NS3D_ERR_ARG, "ns3d_copy: negative length");                                  \
        return finish(c, hipMemcpyAsync(dst, src, (size_t)n * sizeof(T), hipMemcpyDeviceToDevice, c->stream),\
                      "copy");                                                                               \
    }                                                                                                        \
    extern "C" int ns3d_advect_##S(ns3d_ctx *c, T *Vx, const T *Vx_o, T *Vy, const T *Vy_o, T *Vz,           \
                                   const T *Vz_o, T *C, const T *C_o, double dt, double dx, double dy,       \
                                   double dz, int nx, int ny, int nz, int faithful)                          \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o); CHECK_GRID(nx, ny, nz, 1);           \
        return finish(c, DISPATCHG(c, dx, dy, dz, advect<T>(c->stream, Vx, Vx_o, Vy, Vy_o, Vz, Vz_o, C, C_o, dt, dx, dy,  \
                                               dz, nx, ny, nz, faithful ? 1 : 0, 0, 0)), "advect");                        \
    }                                                                                                        \
    extern "C" int ns3d_copy_advect_##S(ns3d_ctx *c, T *Vx_new, const T *Vx, T *Vy_new, const T *Vy, T *Vz_new, \
                                        const T *Vz, T *C_new, const T *C, double dt, double dx, double dy,  \
                                        double dz, int nx, int ny, int nz, int faithful)                     \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx_new, Vx, Vy_new, Vy, Vz_new, Vz, C_new, C); CHECK_GRID(nx, ny, nz, 1);   \
        if (Vx_new == Vx || Vy_new == Vy || C_new == C || (!faithful && Vz_new == Vz))                       \
            return fail(NS3D_ERR_ARG, "ns3d_copy_advect: outputs must be buffers of their own (only Vz_new may be Vz, and only " \
                                      "in faithful mode, where Vz is never advected)");                      \
        return finish(c, DISPATCHG(c, dx, dy, dz, advect<T>(c->stream, Vx_new, Vx, Vy_new, Vy, Vz_new, Vz, C_new, C, dt, dx, dy, \
                                               dz, nx, ny, nz, (faithful ? 1 : 0) | 2, 0, 0)), "copy_advect");     \
    }                                                                                                        \
    extern "C" int ns3d_set_bc_Pr_##S(ns3d_ctx *c, T *Pr, int bc_kind, int owns_outlet, double outlet_val,   \
                                      double dz, int nz_arg, double g, double rho, int nx, int ny, int nz)   \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr); CHECK_GRID(nx, ny, nz, 2);                                             \
        hipError_t e = hipSuccess;                                                                           \
        hipStream_t s = c->stream;                                                                           \
        if (bc_kind != NS3D_BC_MULTI && bc_kind != NS3D_BC_GPU) return fail(NS3D_ERR_ARG, "ns3d_set_bc_Pr: bad bc_kind %d", bc_kind); \
        if (bc_fused_enabled()) {       /* the whole sequence as one gather launch (k_bc_fused) */          \
            e = DISPATCH(c, bc_fused<T>(s, 1, bc_kind, Pr, (T *)nullptr, (T *)nullptr, nx, ny, nz, owns_outlet, outlet_val, \
                                        (double)((T)rho * (T)g), dz, nz_arg));                               \
            if (e != hipErrorInvalidValue) return finish(c, e, "set_bc_Pr");                                 \
            (void)hipGetLastError();                                                                         \
            e = hipSuccess;                                                                                  \
        }                                                                                                    \
        if (bc_kind == NS3D_BC_MULTI) { /* multi.jl:176-181 */                                               \
            e = DISPATCH(c, bc_plane<T>(s, 0, Pr, nx, ny, nz, 0, 0, 0, 0));                                  \
            if (e == hipSuccess) e = DISPATCH(c, bc_plane<T>(s, 1, Pr, nx, ny, nz, 0, 0, 0, 0));             \
            if (e == hipSuccess) e = DISPATCH(c, bc_plane<T>(s, 2, Pr, nx, ny, nz, 0, 0, 0, 0));             \
            if (e == hipSuccess && owns_outlet) e = DISPATCH(c, bc_plane<T>(s, 6, Pr, nx, ny, nz, outlet_val, 0, 0, 0)); \
        } else if (bc_kind == NS3D_BC_GPU) { /* gpu.jl:282-284 */                                            \
            e = DISPATCH(c, bc_plane<T>(s, 1, Pr, nx, ny, nz, 0, 0, 0, 0));                                  \
            if (e == hipSuccess) e = DISPATCH(c, bc_plane<T>(s, 2, Pr, nx, ny, nz, 0, 0, 0, 0));             \
            if (e == hipSuccess) e = DISPATCH(c, bc_plane<T>(s, 4, Pr, nx, ny, nz, (double)((T)rho * (T)g), dz, 0, nz_arg)); \
        } else return fail(NS3D_ERR_ARG, "ns3d_set_bc_Pr: bad bc_kind %d", bc_kind);                         \
        return finish(c, e, "set_bc_Pr");                                                                    \
    }                                                                                                        \
    extern "C" int ns3d_set_bc_Vel_##S(ns3d_ctx *c, T *Vx, T *Vy, T *Vz, int bc_kind, int owns_inlet,        \
                                       double vin, int nx, int ny, int nz)                                   \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Vx, Vy, Vz); CHECK_GRID(nx, ny, nz, 2);                                     \
        hipStream_t s = c->stream;                                                                           \
        hipError_t e = hipSuccess;                                                                           \
        if (bc_kind != NS3D_BC_MULTI && bc_kind != NS3D_BC_GPU) return fail(NS3D_ERR_ARG, "ns3d_set_bc_Vel: bad bc_kind %d", bc_kind); \
        if (bc_fused_enabled()) {       /* the whole sequence as one gather launch (k_bc_fused) */          \
            e = DISPATCH(c, bc_fused<T>(s, 0, bc_kind, Vx, Vy, Vz, nx, ny, nz, bc_kind == NS3D_BC_MULTI && owns_inlet, vin, 0.0, 0.0, 0)); \
            if (e != hipErrorInvalidValue) return finish(c, e, "set_bc_Vel");                                \
            (void)hipGetLastError();                                                                         \
            e = hipSuccess;                                                                                  \
        }                                                                                                    \
        struct { int which; T *A; int sx, sy, sz; } seq[9];                                                  \
        int n = 0;                                                                                           \
        if (bc_kind == NS3D_BC_MULTI) { /* multi.jl:157-163 */                                               \
            seq[n++] = {0, Vx, nx + 1, ny, nz}; seq[n++] = {1, Vx, nx + 1, ny, nz}; seq[n++] = {2, Vx, nx + 1, ny, nz}; \
            seq[n++] = {0, Vy, nx, ny + 1, nz}; seq[n++] = {2, Vy, nx, ny + 1, nz};                          \
            seq[n++] = {0, Vz, nx, ny, nz + 1}; seq[n++] = {1, Vz, nx, ny, nz + 1};                          \
        } else if (bc_kind == NS3D_BC_GPU) { /* gpu.jl:265-276 */                                            \
            seq[n++] = {0, Vx, nx + 1, ny, nz}; seq[n++] = {1, Vx, nx + 1, ny, nz}; seq[n++] = {3, Vx, nx + 1, ny, nz}; \
            seq[n++] = {0, Vy, nx, ny + 1, nz}; seq[n++] = {1, Vy, nx, ny + 1, nz}; seq[n++] = {3, Vy, nx, ny + 1, nz}; \
            seq[n++] = {0, Vz, nx, ny, nz + 1}; seq[n++] = {1, Vz, nx, ny, nz + 1}; seq[n++] = {3, Vz, nx, ny, nz + 1}; \
        } else return fail(NS3D_ERR_ARG, "ns3d_set_bc_Vel: bad bc_kind %d", bc_kind);                        \
        for (int q = 0; q < n && e == hipSuccess; ++q)                                                       \
            e = DISPATCH(c, bc_plane<T>(s, seq[q].which, seq[q].A, seq[q].sx, seq[q].sy, seq[q].sz, 0, 0, 0, 0)); \
        if (e == hipSuccess && bc_kind == NS3D_BC_MULTI && owns_inlet) /* multi.jl:164-166 */                \
            e = DISPATCH(c, bc_plane<T>(s, 5, Vx, nx + 1, ny, nz, vin, 0, 0, 0));                            \
        return finish(c, e, "set_bc_Vel");                                                                   \
    }                                                                                                        \
    extern "C" int ns3d_pt_iterate_##S(ns3d_ctx *c, T *Pr, T *dPrdtau, const T *divV,                        \
                                       const ns3d_pt_params *p, int n_iters)                                 \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, dPrdtau, divV);                                                         \
        int rc = ns3d_check_pt_params(p, "ns3d_pt_iterate");                                                      \
        if (rc) return rc;                                                                                   \
        if ((rc = pt_iterate_impl<T>(c, Pr, dPrdtau, divV, p, n_iters))) return rc;                          \
        return finish(c, hipSuccess, "pt_iterate");                                                          \
    }                                                                                                        \
    extern "C" int ns3d_pt_sweep_##S(ns3d_ctx *c, const T *Pr_in, T *Pr_out, T *dPrdtau, const T *divV,      \
                                     const ns3d_pt_params *p, int k0, int k1)                                \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr_in, Pr_out, dPrdtau, divV);                                              \
        int rc = ns3d_check_pt_params(p, "ns3d_pt_sweep");                                                        \
        if (rc) return rc;                                                                                   \
        if (Pr_in == Pr_out) return fail(NS3D_ERR_ARG, "ns3d_pt_sweep: Pr_in and Pr_out must differ");       \
        if (k0 < 1 || k1 > p->nz - 1 || k0 > k1)                                                             \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweep: plane range [%d,%d) outside [1,%d)", k0, k1, p->nz - 1);\
        return finish(c, DISPATCHG(c, p->dx, p->dy, p->dz, pt_sweep<T>(c->stream, c->pt_variant, Pr_in, Pr_out, dPrdtau, divV, *p, \
                                                 k0, k1)), "pt_sweep");                                      \
    }                                                                                                        \
    extern "C" int ns3d_pt_sweep2_##S(ns3d_ctx *c, const T *Pr_in, T *Pr_out, const T *dPrdtau, T *dPrdtau_out,\
                                      const T *divV, const ns3d_pt_params *p, int k0, int k1)               \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV);                                 \
        if (dPrdtau == dPrdtau_out) return fail(NS3D_ERR_ARG, "ns3d_pt_sweep2: dPrdtau_in and dPrdtau_out must differ"); \
        int rc = ns3d_check_pt_params(p, "ns3d_pt_sweep2");                                                       \
        if (rc) return rc;                                                                                   \
        if (Pr_in == Pr_out) return fail(NS3D_ERR_ARG, "ns3d_pt_sweep2: Pr_in and Pr_out must differ");      \
        if (p->z_lo_is_halo || p->z_hi_is_halo)                                                              \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweep2: z-slab ranks pass ghost-extended buffers, not halo flags"); \
        if (k0 < 1 || k1 > p->nz - 1 || k0 > k1)                                                             \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweep2: plane range [%d,%d) outside [1,%d)", k0, k1, p->nz - 1); \
        return finish(c, launch_pt2<T>(c, c->stream, Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV, p, k0, k1, false), "pt_sweep2"); \
    }                                                                                                        \
    extern "C" int ns3d_pt_sweepn_##S(ns3d_ctx *c, int nlev, const T *Pr_in, T *Pr_out, const T *dPrdtau,  \
                                      T *dPrdtau_out, const T *divV, const ns3d_pt_params *p, int k0, int k1)\
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV);                                 \
        int rc = ns3d_check_pt_params(p, "ns3d_pt_sweepn");                                                  \
        if (rc) return rc;                                                                                   \
        if (nlev < 2 || nlev > (sizeof(T) == 4 ? 5 : 4))                                                     \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweepn: %d levels (2…4; 5 with float32 fields)", nlev);       \
        if (Pr_in == Pr_out || dPrdtau == dPrdtau_out)                                                       \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweepn: input and output buffers must differ");               \
        if (p->z_lo_is_halo || p->z_hi_is_halo)                                                              \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweepn: z-slab ranks pass ghost-extended buffers, not halo flags"); \
        if (k0 < 1 || k1 > p->nz - 1 || k0 > k1)                                                             \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweepn: plane range [%d,%d) outside [1,%d)", k0, k1, p->nz - 1); \
        hipError_t e = ns3d_enqueue_pass<T>(c, c->stream, nlev, Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV, p, k0, k1); \
        if (e == hipErrorInvalidValue)                                                                       \
            return fail(NS3D_ERR_ARG, "ns3d_pt_sweepn: tile variant %d cannot run %d levels", c->ptn_variant, nlev); \
        return finish(c, e, "pt_sweepn");                                                                    \
    }                                                                                                        \
    extern "C" int ns3d_plan_pt_##S(ns3d_ctx *c, const T *Pr_in, T *Pr_out, const T *dPrdtau, T *dPrdtau_out, \
                                    const T *divV, const ns3d_pt_params *p, int k0, int k1)                  \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV);                                 \
        int rc = ns3d_check_pt_params(p, "ns3d_plan_pt");                                                    \
        if (rc) return rc;                                                                                   \
        if (Pr_in == Pr_out || dPrdtau == dPrdtau_out)                                                       \
            return fail(NS3D_ERR_ARG, "ns3d_plan_pt: input and output buffers must differ");                 \
        if (k0 < 1 || k1 > p->nz - 1 || k0 > k1)                                                             \
            return fail(NS3D_ERR_ARG, "ns3d_plan_pt: plane range [%d,%d) outside [1,%d)", k0, k1, p->nz - 1);\
        (void)ns3d_plan_pt_internal<T>(c, Pr_in, Pr_out, dPrdtau, dPrdtau_out, divV, p, k0, k1);             \
        return finish(c, hipSuccess, "plan_pt");                                                             \
    }                                                                                                        \
    extern "C" int ns3d_residual_max_##S(ns3d_ctx *c, const T *Pr, const T *divV, const ns3d_pt_params *p,   \
                                         double *out_host)                                                   \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, divV, out_host);                                                        \
        int rc = ns3d_check_pt_params(p, "ns3d_residual_max");                                                    \
        if (rc) return rc;                                                                                   \
        hipError_t e = DISPATCHG(c, p->dx, p->dy, p->dz, residual_max_key<T>(c->stream, Pr, divV, *p, c->key_dev));                \
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "residual launch: %s", hipGetErrorString(e));         \
        return fetch_key(c, c->stream, out_host);                                                                       \
    }                                                                                                        \
    extern "C" int ns3d_selftest_exact_div_##S(ns3d_ctx *c, double d, long n, unsigned long long seed,        \
                                               long *mismatches)                                             \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(mismatches);                                                                \
        if (!recip_ok(d)) return fail(NS3D_ERR_ARG, "ns3d_selftest_exact_div: divisor %g is not eligible", d);\
        hipError_t e = ns3d_strictx::divtest<T>(c->stream, d, n, seed, c->key_dev);                          \
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "divtest launch: %s", hipGetErrorString(e));          \
        HIPCHK(c, hipMemcpyAsync(c->key_host, c->key_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, \
                                 c->stream));                                                                \
        HIPCHK(c, hipStreamSynchronize(c->stream));                                                          \
        *mismatches = (long)*c->key_host;                                                                    \
        return NS3D_OK;                                                                                      \
    }                                                                                                        \
    extern "C" int ns3d_pt_solve_##S(ns3d_ctx *c, T *Pr, T *dPrdtau, const T *divV, const ns3d_pt_params *p, \
                                     double eps, int niter, int nchk, double err_mul, double err_div,        \
                                     int *iters_done, double *err_hist, int max_checks, int *n_checks)       \
    {                                                                                                        \
        CHECK_CTX(c); CHECK_PTRS(Pr, dPrdtau, divV);                                                         \
        int rc = ns3d_check_pt_params(p, "ns3d_pt_solve");                                                        \
        if (rc) return rc;                                                                                   \
        if (p->z_lo_is_halo || p->z_hi_is_halo)                                                              \
            return fail(NS3D_ERR_ARG, "ns3d_pt_solve: single-rank loop; drive z-slab ranks with ns3d_pt_sweep + halo exchange"); \
        if (niter < 0 || nchk < 0) return fail(NS3D_ERR_ARG, "ns3d_pt_solve: negative niter/nchk");          \
        if ((rc = pt_solve_impl<T>(c, Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul, err_div, iters_done,  \
                                   err_hist, max_checks, n_checks))) return rc;                              \
        return finish(c, hipSuccess, "pt_solve");                                                            \
    }

NS3D_DEFINE(double, f64)
NS3D_DEFINE(float, f32)

// ---- one whole time step per call (include/ns3d.h ns3d_time_step) -----------------------------------------------------------
// multi.jl:449-477 on one rank / gpu.jl:121-142 as the fused sequence the Python driver issues call by call — through the SAME entry
// points (argument checks, arithmetic builds and all), with the context switched to non-blocking for the duration so that the residual
// read-backs inside ns3d_pt_solve are the only synchronisations; one synchronisation at the end restores a blocking context's
// contract.  ≈40 host calls per step become one: with the direct pressure solve a step of the 255×153×153 case is otherwise mostly
// the driver's calls.
extern "C" int ns3d_poisson_direct_f64(ns3d_ctx *, double *, double *, const double *, const ns3d_pt_params *);
extern "C" int ns3d_poisson_direct_f32(ns3d_ctx *, float *, float *, const float *, const ns3d_pt_params *);
#define NS3D_DEFINE_STEP(T, S)                                                                                \
    extern "C" int ns3d_time_step_##S(ns3d_ctx *c, ns3d_step_fields *f, const ns3d_step_params *p, int *iters_done, \
                                      double *err_hist, int max_checks, int *n_checks)                       \
    {                                                                                                        \
        CHECK_CTX(c);                                                                                        \
        if (!f || !p) return fail(NS3D_ERR_ARG, "ns3d_time_step: null argument");                            \
        if (p->script != NS3D_BC_MULTI && p->script != NS3D_BC_GPU)                                          \
            return fail(NS3D_ERR_ARG, "ns3d_time_step: script %d (NS3D_BC_MULTI: multi.jl, NS3D_BC_GPU: gpu.jl)", p->script); \
        if (p->write_stress && !(f->txx && f->tyy && f->tzz && f->txy && f->txz && f->tyz))                  \
            return fail(NS3D_ERR_ARG, "ns3d_time_step: write_stress needs the six stress arrays");           \
        const int nx = p->nx, ny = p->ny, nz = p->nz;                                                        \
        const int was = c->flags;                                                                            \
        c->flags |= NS3D_ASYNC;                                                                              \
        int rc = NS3D_OK;                                                                                    \
        bool deferred_residual = false;                                                                      \
        T *Vx = (T *)f->Vx, *Vy = (T *)f->Vy, *Vz = (T *)f->Vz, *Vxo = (T *)f->Vx_o, *Vyo = (T *)f->Vy_o, *Vzo = (T *)f->Vz_o; \
        T *C = (T *)f->C, *Co = (T *)f->C_o, *Pr = (T *)f->Pr, *D = (T *)f->dPrdtau, *divV = (T *)f->divV;    \
        auto cylinder = [&]() -> int {          /* multi.jl:249-281 (global coordinates) / gpu.jl:336-368 (local) */ \
            return p->script == NS3D_BC_MULTI                                                                \
                       ? ns3d_set_cylinder_##S(c, C, Vx, Vy, Vz, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->xco_g, p->yco_g, \
                                               p->zco_g, p->lx, p->ly, p->lz, p->dx, p->dy, p->dz, nx, ny, nz)  \
                       : ns3d_set_cylinder_local_##S(c, C, Vx, Vy, Vz, p->a2, p->b2, p->ox, p->oy, p->sinb, p->cosb, p->lx,    \
                                                     p->ly, p->lz, p->dx, p->dy, p->dz, nx, ny, nz);            \
        };                                                                                                   \
        do {                                                                                                 \
            if (p->write_stress &&                                                                           \
                (rc = ns3d_update_tau_##S(c, (T *)f->txx, (T *)f->tyy, (T *)f->tzz, (T *)f->txy, (T *)f->txz, (T *)f->tyz, Vx, Vy, Vz, \
                                          p->mu, p->dx, p->dy, p->dz, nx, ny, nz))) break;                   \
            /* :449-451 / :121-122 in one pass; the predicted fields land in the *_o buffers and the names swap */ \
            if ((rc = ns3d_predict_fused_##S(c, Vxo, Vyo, Vzo, Vx, Vy, Vz, p->mu, p->rho, p->g, p->dt, p->dx, p->dy, p->dz, nx, ny, nz))) break; \
            std::swap(Vx, Vxo); std::swap(Vy, Vyo); std::swap(Vz, Vzo);                                      \
            if ((rc = cylinder())) break;                                                   /* :452 / :123 */ \
            if ((rc = ns3d_update_divV_##S(c, divV, Vx, Vy, Vz, p->dx, p->dy, p->dz, nx, ny, nz))) break;     /* :454 / :124 */ \
            ns3d_pt_params pt;                                                                               \
            pt.rho = p->rho; pt.dt = p->dt; pt.dtau = p->dtau; pt.damp = p->damp; pt.dx = p->dx; pt.dy = p->dy; pt.dz = p->dz; \
            pt.nx = nx; pt.ny = ny; pt.nz = nz; pt.bc_kind = p->script; pt.owns_outlet = p->script == NS3D_BC_MULTI ? p->owns_outlet : 0; \
            pt.outlet_val = 0.0; pt.g = p->g; pt.z_lo_is_halo = 0; pt.z_hi_is_halo = 0;                        \
            if (p->pressure == 1) {             /* outside parity: the exact solution of what :458-471 iterates towards */ \
                if ((rc = ns3d_poisson_direct_##S(c, Pr, D, divV, &pt))) break;                              \
                /* its residual steers nothing: the 8 bytes come back behind the REST of the step (read after the final            \
                 * synchronisation) instead of stalling the stream in the middle of it */                     \
                hipError_t e = DISPATCHG(c, p->dx, p->dy, p->dz, residual_max_key<T>(c->stream, Pr, divV, pt, c->key_dev)); \
                if (e == hipSuccess) e = hipMemcpyAsync(c->key_host, c->key_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream); \
                if (e != hipSuccess) { rc = fail(NS3D_ERR_HIP, "ns3d_time_step: residual launch: %s", hipGetErrorString(e)); break; } \
                deferred_residual = true;                                                                    \
            } else if ((rc = ns3d_pt_solve_##S(c, Pr, D, divV, &pt, p->eps, p->niter, p->nchk, p->err_mul, p->err_div, iters_done, \
                                               err_hist, max_checks, n_checks))) break;     /* :458-471 / :126-137 */ \
            if ((rc = ns3d_correct_V_##S(c, Vx, Vy, Vz, Pr, p->dt, p->rho, p->dx, p->dy, p->dz, nx, ny, nz))) break;   /* :472 / :138 */ \
            if ((rc = cylinder())) break;                                                   /* :473 / :139 */ \
            if ((rc = ns3d_set_bc_Vel_##S(c, Vx, Vy, Vz, p->script, p->script == NS3D_BC_MULTI ? p->owns_inlet : 0, p->vin, nx, ny, nz))) break; \
            /* :475-476 / :141-142 in one pass: complete new fields into the *_o buffers, then the roles swap */ \
            if ((rc = ns3d_copy_advect_##S(c, Vxo, Vx, Vyo, Vy, p->faithful ? Vz : Vzo, Vz, Co, C, p->dt, p->dx, p->dy, p->dz, nx, ny, nz, \
                                           p->faithful ? 1 : 0))) break;                                     \
            std::swap(Vx, Vxo); std::swap(Vy, Vyo); std::swap(C, Co);                                        \
            if (!p->faithful) std::swap(Vz, Vzo);                                                            \
        } while (0);                                                                                         \
        c->flags = was;                                                                                      \
        f->Vx = Vx; f->Vy = Vy; f->Vz = Vz; f->Vx_o = Vxo; f->Vy_o = Vyo; f->Vz_o = Vzo; f->C = C; f->C_o = Co; \
        if (rc) return rc;                                                                                   \
        if (deferred_residual) {                                                                             \
            HIPCHK(c, hipStreamSynchronize(c->stream));                                                      \
            double mx;                                                                                       \
            std::memcpy(&mx, c->key_host, sizeof mx);                                                        \
            if (iters_done) *iters_done = 0;                                                                 \
            if (err_hist && max_checks > 0) err_hist[0] = mx * p->err_mul / p->err_div;                      \
            if (n_checks) *n_checks = 1;                                                                     \
            return NS3D_OK;                                                                                  \
        }                                                                                                    \
        return finish(c, hipSuccess, "time_step");                                                           \
    }
NS3D_DEFINE_STEP(double, f64)
NS3D_DEFINE_STEP(float, f32)
