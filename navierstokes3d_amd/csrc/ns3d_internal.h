// ns3d_internal.h — what the host-side translation units of libns3d.so share (ns3d_api.cpp: single-device entry points
// and the PT loop; ns3d_mgpu.cpp: the z-slab multi-GPU layer).  Internal; the public boundary is include/ns3d.h.
#pragma once
#include <cstring>
#include <vector>

#include "ns3d_launch.h"

// records the message behind ns3d_last_error() and returns `code`
int ns3d_fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail ns3d_fail

struct ns3d_ctx {
    int device;
    int flags;
    hipStream_t own_stream;
    hipStream_t masked_stream = nullptr;  // ns3d_reserve_cus: a stream whose CU mask leaves `reserved_cus` compute units out
    int reserved_cus = 0;
    hipStream_t stream;
    unsigned long long *key_dev;  // device scratch for max reductions
    unsigned long long *key_host; // pinned host mirror
    void *pingpong;               // second Pr buffer of the fused PT path (lazily sized)
    size_t pingpong_bytes;
    void *pingpong_d;             // second dPrdτ buffer (temporal blocking only)
    size_t pingpong_d_bytes;
    int pt_variant;
    int pt2_variant; // tile shape of the two-iteration sweep; <0: temporal blocking off
    int ptn_variant; // tile shape of the N-iteration sweep (k_pt_sweepN); 0: built-in
    int pt_depth;    // PT iterations per pass in pt_iterate / pt_solve: 0 automatic, 1…4 forced
    int graph_mode;  // HIP-graph replay of residual-check blocks: -1 auto (launch-bound grids), 0 off, 1 on
    int persist_mode = -1; // k_pt_persist (a whole residual-check block in one cooperative launch): -1 auto (small grids), 0 off, 1 on
    int autotune;    // time the tile shapes of the two-iteration sweep on first use of a grid (pt2_variant == 0 only)
    int last_pt2;    // variant of the latest two-iteration launch (0: built-in choice by grid)
    int last_ptn;    // variant of the latest N-iteration launch
    int last_depth;  // PT iterations of the latest multi-iteration pass
    hipEvent_t tune_ev[2];
    hipEvent_t fence;
    struct BlockGraph {
        const void *src, *dst, *dsrc, *ddst, *rhs;
        void *src_out, *dst_out, *dsrc_out, *ddst_out;
        int n, mode, v1, v2, vn, depth, esize;
        bool two;
        ns3d_pt_params p;
        hipGraphExec_t exec;
    };
    std::vector<BlockGraph> graphs;
    ns3d_persist_state persist;             // k_pt_persist's exchange area
    void *direct_plan = nullptr;            // ns3d_direct.hip: eigenvector matrices and scratch of the direct Poisson solve
    void (*direct_free)(void *) = nullptr;
    void clear_graphs()
    {
        for (auto &g : graphs) (void)hipGraphExecDestroy(g.exec);
        graphs.clear();
    }
};

// Every entry point runs with its context's device current and puts the caller's device back on return: a process
// that drives several devices (ns3d_mgpu_create, or PyTorch beside us) keeps its own notion of "current device".
struct ns3d_device_guard {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit ns3d_device_guard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != device) {
            err = hipSetDevice(device);
            if (err != hipSuccess) { (void)hipGetLastError(); prev = -1; }   // leave no sticky error behind for the host framework
        } else prev = -1;                    // nothing to restore
    }
    ~ns3d_device_guard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    ns3d_device_guard(const ns3d_device_guard &) = delete;
    ns3d_device_guard &operator=(const ns3d_device_guard &) = delete;
};

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) {                                                                             \
            (void)hipGetLastError(); /* reported through our status: leave no sticky error for the host framework */ \
            return fail(NS3D_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_));                              \
        }                                                                                                   \
    } while (0)

#define CHECK_CTX(ctx)                                                                                      \
    if (!(ctx)) return fail(NS3D_ERR_ARG, "%s: null context", __func__);                                    \
    ns3d_device_guard dev_guard_((ctx)->device);                                                            \
    if (dev_guard_.err != hipSuccess) return fail(NS3D_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(dev_guard_.err))

#define CHECK_PTRS(...)                                                                                     \
    do {                                                                                                    \
        const void *ps_[] = {__VA_ARGS__};                                                                  \
        for (size_t q_ = 0; q_ < sizeof ps_ / sizeof ps_[0]; ++q_)                                          \
            if (!ps_[q_]) return fail(NS3D_ERR_ARG, "%s: null field pointer (argument %zu)", __func__, q_); \
    } while (0)

#define CHECK_GRID(nx, ny, nz, m)                                                                           \
    do {                                                                                                    \
        if ((nx) < (m) || (ny) < (m) || (nz) < (m))                                                         \
            return fail(NS3D_ERR_ARG, "%s: grid %dx%dx%d too small (need >= %d per direction)", __func__,   \
                        (nx), (ny), (nz), (m));                                                             \
    } while (0)

// ---- pieces of ns3d_api.cpp that the multi-GPU layer drives on its ranks' contexts (no host synchronisation) ----------
int ns3d_check_pt_params(const ns3d_pt_params *p, const char *fn);
// two fused PT iterations on stream s, tile shape as the context would choose it (looked up or built-in; never tuned here)
template <class T>
hipError_t ns3d_enqueue_pt2(ns3d_ctx *c, hipStream_t s, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                            const ns3d_pt_params *p, int k0, int k1);
// one pass of `depth` (2…4) PT iterations on stream s with the planned tile shapes (looked up; never measured here) — or with the
// shapes the caller kept from its own plan phase (v2 / vn ≥ 0: ns3d_slab_plan and box_plan measure under a depth they pin for the
// measurement only, and such entries do not answer unpinned look-ups)
template <class T>
hipError_t ns3d_enqueue_pass(ns3d_ctx *c, hipStream_t s, int depth, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV,
                             const ns3d_pt_params *p, int k0, int k1, int v2 = -1, int vn = -1, const ns3d_tile_window *win = nullptr,
                             int skip_faces = 0);
// win: a sub-rectangle of the plane's tiles, or (win->geom) a query of the tile grid this pass would use — nothing is launched then;
// skip_faces: no boundary-cell launch behind the sweep (the caller completes them with ns3d_enqueue_faces_region)
template <class T>
hipError_t ns3d_enqueue_faces_region(ns3d_ctx *c, hipStream_t s, T *Pout, const ns3d_pt_params *p, const int c0[3], const int c1[3],
                                     int want_core);
// the plan phase of ns3d_plan_pt on the context's stream (blocks on its own events); returns the planned depth
template <class T>
int ns3d_plan_pt_internal(ns3d_ctx *c, const T *src, T *dst, const T *dsrc, T *ddst, const T *divV, const ns3d_pt_params *p,
                          int k0, int k1);
// one fused PT iteration on stream s
template <class T>
hipError_t ns3d_enqueue_pt1(ns3d_ctx *c, hipStream_t s, const T *src, T *dst, T *d, const T *divV, const ns3d_pt_params *p,
                            int k0, int k1);
// max|∇²Pr − ρ/dt ∇V| as the IEEE bit pattern of a non-negative double (NaN → 0x7FF8…) into key_dev, on stream s
template <class T>
hipError_t ns3d_enqueue_residual_key(ns3d_ctx *c, hipStream_t s, const T *Pr, const T *divV, const ns3d_pt_params *p,
                                     unsigned long long *key_dev);
// halo-stripped copy A[1:sx-1,1:sy-1,1:sz-1] → packed out (gather!, multi.jl:399)
template <class T>
hipError_t ns3d_enqueue_strip_inner(ns3d_ctx *c, hipStream_t s, const T *A, T *out, int sx, int sy, int sz);
// advect! on stream s; flags: bit 0 faithful, bit 1 write-through (ns3d_kernels.hip, advect())
template <class T>
hipError_t ns3d_enqueue_advect(ns3d_ctx *c, hipStream_t s, T *Vx, const T *Vx_o, T *Vy, const T *Vy_o, T *Vz, const T *Vz_o, T *C,
                               const T *C_o, double dt, double dx, double dy, double dz, int nx, int ny, int nz, int flags, int koff,
                               int nzg);       // koff / nzg > 0: the arrays are a window koff planes into a global grid of nzg cell planes
// x (dim 0) or y (dim 1) face `idx` of A ↔ packed buffer (update_halo! of a 3-D topology)
template <class T>
hipError_t ns3d_enqueue_face_copy(ns3d_ctx *c, hipStream_t s, T *A, T *buf, int sx, int sy, int sz, int dim, int idx, int unpack);
// up to NS3D_SUBBOX_MAX blocks between column-major arrays of different pitches in one launch (ns3d_launch.h: ns3d_subbox_batch)
template <class T>
hipError_t ns3d_enqueue_subbox_copy(ns3d_ctx *c, hipStream_t s, const ns3d_subbox_batch<T> &batch);
