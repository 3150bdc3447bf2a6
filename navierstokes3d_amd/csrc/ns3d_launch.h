// ns3d_launch.h — host-side launcher declarations shared by ns3d_kernels.hip (compiled twice: once per
// arithmetic mode) and ns3d_api.cpp.  Internal header; the public boundary is include/ns3d.h.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ns3d.h"

// k_pt_persist's exchange area (face values handed between workgroups) and error word: owned by a context, freed with it
// A sweep over a sub-rectangle of a plane's tiles (ns3d_mgpu.cpp box_pass: the shells next to decomposed faces first, the core while
// the exchange runs).  Tiles [x0,x1) × [y0,y1) of the tile grid the launch would otherwise cover; x1 <= x0 means the whole grid.
// geom != nullptr: nothing is launched — the launcher reports the tile grid this depth / variant / grid would use.
// pass_flags of pt_sweep2 / pt_sweepn: bit 1 — no boundary-cell launch behind the sweep; bits 8… — compute units (in eights) the stream's
// CU mask leaves out (ns3d_reserve_cus), which the z-chunking must not count on
#define NS3D_PASS_SKIP_FACES 2
#define NS3D_FACES_FOLDED 2             /* SweepArgs::no_faces: the sweep kernel forms the boundary cells itself (fold_faces) */
#define NS3D_FOLD_MAX_CELLS 20000000L   /* … by default on grids up to this many cells */
struct ns3d_tile_geom { int TX, TY, OV, ntx, nty; };      // columns × rows per tile, overlap 2(NL−1), tiles per plane
struct ns3d_tile_window { int x0, x1, y0, y1; ns3d_tile_geom *geom; };
struct ns3d_persist_state {
    void *H = nullptr;
    unsigned *err = nullptr;            // device word: ticket of the latest launch in which a bounded wait expired
    unsigned *err_host = nullptr;       // the same, in pinned host memory the device writes straight into: the host reads it after any
    unsigned *err_host_dev = nullptr;   //   synchronisation of the stream, without a copy (device-side address of err_host)
    size_t bytes = 0;
    unsigned long long launches = 0;    // key epoch: a launch never accepts a value an earlier one left in a slot
    unsigned ticket = 0;                // number of the latest launch (what a failing launch leaves in the error words)
    unsigned checked = 0;               // launches up to this ticket have been checked by the host
    unsigned faults = 0;                // launches found failed (and redone by the launch-per-iteration path) so far
    hipEvent_t ev = nullptr;            // recorded behind every launch: a launch on ANOTHER stream waits for it (two grids must
                                        // not meet in the same slots) — no stream handle is kept
    // the whole-solve form (residual checks inside the launch): a value/key pair per workgroup on the device, and what the host reads
    // after synchronising — [0] iterations done, [1] checks made, [2+q] the key of max|Rp| at check q — in pinned host memory
    unsigned long long *red = nullptr;
    unsigned long long *res_host = nullptr, *res_host_dev = nullptr;
};
#define NS3D_PERSIST_MAXCHK 2048
#define NS3D_PERSIST_MAXWG 1024        /* < MAXCHK: the workgroups' pairs, the result pair and the parameters share one block */

// k_subbox_copy: one cx·cy·cz block between two column-major arrays (pitches in elements), and a batch of them for one launch
#define NS3D_SUBBOX_MAX 8
template <class T>
struct ns3d_subbox {
    T *dst;
    const T *src;
    long dpx, dpl, spx, spl;
    int cx, cy, cz;
    unsigned blocks;               // filled by the launcher
};
template <class T>
struct ns3d_subbox_batch {
    ns3d_subbox<T> p[NS3D_SUBBOX_MAX];
    int n = 0;
    bool add(T *dst, long dpx, long dpl, const T *src, long spx, long spl, int cx, int cy, int cz)
    {
        if (n >= NS3D_SUBBOX_MAX) return false;
        p[n++] = {dst, src, dpx, dpl, spx, spl, cx, cy, cz, 0u};
        return true;
    }
};

#define NS3D_LAUNCHER_DECLS(NS)                                                                              \
    namespace NS {                                                                                           \
    template <class T>                                                                                       \
    hipError_t update_tau(hipStream_t, T *, T *, T *, T *, T *, T *, const T *, const T *, const T *, double,\
                          double, double, double, int, int, int);                                            \
    template <class T>                                                                                       \
    hipError_t predict_V(hipStream_t, T *, T *, T *, const T *, const T *, const T *, const T *, const T *,  \
                         const T *, double, double, double, double, double, double, int, int, int);          \
    template <class T>                                                                                       \
    hipError_t set_cylinder(hipStream_t, T *, T *, T *, T *, double, double, double, double, double, double, \
                            int local_form, double, double, double, double, double, double, int, int, int);  \
    template <class T>                                                                                       \
    hipError_t update_divV(hipStream_t, T *, const T *, const T *, const T *, double, double, double, int,   \
                           int, int);                                                                        \
    template <class T>                                                                                       \
    hipError_t update_dPrdtau(hipStream_t, const T *, T *, const T *, double, double, double, double, double,\
                              double, double, int, int, int);                                                \
    template <class T>                                                                                       \
    hipError_t update_Pr(hipStream_t, T *, const T *, double, int, int, int);                                \
    template <class T>                                                                                       \
    hipError_t compute_res(hipStream_t, T *, const T *, const T *, double, double, double, double, double,   \
                           int, int, int);                                                                   \
    template <class T>                                                                                       \
    hipError_t max_abs_key(hipStream_t, const T *, long, unsigned long long *key_dev);                       \
    template <class T>                                                                                       \
    hipError_t correct_V(hipStream_t, T *, T *, T *, const T *, double, double, double, double, double, int, \
                         int, int);                                                                          \
    /* which: 0 bc_x 1 bc_y 2 bc_z 3 bc_zV 4 bc_xhydstatic 5 bc_x_Vx 6 bc_x_Pr */                            \
    template <class T>                                                                                       \
    hipError_t bc_plane(hipStream_t, int which, T *, int, int, int, double a, double b, double c, int nz_arg);\
    /* set_bc_Vel! (what 0: A0..A2 = Vx, Vy, Vz) or set_bc_Pr! (what 1: A0 = Pr) as one gather launch; hipErrorInvalidValue where an \
     * extent is too small for that form (the caller then launches rule by rule) */                           \
    template <class T>                                                                                       \
    hipError_t bc_fused(hipStream_t, int what, int bc_kind, T *A0, T *A1, T *A2, int nx, int ny, int nz, int owns, double val, \
                        double rho_g, double dz, int nz_arg);                                                \
    template <class T>                                                                                       \
    hipError_t advect(hipStream_t, T *, const T *, T *, const T *, T *, const T *, T *, const T *, double,   \
                      double, double, double, int, int, int, int, int koff, int nzg);                        \
    template <class T>                                                                                       \
    hipError_t pt_sweep(hipStream_t, int variant, const T *, T *, T *, const T *, const ns3d_pt_params &,    \
                        int k0, int k1);                                                                     \
    template <class T>                                                                                       \
    hipError_t predict_fused(hipStream_t, T *, T *, T *, const T *, const T *, const T *, double mu, double rho, \
                             double g, double dt, double dx, double dy, double dz, int, int, int);           \
    template <class T>                                                                                       \
    hipError_t pt_persist(hipStream_t, const T *, T *, const T *, T *, const T *, const ns3d_pt_params &,    \
                          int n_iters, ns3d_persist_state *, int nchk = 0, double eps = -1.0, double err_mul = 1.0,   \
                          double err_div = 1.0);                                                              \
    template <class T>                                                                                       \
    hipError_t pt_sweep2(hipStream_t, int variant, const T *, T *, const T *, T *, const T *,                \
                         const ns3d_pt_params &, int k0, int k1, int pass_flags, const ns3d_tile_window *win = nullptr); \
    template <class T>                                                                                       \
    hipError_t pt_sweepn(hipStream_t, int nlev, int variant, const T *, T *, const T *, T *, const T *,      \
                         const ns3d_pt_params &, int k0, int k1, int pass_flags, const ns3d_tile_window *win = nullptr); \
    /* the boundary cells of Pout whose SOURCE cell (the interior cell they clamp onto) lies inside (want_core) or outside the  \
     * box [c0,c1) of cells: the partial boundary-cell launches of a split pass (box_pass) */                 \
    template <class T>                                                                                       \
    hipError_t pt_faces_region(hipStream_t, T *Pout, const ns3d_pt_params &, const int c0[3], const int c1[3], int want_core); \
    template <class T>                                                                                       \
    hipError_t residual_max_key(hipStream_t, const T *, const T *, const ns3d_pt_params &,                   \
                                unsigned long long *key_dev);                                                \
    template <class T>                                                                                       \
    hipError_t divtest(hipStream_t, double d, long n, unsigned long long seed, unsigned long long *bad_dev); \
    template <class T>                                                                                       \
    hipError_t strip_inner(hipStream_t, const T *A, T *out, int sx, int sy, int sz);                         \
    template <class T>                                                                                       \
    hipError_t face_copy(hipStream_t, T *A, T *buf, int sx, int sy, int sz, int dim, int idx, int unpack);   \
    template <class T>                                                                                       \
    hipError_t subbox_copy(hipStream_t, const ns3d_subbox_batch<T> &);                                       \
    }

NS3D_LAUNCHER_DECLS(ns3d_strict)
NS3D_LAUNCHER_DECLS(ns3d_strictx)
NS3D_LAUNCHER_DECLS(ns3d_strictp)
NS3D_LAUNCHER_DECLS(ns3d_fast)
