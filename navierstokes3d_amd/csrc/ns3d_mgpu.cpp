// ns3d_mgpu.cpp — the multi-GPU half of the boundary (include/ns3d.h, "multi-GPU" section): what the reference
// gets from ImplicitGlobalGrid.jl + MPI.jl — init_global_grid (multi.jl:325), update_halo! (:371,373,450,453,455,460,462,
// 182,167,477), max_g (:21), gather! (:399-403,528-532), finalize_global_grid (:534) — plus the pseudo-transient loop of
// a z-slab rank (multi.jl:458-471) with its halo traffic hidden behind the interior sweep.
//
// Decomposition: 1-D slabs along z by default, ImplicitGlobalGrid's indexing (overlap 2, halo width 1, nz_g = P·(nz−2)+2, an
// array with nz+s planes has overlap 2+s, physical ends untouched).  Arrays are packed column-major, so every xy-plane — every
// z halo message — is ONE contiguous block: no pack/unpack kernels on the z-slab path.  Any Cartesian topology
// (ns3d_mgpu_create_cart; ns3d_dims_create = MPI_Dims_create, what init_global_grid picks when the script passes no dims) is
// served by update_halo! / gather! / max_g with the reference's kernel-by-kernel loop on top: x and y faces are strided and
// go through k_face_copy on both ends, dimensions in the order x, y, z.  The fused slab path below is z-slab only.
//
// Two forms, one schedule:
//   * ns3d_mgpu_create      — ONE process drives P devices (a device may repeat: P virtual ranks on one GPU, which is how the
//                             one-GPU test box runs the whole layer).  Planes move by hipMemcpyPeerAsync over xGMI, receiver
//                             pulls, ordered by events between the ranks' streams.
//   * ns3d_mgpu_create_rank — one process per GPU (the reference's model, one MPI rank per GPU): an RCCL communicator over
//                             the P ranks; planes move by ncclSend/ncclRecv in one group per exchange on a dedicated
//                             high-priority stream; the residual is reduced by ncclAllReduce(max) on the unsigned bit
//                             pattern of the non-negative double (NaN sorts on top: NaN-propagating like Julia's maximum).
//                             RCCL is resolved with dlopen at the first use, so libns3d.so has no link-time dependency on it
//                             and binds to the librccl already in the process (PyTorch's) when there is one.
// xGMI is point-to-point: a slab chain uses 2 of a GPU's 7 links and moves a few MB per exchange, so the cost is latency
// and ordering, not bandwidth — hence one group / one burst of peer copies per exchange and no per-plane handshakes.
//
// Pseudo-transient loop of a slab rank (ns3d_slab_*, ns3d_pt_solve_slab): the single-GPU fast path advances `depth` PT
// iterations per pass over memory.  Level 2 of a rank's first own plane needs level 1 of the seam halo plane, which needs
// the previous iterate one plane further out: the solve state therefore lives in library-owned buffers EXTENDED by
// G = depth−1 ghost planes per seam (depth = the most iterations a pass may advance, 4 by default).  Every pass recomputes the lower levels on the ghost planes (bit-identical on both
// ranks: same inputs, same arithmetic), afterwards the depth outermost own planes of Pr and the G outermost own planes of
// dPrdτ travel to the neighbour (depth=2: 3 planes per two iterations instead of the reference's ≥2 exchanges per single
// iteration, multi.jl:460-463,182).  The seam-adjacent output planes are swept first, their exchange is posted on the
// communication stream, the interior sweep runs behind it.  Jacobi sweeps are decomposition independent:
// the iterates are bit-identical to the single-device solve of the global grid.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

#include <rccl/rccl.h>   // types and prototypes only; every symbol is resolved by dlopen below

#include "ns3d_internal.h"

namespace {

// ---- RCCL through dlopen ---------------------------------------------------------------------------------------
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string path;
};
RcclApi g_rccl;

int load_rccl()
{
    if (g_rccl.handle) return NS3D_OK;
    const char *cands[] = {std::getenv("NS3D_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    std::string tried;
    for (const char *c : cands) {
        if (!c || !*c) continue;
        h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (h) { g_rccl.path = c; break; }
        tried += std::string(tried.empty() ? "" : ", ") + c;
    }
    if (!h) return fail(NS3D_ERR_RCCL, "RCCL not found (tried %s): %s", tried.c_str(), dlerror());
#define SYM(name)                                                                                           \
    if (!(g_rccl.name = (decltype(g_rccl.name))dlsym(h, "nccl" #name))) {                                   \
        dlclose(h);                                                                                         \
        return fail(NS3D_ERR_RCCL, "%s lacks nccl" #name, g_rccl.path.c_str());                             \
    }
    SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(CommCount) SYM(Send) SYM(Recv) SYM(AllReduce)
    SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
#undef SYM
    g_rccl.handle = h;
    return NS3D_OK;
}

#define NCCLCHK(expr)                                                                                       \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) return fail(NS3D_ERR_RCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_));      \
    } while (0)

static_assert(sizeof(ncclUniqueId) == NS3D_UNIQUE_ID_BYTES, "NS3D_UNIQUE_ID_BYTES must equal sizeof(ncclUniqueId)");

// ---- one z-slab rank driven by this process ----------------------------------------------------------------------
struct SlabState {                // deep-ghost pseudo-transient state (library-owned)
    void *P[2] = {nullptr, nullptr}, *D[2] = {nullptr, nullptr}, *R = nullptr;      // working bases (plane 0 of the CURRENT ghost depth)
    void *Pa[2] = {nullptr, nullptr}, *Da[2] = {nullptr, nullptr}, *Ra = nullptr;   // the allocations (ghost depth at load)
    size_t bytes_P = 0, bytes_D = 0;
    int ip = 0, id = 0;           // current Pr / dPrdτ buffer
    int glo = 0, ghi = 0, nze = 0;
};
struct MRank {
    int rank = 0, device = 0;
    int coords[3] = {0, 0, 0};                     // Cartesian coordinates (MPI_Cart_coords order: last dimension fastest)
    int nbr[3][2] = {{-1, -1}, {-1, -1}, {-1, -1}}; // neighbour ranks per dimension (lower, upper), −1 at a physical end
    void *hbuf = nullptr;                          // update_halo!: packed x / y faces (send lo, send hi, recv lo, recv hi per field)
    size_t hbuf_bytes = 0;
    ns3d_ctx *ctx = nullptr;
    hipStream_t comm = nullptr;                    // halo traffic (high priority)
    hipEvent_t ev_ready = nullptr, ev_landed = nullptr;
    hipEvent_t ev_pass = nullptr;                  // slab_pass: everything the compute stream held when the pass began
    int plan_v2 = -1, plan_vn = -1;                // tile shapes ns3d_slab_plan / box_plan measured for this rank's interior sweeps
    int plan_lo = 0, plan_hi = 0;                  // … and the plane range they were measured on (sub-ranges of it use them too)
    SlabState st;
    void *cbuf = nullptr;                          // solve_cart: the second pressure buffer
    size_t cbuf_bytes = 0;
    void *bbuf = nullptr;                          // solve_box: packed x / y ghost layers (send lo, send hi, recv lo, recv hi per array)
    size_t bbuf_bytes = 0;
    void *gbuf = nullptr;                          // gather!: packed halo-stripped block
    size_t gbuf_bytes = 0;
    void *wbuf = nullptr;                          // advect_wide: the four old and four new fields, one plane wider per seam
    size_t wbuf_bytes = 0;
};
struct Block {                    // one contiguous piece that travels to both neighbours (pointers on the owning rank)
    void *send_lo, *recv_lo, *send_hi, *recv_hi;
    size_t bytes;
};

} // namespace

struct ns3d_mgpu {
    int P = 1, nx = 0, ny = 0, nz = 0, flags = 0;
    int dims[3] = {1, 1, 1};      // process topology; (1,1,P) = z-slabs (the only one the fused ns3d_slab_* path takes)
    std::vector<char> hstage;     // gather! of an x/y-decomposed grid: rank blocks on the host before they are placed
    std::vector<MRank> loc;
    bool rccl = false;
    ncclComm_t comm = nullptr;
    int rccl_ranks = 0;
    int depth = 4;                // most PT iterations a pass may advance = ghost depth + 1 (1: single sweeps, plain one-plane halo)
    bool serial_seams = false;    // A/B switch (NS3D_SLAB_SERIAL_SEAMS=1): the round-2 schedule, seam sweeps ahead of the interior on one stream
    int interior_chunks = 1;      // ns3d_mgpu_set_interior_chunks: launches the interior sweep of a z-slab pass is cut into
    int pass_depth = 2;           // iterations per pass actually used (ns3d_slab_plan may raise it to `depth`; same on every rank)
    // loaded solve
    bool loaded = false;
    int esize = 0, G = 0;
    ns3d_pt_params p;             // the caller's (local grid)
    void *stage = nullptr;        // rank 0, RCCL gather: the global halo-stripped array on the device
    size_t stage_bytes = 0;
};

namespace {

bool has_lower(const ns3d_mgpu *m, const MRank &r) { (void)m; return r.nbr[2][0] >= 0; }     // z neighbours (slab code)
bool has_upper(const ns3d_mgpu *m, const MRank &r) { (void)m; return r.nbr[2][1] >= 0; }
hipStream_t compute(const MRank &r) { return r.ctx->stream; }
bool z_slabs(const ns3d_mgpu *m) { return m->dims[0] == 1 && m->dims[1] == 1; }

void cart_coords(int rank, const int dims[3], int c[3])
{
    c[0] = rank / (dims[1] * dims[2]); c[1] = (rank / dims[2]) % dims[1]; c[2] = rank % dims[2];
}
int cart_rank(const int c[3], const int dims[3]) { return (c[0] * dims[1] + c[1]) * dims[2] + c[2]; }

// Post the exchange of `blocks[l]` (same count and sizes on every rank): everything up to "the ghosts have landed" is
// enqueued on the communication streams, ordered after what the compute streams hold NOW.
// sends_on_comm: what is sent was produced ON the communication streams (slab_pass sweeps the seam planes there), so "ready"
// is a point of those streams instead of the compute streams
int exchange_begin(ns3d_mgpu *m, const std::vector<std::vector<Block>> &blocks, int dim = 2, bool sends_on_comm = false)
{
    const int n = (int)m->loc.size();
    if (m->P == 1) return NS3D_OK;
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipEventRecord(r.ev_ready, sends_on_comm ? r.comm : compute(r)));
    }
    if (m->rccl) {
        MRank &r = m->loc[0];
        ns3d_device_guard g(r.device);
        if (!sends_on_comm) HIPCHK(0, hipStreamWaitEvent(r.comm, r.ev_ready, 0));
        NCCLCHK(g_rccl.GroupStart());
        const int lo = r.nbr[dim][0], hi = r.nbr[dim][1];
        for (const Block &b : blocks[0]) {
            if (lo >= 0) {
                NCCLCHK(g_rccl.Send(b.send_lo, b.bytes, ncclUint8, lo, m->comm, r.comm));
                NCCLCHK(g_rccl.Recv(b.recv_lo, b.bytes, ncclUint8, lo, m->comm, r.comm));
            }
            if (hi >= 0) {
                NCCLCHK(g_rccl.Send(b.send_hi, b.bytes, ncclUint8, hi, m->comm, r.comm));
                NCCLCHK(g_rccl.Recv(b.recv_hi, b.bytes, ncclUint8, hi, m->comm, r.comm));
            }
        }
        NCCLCHK(g_rccl.GroupEnd());
        HIPCHK(0, hipEventRecord(r.ev_landed, r.comm));
        return NS3D_OK;
    }
    // one process, P devices (local index == rank): every receiver pulls its ghost planes from its neighbours once BOTH
    // sides are ready
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        const int lo = r.nbr[dim][0], hi = r.nbr[dim][1];
        ns3d_device_guard g(r.device);
        if (!sends_on_comm) HIPCHK(0, hipStreamWaitEvent(r.comm, r.ev_ready, 0));
        if (lo >= 0) HIPCHK(0, hipStreamWaitEvent(r.comm, m->loc[lo].ev_ready, 0));
        if (hi >= 0) HIPCHK(0, hipStreamWaitEvent(r.comm, m->loc[hi].ev_ready, 0));
        for (size_t q = 0; q < blocks[l].size(); ++q) {
            const Block &b = blocks[l][q];
            if (lo >= 0)
                HIPCHK(0, hipMemcpyPeerAsync(b.recv_lo, r.device, blocks[lo][q].send_hi, m->loc[lo].device, b.bytes, r.comm));
            if (hi >= 0)
                HIPCHK(0, hipMemcpyPeerAsync(b.recv_hi, r.device, blocks[hi][q].send_lo, m->loc[hi].device, b.bytes, r.comm));
        }
        HIPCHK(0, hipEventRecord(r.ev_landed, r.comm));
    }
    return NS3D_OK;
}
// What the compute streams enqueue from now on sees the ghosts — and does not overwrite a plane a neighbour still reads.
int exchange_end(ns3d_mgpu *m, int dim = 2)
{
    const int n = (int)m->loc.size();
    if (m->P == 1) return NS3D_OK;
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipStreamWaitEvent(compute(r), r.ev_landed, 0));
        if (!m->rccl) {
            if (r.nbr[dim][0] >= 0) HIPCHK(0, hipStreamWaitEvent(compute(r), m->loc[r.nbr[dim][0]].ev_landed, 0));
            if (r.nbr[dim][1] >= 0) HIPCHK(0, hipStreamWaitEvent(compute(r), m->loc[r.nbr[dim][1]].ev_landed, 0));
        }
    }
    return NS3D_OK;
}
int sync_all(ns3d_mgpu *m)
{
    for (MRank &r : m->loc) {
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipStreamSynchronize(compute(r)));
        HIPCHK(0, hipStreamSynchronize(r.comm));
    }
    return NS3D_OK;
}
int finish_m(ns3d_mgpu *m)
{
    return (m->flags & NS3D_ASYNC) ? NS3D_OK : sync_all(m);
}

void free_slab(MRank &r)
{
    ns3d_device_guard g(r.device);
    for (int q = 0; q < 2; ++q) {
        if (r.st.Pa[q]) (void)hipFree(r.st.Pa[q]);
        if (r.st.Da[q]) (void)hipFree(r.st.Da[q]);
        r.st.P[q] = r.st.D[q] = r.st.Pa[q] = r.st.Da[q] = nullptr;
    }
    if (r.st.Ra) (void)hipFree(r.st.Ra);
    r.st.R = r.st.Ra = nullptr;
    r.st.bytes_P = r.st.bytes_D = 0;
}

int init_rank(ns3d_mgpu *m, MRank &r, int rank, int device, int flags)
{
    r.rank = rank;
    r.device = device;
    cart_coords(rank, m->dims, r.coords);
    for (int d = 0; d < 3; ++d)
        for (int side = 0; side < 2; ++side) {
            int c[3] = {r.coords[0], r.coords[1], r.coords[2]};
            c[d] += side ? 1 : -1;
            r.nbr[d][side] = (c[d] < 0 || c[d] >= m->dims[d]) ? -1 : cart_rank(c, m->dims);
        }
    r.ctx = ns3d_create(device, flags);
    if (!r.ctx) return NS3D_ERR_HIP;          // message already recorded
    ns3d_device_guard g(device);
    int lo = 0, hi = 0;
    HIPCHK(0, hipDeviceGetStreamPriorityRange(&lo, &hi));       // hi = numerically lowest = highest priority
    HIPCHK(0, hipStreamCreateWithPriority(&r.comm, hipStreamNonBlocking, hi));
    HIPCHK(0, hipEventCreateWithFlags(&r.ev_ready, hipEventDisableTiming));
    HIPCHK(0, hipEventCreateWithFlags(&r.ev_landed, hipEventDisableTiming));
    HIPCHK(0, hipEventCreateWithFlags(&r.ev_pass, hipEventDisableTiming));
    return NS3D_OK;
}

ns3d_mgpu *new_mgpu(const int *dims, int nx, int ny, int nz, int flags, const char *fn)
{
    if (!dims) { fail(NS3D_ERR_ARG, "%s: null dims", fn); return nullptr; }
    if (dims[0] < 1 || dims[1] < 1 || dims[2] < 1 || (long)dims[0] * dims[1] * dims[2] > (1 << 20)) {
        fail(NS3D_ERR_ARG, "%s: dims = (%d,%d,%d)", fn, dims[0], dims[1], dims[2]);
        return nullptr;
    }
    if (nx < 3 || ny < 3 || nz < 3) { fail(NS3D_ERR_ARG, "%s: local grid %dx%dx%d too small (need >= 3)", fn, nx, ny, nz); return nullptr; }
    ns3d_mgpu *m = new ns3d_mgpu();
    m->P = dims[0] * dims[1] * dims[2];
    for (int d = 0; d < 3; ++d) m->dims[d] = dims[d];
    m->nx = nx; m->ny = ny; m->nz = nz; m->flags = flags;
    if (const char *ev = std::getenv("NS3D_SLAB_DEPTH")) m->depth = std::max(1, std::min(4, std::atoi(ev)));
    m->pass_depth = std::min(2, m->depth);
    if (const char *ev = std::getenv("NS3D_SLAB_SERIAL_SEAMS")) m->serial_seams = std::atoi(ev) != 0;
    if (const char *ev = std::getenv("NS3D_SLAB_INTERIOR_CHUNKS")) m->interior_chunks = std::max(1, std::min(16, std::atoi(ev)));
    return m;
}

// ---- deep-ghost slab state --------------------------------------------------------------------------------------
template <class T>
struct Ext {                       // geometry of one rank's extended buffers
    int nx, ny, nz, G, glo, ghi, nze, k0, k1;
    size_t plane, dplane;
    Ext(const ns3d_mgpu *m, const MRank &r)
    {
        nx = m->nx; ny = m->ny; nz = m->nz; G = m->G;
        glo = has_lower(m, r) ? G : 0;
        ghi = has_upper(m, r) ? G : 0;
        nze = nz + glo + ghi;
        k0 = 1 + glo; k1 = nz - 1 + glo;
        plane = (size_t)nx * ny; dplane = (size_t)(nx - 2) * (ny - 2);
    }
};

ns3d_pt_params ext_params(const ns3d_mgpu *m, const MRank &r)
{
    ns3d_pt_params pe = m->p;
    const int glo = has_lower(m, r) ? m->G : 0, ghi = has_upper(m, r) ? m->G : 0;
    pe.nz = m->nz + glo + ghi;
    // G = 0: the halo planes are the buffers' outermost planes and must not be treated as faces; G > 0: the sweeps never
    // produce the outermost planes of a seam side at all (output planes start at 1+G)
    pe.z_lo_is_halo = (m->G == 0 && has_lower(m, r)) ? 1 : 0;
    pe.z_hi_is_halo = (m->G == 0 && has_upper(m, r)) ? 1 : 0;
    return pe;
}

template <class T>
std::vector<Block> ghost_blocks(const ns3d_mgpu *m, const MRank &r, int ip, int idd)
{
    const Ext<T> e(m, r);
    const int G = m->G;
    T *Pq = (T *)r.st.P[ip], *Dq = (T *)r.st.D[idd];
    std::vector<Block> v;
    // Pr: the depth outermost own planes → the neighbour's ghost + halo planes
    const int np = G + 1;
    v.push_back({Pq + e.plane * e.k0, Pq, Pq + e.plane * (e.k1 - np), Pq + e.plane * (e.nze - np), e.plane * np * sizeof(T)});
    // dPrdτ (plane k ↔ index k−1): the G outermost own planes → the neighbour's ghost planes
    if (G > 0)
        v.push_back({Dq + e.dplane * (e.k0 - 1), Dq, Dq + e.dplane * (e.k1 - 1 - G), Dq + e.dplane * (e.nze - 2 - G),
                     e.dplane * G * sizeof(T)});
    return v;
}

template <class T>
int slab_exchange(ns3d_mgpu *m, int ip_of_all, int id_of_all, bool wait)
{
    std::vector<std::vector<Block>> blocks;
    for (MRank &r : m->loc) blocks.push_back(ghost_blocks<T>(m, r, ip_of_all, id_of_all));
    int rc = exchange_begin(m, blocks);
    if (rc) return rc;
    return wait ? exchange_end(m) : NS3D_OK;
}

// one pass: `its` (1 … depth) PT iterations on every local rank.  The seam-adjacent output planes are swept on the rank's
// high-priority COMMUNICATION stream, their exchange follows them there in stream order, and the interior sweep runs on the
// compute stream AT THE SAME TIME (the three sweeps read the same input buffers and write disjoint planes): no launch of a pass
// waits for the tail of another one, and the exchange starts as soon as the thin seam sweeps are done.  (Until round 3 the seam
// sweeps ran on the compute stream ahead of the interior: two more kernel boundaries per pass, 1.5× the global solve's time on
// 66-plane slabs.)  Ordering: the communication stream first waits for everything the compute stream held when the pass began
// (the previous pass's interior planes, and through its exchange_end the neighbours' pulls of the planes this pass overwrites).
template <class T>
int slab_pass(ns3d_mgpu *m, int its)
{
    const int ip = m->loc[0].st.ip, idd = m->loc[0].st.id;          // the ranks advance in lockstep
    const int idd_out = its >= 2 ? idd ^ 1 : idd;
    struct Rng { int lo_end, hi_beg; };
    std::vector<Rng> rng(m->loc.size());
    auto sweep = [&](MRank &r, hipStream_t st, int a, int b) -> int {
        if (b <= a) return NS3D_OK;
        const ns3d_pt_params pe = ext_params(m, r);
        ns3d_device_guard g(r.device);
        hipError_t e;
        if (its >= 2) {
            const bool planned = a >= r.plan_lo && b <= r.plan_hi;              // the interior (or a chunk of it): the rank's own plan
            e = ns3d_enqueue_pass<T>(r.ctx, st, its, (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (const T *)r.st.D[idd],
                                     (T *)r.st.D[idd_out], (const T *)r.st.R, &pe, a, b, planned ? r.plan_v2 : -1, planned ? r.plan_vn : -1);
        }
        else
            e = ns3d_enqueue_pt1<T>(r.ctx, st, (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (T *)r.st.D[idd],
                                    (const T *)r.st.R, &pe, a, b);
        return e == hipSuccess ? NS3D_OK : fail(NS3D_ERR_HIP, "slab sweep launch: %s", hipGetErrorString(e));
    };
    int rc;
    const bool split = m->P > 1 && !m->serial_seams;
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        const Ext<T> e(m, r);
        const int np = m->G + 1;                                    // planes the neighbour needs
        rng[l].lo_end = has_lower(m, r) ? std::min(e.k0 + np, e.k1) : e.k0;
        rng[l].hi_beg = has_upper(m, r) ? std::max(e.k1 - np, rng[l].lo_end) : e.k1;
        hipStream_t seam = compute(r);
        if (split) {
            ns3d_device_guard g(r.device);
            HIPCHK(0, hipEventRecord(r.ev_pass, compute(r)));
            HIPCHK(0, hipStreamWaitEvent(r.comm, r.ev_pass, 0));
            seam = r.comm;
        }
        if ((rc = sweep(r, seam, e.k0, rng[l].lo_end))) return rc;
        if ((rc = sweep(r, seam, rng[l].hi_beg, e.k1))) return rc;
    }
    std::vector<std::vector<Block>> blocks;
    for (MRank &r : m->loc) blocks.push_back(ghost_blocks<T>(m, r, ip ^ 1, idd_out));
    if ((rc = exchange_begin(m, blocks, 2, split))) return rc;
    // the interior, in `interior_chunks` launches over consecutive plane ranges while an exchange is pending (ns3d.h: a kernel of the
    // exchange finds CUs at the latest when a chunk ends; every chunk pays its own pipeline fill, so one launch unless asked)
    for (size_t l = 0; l < m->loc.size(); ++l) {
        const int a = rng[l].lo_end, b = rng[l].hi_beg;
        const int nch = (m->P > 1 && b - a >= 8 * m->interior_chunks) ? m->interior_chunks : 1;
        for (int q = 0; q < nch; ++q)
            if ((rc = sweep(m->loc[l], compute(m->loc[l]), a + (int)((long)(b - a) * q / nch), a + (int)((long)(b - a) * (q + 1) / nch)))) return rc;
    }
    if ((rc = exchange_end(m))) return rc;
    for (MRank &r : m->loc) { r.st.ip = ip ^ 1; r.st.id = idd_out; }
    return NS3D_OK;
}

template <class T>
int slab_iterate(ns3d_mgpu *m, int n)
{
    int rc;
    for (int it = 0; it < n;) {
        const int rem = n - it, d = std::min(m->pass_depth, m->G + 1);
        int its = rem >= d ? ((rem == d + 1 && d >= 3) ? d - 1 : d) : rem;      // 4 = 2+2, not 3+1
        if ((rc = slab_pass<T>(m, its))) return rc;
        it += its;
    }
    return NS3D_OK;
}

// max_g of per-rank keys that sit in each rank's ctx->key_dev (device): host double out
int reduce_keys(ns3d_mgpu *m, double *out)
{
    unsigned long long best = 0ull;
    for (MRank &r : m->loc) {
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        if (m->rccl)        // also with one rank: the same RCCL path whatever the world size
            NCCLCHK(g_rccl.AllReduce(r.ctx->key_dev, r.ctx->key_dev, 1, ncclUint64, ncclMax, m->comm, s));
        HIPCHK(0, hipMemcpyAsync(r.ctx->key_host, r.ctx->key_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    }
    for (MRank &r : m->loc) {
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipStreamSynchronize(compute(r)));
        best = std::max(best, *r.ctx->key_host);
    }
    double v;
    std::memcpy(&v, &best, sizeof v);
    *out = v;
    return NS3D_OK;
}

template <class T>
int slab_residual(ns3d_mgpu *m, double *out)
{
    for (MRank &r : m->loc) {
        const ns3d_pt_params pe = ext_params(m, r);
        ns3d_device_guard g(r.device);
        hipError_t e = ns3d_enqueue_residual_key<T>(r.ctx, compute(r), (const T *)r.st.P[r.st.ip], (const T *)r.st.R, &pe,
                                                    r.ctx->key_dev);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "residual launch: %s", hipGetErrorString(e));
    }
    return reduce_keys(m, out);
}

template <class T>
int slab_load(ns3d_mgpu *m, const T *const *Pr, const T *const *D, const T *const *divV, const ns3d_pt_params *p)
{
    if (!z_slabs(m))
        return fail(NS3D_ERR_STATE, "the fused pseudo-transient path takes z-slab topologies only (dims = (1,1,P)); this grid is "
                    "(%d,%d,%d): run the loop multi.jl:458-471 with the kernel entry points and ns3d_update_halo", m->dims[0],
                    m->dims[1], m->dims[2]);
    int rc = ns3d_check_pt_params(p, "ns3d_slab_load");
    if (rc) return rc;
    if (p->nx != m->nx || p->ny != m->ny || p->nz != m->nz)
        return fail(NS3D_ERR_ARG, "ns3d_slab_load: params grid %dx%dx%d differs from the grid of ns3d_mgpu_create %dx%dx%d", p->nx,
                    p->ny, p->nz, m->nx, m->ny, m->nz);
    if (p->bc_kind != NS3D_BC_MULTI && m->P > 1)
        return fail(NS3D_ERR_ARG, "ns3d_slab_load: gpu.jl's boundary set is single-device");
    if (m->nz < 4 && m->P > 1) return fail(NS3D_ERR_ARG, "ns3d_slab_load: z-slab ranks need at least two interior planes");
    m->p = *p;
    // a rank sends its G+1 outermost OWN planes: slabs thinner than that get fewer ghost planes (and shallower passes)
    m->G = (m->P > 1 ? std::min(m->depth, m->nz - 2) : m->depth) - 1;
    m->esize = (int)sizeof(T);
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        if (!Pr[l] || !D[l] || !divV[l]) return fail(NS3D_ERR_ARG, "ns3d_slab_load: null field pointer (local rank %zu)", l);
        const Ext<T> e(m, r);
        const size_t bp = e.plane * e.nze * sizeof(T), bd = e.dplane * (e.nze - 2) * sizeof(T);
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        if (r.st.bytes_P != bp || r.st.bytes_D != bd) {
            HIPCHK(0, hipStreamSynchronize(s));
            HIPCHK(0, hipStreamSynchronize(r.comm));
            free_slab(r);
            for (int q = 0; q < 2; ++q) {
                HIPCHK(0, hipMalloc(&r.st.Pa[q], bp));
                HIPCHK(0, hipMalloc(&r.st.Da[q], bd));
                HIPCHK(0, hipMemsetAsync(r.st.Pa[q], 0, bp, s));
                HIPCHK(0, hipMemsetAsync(r.st.Da[q], 0, bd, s));
            }
            HIPCHK(0, hipMalloc(&r.st.Ra, bp));
            HIPCHK(0, hipMemsetAsync(r.st.Ra, 0, bp, s));
            r.st.bytes_P = bp; r.st.bytes_D = bd;
        }
        for (int q = 0; q < 2; ++q) { r.st.P[q] = r.st.Pa[q]; r.st.D[q] = r.st.Da[q]; }
        r.st.R = r.st.Ra;
        r.st.glo = e.glo; r.st.ghi = e.ghi; r.st.nze = e.nze; r.st.ip = r.st.id = 0;
        HIPCHK(0, hipMemcpyAsync((T *)r.st.P[0] + e.plane * e.glo, Pr[l], e.plane * e.nz * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIPCHK(0, hipMemcpyAsync((T *)r.st.D[0] + e.dplane * e.glo, D[l], e.dplane * (e.nz - 2) * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIPCHK(0, hipMemcpyAsync((T *)r.st.R + e.plane * e.glo, divV[l], e.plane * e.nz * sizeof(T), hipMemcpyDeviceToDevice, s));
        if (m->G == 0) {    // single sweeps write Pr_out's interior planes only: seed the other buffer's outer planes
            HIPCHK(0, hipMemcpyAsync(r.st.P[1], r.st.P[0], e.plane * sizeof(T), hipMemcpyDeviceToDevice, s));
            HIPCHK(0, hipMemcpyAsync((T *)r.st.P[1] + e.plane * (e.nze - 1), (T *)r.st.P[0] + e.plane * (e.nze - 1),
                                     e.plane * sizeof(T), hipMemcpyDeviceToDevice, s));
        }
    }
    for (MRank &r : m->loc) { r.plan_v2 = r.plan_vn = -1; r.plan_lo = r.plan_hi = 0; }      // a new state: planned again by ns3d_slab_plan
    m->loaded = true;
    // deep ghosts of the incoming state — and of the right-hand side: level 1 on a ghost plane needs ∇V there, one or two
    // planes beyond the one-plane halo the caller's update_halo!(∇V) filled (that plane is re-sent too: same value)
    std::vector<std::vector<Block>> blocks;
    for (MRank &r : m->loc) {
        const Ext<T> e(m, r);
        const int np = m->G + 1;
        std::vector<Block> v = ghost_blocks<T>(m, r, 0, 0);
        T *Rq = (T *)r.st.R;
        v.push_back({Rq + e.plane * e.k0, Rq, Rq + e.plane * (e.k1 - np), Rq + e.plane * (e.nze - np), e.plane * np * sizeof(T)});
        blocks.push_back(v);
    }
    int rc2 = exchange_begin(m, blocks);
    return rc2 ? rc2 : exchange_end(m);
}

template <class T>
int slab_store(ns3d_mgpu *m, T *const *Pr, T *const *D)
{
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        if (!Pr[l] || !D[l]) return fail(NS3D_ERR_ARG, "ns3d_slab_store: null field pointer (local rank %zu)", l);
        const Ext<T> e(m, r);
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        HIPCHK(0, hipMemcpyAsync(Pr[l], (T *)r.st.P[r.st.ip] + e.plane * e.glo, e.plane * e.nz * sizeof(T), hipMemcpyDeviceToDevice, s));
        HIPCHK(0, hipMemcpyAsync(D[l], (T *)r.st.D[r.st.id] + e.dplane * e.glo, e.dplane * (e.nz - 2) * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    return NS3D_OK;
}

// the ranks advance in lockstep: the MINIMUM of the depths the processes measured (min through the max reduction)
int agree_min_depth(ns3d_mgpu *m, int &depth)
{
    if (!m->rccl) return NS3D_OK;
    MRank &r = m->loc[0];
    ns3d_device_guard g(r.device);
    hipStream_t s = compute(r);
    r.ctx->key_host[3] = ~(unsigned long long)depth;
    HIPCHK(0, hipMemcpyAsync(r.ctx->key_dev + 3, r.ctx->key_host + 3, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
    NCCLCHK(g_rccl.AllReduce(r.ctx->key_dev + 3, r.ctx->key_dev + 3, 1, ncclUint64, ncclMax, m->comm, s));
    HIPCHK(0, hipMemcpyAsync(r.ctx->key_host + 3, r.ctx->key_dev + 3, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    HIPCHK(0, hipStreamSynchronize(s));
    depth = (int)~r.ctx->key_host[3];
    return NS3D_OK;
}

// Tile shapes and iterations per pass for the interior range of every rank: measured once, with no exchange in flight.  The
// ranks must advance in lockstep, so the pass depth is the MINIMUM of what the ranks measured (ncclAllReduce(min) across
// processes), capped by the ghost depth.
template <class T>
int slab_plan(ns3d_mgpu *m)
{
    int depth = m->G + 1;
    if (depth >= 2) {
        for (MRank &r : m->loc) {
            const Ext<T> e(m, r);
            const int np = m->G + 1;
            const int a = has_lower(m, r) ? std::min(e.k0 + np, e.k1) : e.k0, b = has_upper(m, r) ? std::max(e.k1 - np, a) : e.k1;
            // Between ranks a pass costs its sweeps PLUS an exchange, and the deepest pass the ghosts allow wins on every slab
            // measured (tools/cart_rates.py, virtual ranks: 8 × 130³ 0.46 / 0.38 / 0.28 ms per iteration with two / three / four per
            // pass, 2 × 258³ 0.25 / 0.22 / 0.20; 512² planes: DESIGN §6) although the sweep alone prefers two or three on small
            // planes: with neighbours the tile shapes are measured FOR that depth unless the caller pinned one (ns3d_set_pt_depth).
            const int pinned = r.ctx->pt_depth;
            const int want = (pinned <= 0 && m->P > 1) ? m->G + 1 : pinned;
            if (b - a < 2) {        // nothing to measure on
                depth = std::min(depth, want >= 2 ? want : 2);
                continue;
            }
            const ns3d_pt_params pe = ext_params(m, r);
            ns3d_device_guard g(r.device);
            // outputs go to the buffers the next pass overwrites anyway
            r.ctx->pt_depth = want;
            const int d = ns3d_plan_pt_internal<T>(r.ctx, (const T *)r.st.P[r.st.ip], (T *)r.st.P[r.st.ip ^ 1],
                                                   (const T *)r.st.D[r.st.id], (T *)r.st.D[r.st.id ^ 1], (const T *)r.st.R, &pe, a, b);
            r.ctx->pt_depth = pinned;
            r.plan_v2 = ns3d_last_pt2_variant(r.ctx); r.plan_vn = ns3d_last_ptn_variant(r.ctx); r.plan_lo = a; r.plan_hi = b;
            depth = std::min(depth, std::max(2, d));
        }
    }
    { const int rcd = agree_min_depth(m, depth); if (rcd) return rcd; }
    m->pass_depth = std::max(1, std::min(depth, m->G + 1));
    // The ghost depth was fixed at load time from the DEEPEST pass allowed (m->depth); when the ranks settle for fewer iterations
    // per pass (exact-division arithmetic, thin slabs) the outer ghost planes are dead weight: every pass would sweep G+1 seam
    // planes per side and exchange (G+1) + G planes where pass_depth + (pass_depth−1) suffice.  Dropping them is a change of
    // base: the buffers keep their allocation, plane 0 moves inwards by the difference on the sides that have a neighbour.
    const int Gnew = m->pass_depth - 1;
    if (m->pass_depth >= 2 && Gnew < m->G) {
        for (MRank &r : m->loc) {
            const Ext<T> e(m, r);
            const size_t off = has_lower(m, r) ? (size_t)(m->G - Gnew) : 0;
            for (int q = 0; q < 2; ++q) {
                r.st.P[q] = (T *)r.st.P[q] + e.plane * off;
                r.st.D[q] = (T *)r.st.D[q] + e.dplane * off;
            }
            r.st.R = (T *)r.st.R + e.plane * off;
        }
        m->G = Gnew;
        for (MRank &r : m->loc) {
            const Ext<T> e(m, r);
            r.st.glo = e.glo; r.st.ghi = e.ghi; r.st.nze = e.nze;
        }
    }
    return NS3D_OK;
}

template <class T>
int solve_slab(ns3d_mgpu *m, T *const *Pr, T *const *D, const T *const *divV, const ns3d_pt_params *p, double eps, int niter,
               int nchk, double err_mul, double err_div, int *iters_done, double *err_hist, int max_checks, int *n_checks)
{
    int rc;
    if ((rc = slab_load<T>(m, Pr, D, divV, p))) return rc;
    if ((rc = slab_plan<T>(m))) return rc;
    int checks = 0, iter = 0, done = niter;
    while (iter < niter) {
        const int n = nchk > 0 ? std::min(nchk - iter % nchk, niter - iter) : niter - iter;   // multi.jl:464
        if ((rc = slab_iterate<T>(m, n))) return rc;
        iter += n;
        if (nchk > 0 && iter % nchk == 0) {                                                     // multi.jl:465-469
            double mx;
            if ((rc = slab_residual<T>(m, &mx))) return rc;
            const double err = mx * err_mul / err_div;
            if (err_hist && checks < max_checks) err_hist[checks] = err;
            ++checks;
            if (eps >= 0 && (err < eps || !std::isfinite(err))) { done = iter; break; }
        }
    }
    if ((rc = slab_store<T>(m, Pr, D))) return rc;
    if (iters_done) *iters_done = done;
    if (n_checks) *n_checks = checks;
    return NS3D_OK;
}

// update_halo!(A…) of ImplicitGlobalGrid: dimension by dimension (x, y, z — corner and edge values travel in two / three
// hops), all fields of the call in one exchange per dimension.  z faces are contiguous planes and travel as they lie; x and
// y faces are packed into / unpacked from the rank's message buffer by k_face_copy on the compute stream.
template <class T>
int update_halo_core(ns3d_mgpu *m, T *const *fields, const int *extents, int nfields)
{
    const int n = (int)m->loc.size();
    const int ncell[3] = {m->nx, m->ny, m->nz};
    for (int f = 0; f < nfields; ++f) {
        const int *e = extents + 3 * f;
        if (e[0] < 1 || e[1] < 1 || e[2] < 1) return fail(NS3D_ERR_ARG, "ns3d_update_halo: field %d has extents %dx%dx%d", f, e[0], e[1], e[2]);
        for (int l = 0; l < n; ++l)
            if (!fields[(size_t)f * n + l]) return fail(NS3D_ERR_ARG, "ns3d_update_halo: null pointer (field %d, local rank %d)", f, l);
        for (int d = 0; d < 3; ++d) {
            const int ol = 2 + (e[d] - ncell[d]);           // ImplicitGlobalGrid: overlap of an array with n+s entries
            if (m->dims[d] > 1 && ol >= 2 && e[d] < 2 * ol - 1)
                return fail(NS3D_ERR_ARG, "ns3d_update_halo: field %d too thin in dimension %d for overlap %d", f, d, ol);
        }
    }
    // message buffer for the strided faces: per (dimension, field) four faces [send lo | send hi | recv lo | recv hi]
    size_t need = 0;
    for (int d = 0; d < 2; ++d) {
        if (m->dims[d] == 1) continue;
        for (int f = 0; f < nfields; ++f) {
            const int *e = extents + 3 * f;
            if (2 + (e[d] - ncell[d]) < 2) continue;
            need += 4 * (size_t)(d == 0 ? e[1] : e[0]) * e[2] * sizeof(T);
        }
    }
    for (int l = 0; l < n && need; ++l) {
        MRank &r = m->loc[l];
        if (r.hbuf_bytes >= need) continue;
        ns3d_device_guard g(r.device);
        if (r.hbuf) {
            // a neighbour may still be pulling from the old buffer: drain every stream of this grid first
            int rc = sync_all(m);
            if (rc) return rc;
            HIPCHK(0, hipFree(r.hbuf));
            r.hbuf = nullptr; r.hbuf_bytes = 0;
        }
        HIPCHK(0, hipMalloc(&r.hbuf, need));
        r.hbuf_bytes = need;
    }
    for (int d = 0; d < 3; ++d) {
        if (m->dims[d] == 1) continue;
        std::vector<std::vector<Block>> blocks(n);
        std::vector<int> fld;                                // fields that have a halo in this dimension
        for (int f = 0; f < nfields; ++f)
            if (2 + (extents[3 * f + d] - ncell[d]) >= 2) fld.push_back(f);
        if (fld.empty()) continue;
        size_t off0 = 0;                                     // dimension 1's faces follow dimension 0's in the buffer
        if (d == 1 && m->dims[0] > 1)
            for (int f = 0; f < nfields; ++f)
                if (2 + (extents[3 * f] - ncell[0]) >= 2) off0 += 4 * (size_t)extents[3 * f + 1] * extents[3 * f + 2] * sizeof(T);
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            size_t off = off0;
            for (int f : fld) {
                const int sx = extents[3 * f], sy = extents[3 * f + 1], sz = extents[3 * f + 2];
                const int sd = extents[3 * f + d], ol = 2 + (sd - ncell[d]);
                T *A = fields[(size_t)f * n + l];
                // sends entry ol (1-based) to the lower / size−(ol−1) to the upper neighbour, receives into 1 / size
                if (d == 2) {
                    const size_t plane = (size_t)sx * sy;
                    blocks[l].push_back({A + plane * (ol - 1), A, A + plane * (sd - ol), A + plane * (sd - 1), plane * sizeof(T)});
                    continue;
                }
                const size_t face = (size_t)(d == 0 ? sy : sx) * sz;
                T *b0 = (T *)((char *)r.hbuf + off);
                off += 4 * face * sizeof(T);
                blocks[l].push_back({b0, b0 + 2 * face, b0 + face, b0 + 3 * face, face * sizeof(T)});
                hipError_t e1 = hipSuccess, e2 = hipSuccess;
                if (r.nbr[d][0] >= 0) e1 = ns3d_enqueue_face_copy<T>(r.ctx, compute(r), A, b0, sx, sy, sz, d, ol - 1, 0);
                if (r.nbr[d][1] >= 0) e2 = ns3d_enqueue_face_copy<T>(r.ctx, compute(r), A, b0 + face, sx, sy, sz, d, sd - ol, 0);
                if (e1 != hipSuccess || e2 != hipSuccess)
                    return fail(NS3D_ERR_HIP, "face pack launch: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            }
        }
        int rc = exchange_begin(m, blocks, d);
        if (rc) return rc;
        if ((rc = exchange_end(m, d))) return rc;
        if (d == 2) continue;
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            for (size_t q = 0; q < fld.size(); ++q) {
                const int f = fld[q];
                const int sx = extents[3 * f], sy = extents[3 * f + 1], sz = extents[3 * f + 2], sd = extents[3 * f + d];
                T *A = fields[(size_t)f * n + l];
                const Block &b = blocks[l][q];
                hipError_t e1 = hipSuccess, e2 = hipSuccess;
                if (r.nbr[d][0] >= 0) e1 = ns3d_enqueue_face_copy<T>(r.ctx, compute(r), A, (T *)b.recv_lo, sx, sy, sz, d, 0, 1);
                if (r.nbr[d][1] >= 0) e2 = ns3d_enqueue_face_copy<T>(r.ctx, compute(r), A, (T *)b.recv_hi, sx, sy, sz, d, sd - 1, 1);
                if (e1 != hipSuccess || e2 != hipSuccess)
                    return fail(NS3D_ERR_HIP, "face unpack launch: %s", hipGetErrorString(e1 != hipSuccess ? e1 : e2));
            }
        }
    }
    return NS3D_OK;
}
template <class T>
int update_halo_impl(ns3d_mgpu *m, T *const *fields, const int *extents, int nfields)
{
    const int rc = update_halo_core<T>(m, fields, extents, nfields);
    return rc ? rc : finish_m(m);
}

// The inner loop multi.jl:458-471 on a topology that is decomposed in x or y: per iteration ONE fused sweep per rank
// ({update_dPrdτ!; update_Pr!; set_bc_Pr!} with the boundary rule folded in on EVERY local face) and ONE update_halo!(Pr),
// which overwrites the faces that have a neighbour with the neighbour's values — x, y, z in turn, so the edge and corner
// entries between a halo face and a physical face arrive from the rank that owns them.  The interior of iterate n+1 reads
// only iterate n, whose halos are complete: the iterates equal the reference's kernel-by-kernel sequence with its four halo
// updates per iteration bit for bit (and the single-device solve of the global grid: Jacobi sweeps are decomposition
// independent).  This is the depth-1 form (ns3d_mgpu_set_temporal(m, 1), NS3D_CART_DEEP=0, ranks too thin for ghosts): solve_box
// below advances several iterations per pass on such topologies too.
template <class T>
int solve_cart(ns3d_mgpu *m, T *const *Pr, T *const *D, const T *const *divV, const ns3d_pt_params *p, double eps, int niter,
               int nchk, double err_mul, double err_div, int *iters_done, double *err_hist, int max_checks, int *n_checks)
{
    int rc = ns3d_check_pt_params(p, "ns3d_pt_solve_slab");
    if (rc) return rc;
    if (p->nx != m->nx || p->ny != m->ny || p->nz != m->nz)
        return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: params grid %dx%dx%d differs from the grid of ns3d_mgpu_create %dx%dx%d", p->nx,
                    p->ny, p->nz, m->nx, m->ny, m->nz);
    if (p->bc_kind != NS3D_BC_MULTI) return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: gpu.jl's boundary set is single-device");
    const int n = (int)m->loc.size();
    const size_t bytes = (size_t)m->nx * m->ny * m->nz * sizeof(T);
    std::vector<T *> cur(n), other(n);
    std::vector<ns3d_pt_params> pe(n, *p);
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        if (!Pr[l] || !D[l] || !divV[l]) return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: null field pointer (local rank %d)", l);
        ns3d_device_guard g(r.device);
        if (r.cbuf_bytes < bytes) {
            HIPCHK(0, hipStreamSynchronize(compute(r)));
            if (r.cbuf) HIPCHK(0, hipFree(r.cbuf));
            r.cbuf = nullptr; r.cbuf_bytes = 0;
            HIPCHK(0, hipMalloc(&r.cbuf, bytes));
            r.cbuf_bytes = bytes;
        }
        cur[l] = Pr[l]; other[l] = (T *)r.cbuf;
        pe[l].owns_outlet = (p->owns_outlet && r.nbr[0][1] < 0) ? 1 : 0;      // multi.jl:179: the ranks on the outlet plane
        pe[l].z_lo_is_halo = pe[l].z_hi_is_halo = 0;                          // every face folded; update_halo! overwrites
    }
    const int ext[3] = {m->nx, m->ny, m->nz};
    int checks = 0, iter = 0, done = niter;
    while (iter < niter) {
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            hipError_t e = ns3d_enqueue_pt1<T>(r.ctx, compute(r), cur[l], other[l], D[l], divV[l], &pe[l], 1, m->nz - 1);
            if (e != hipSuccess) return fail(NS3D_ERR_HIP, "sweep launch: %s", hipGetErrorString(e));
        }
        if ((rc = update_halo_core<T>(m, other.data(), ext, 1))) return rc;
        cur.swap(other);
        ++iter;
        if (nchk > 0 && iter % nchk == 0) {                                                     // multi.jl:464-469
            for (int l = 0; l < n; ++l) {
                MRank &r = m->loc[l];
                ns3d_device_guard g(r.device);
                hipError_t e = ns3d_enqueue_residual_key<T>(r.ctx, compute(r), cur[l], divV[l], &pe[l], r.ctx->key_dev);
                if (e != hipSuccess) return fail(NS3D_ERR_HIP, "residual launch: %s", hipGetErrorString(e));
            }
            double mx;
            if ((rc = reduce_keys(m, &mx))) return rc;
            const double err = mx * err_mul / err_div;
            if (err_hist && checks < max_checks) err_hist[checks] = err;
            ++checks;
            if (eps >= 0 && (err < eps || !std::isfinite(err))) { done = iter; break; }
        }
    }
    for (int l = 0; l < n; ++l)
        if (cur[l] != Pr[l]) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            HIPCHK(0, hipMemcpyAsync(Pr[l], cur[l], bytes, hipMemcpyDeviceToDevice, compute(r)));
        }
    if (iters_done) *iters_done = done;
    if (n_checks) *n_checks = checks;
    return NS3D_OK;
}

// =====================================================================================================================
// Deep ghosts on ANY Cartesian topology (round 3, second session): `init_global_grid(nx,ny,nz)` as multi.jl:325 calls it leaves the
// topology to MPI.Dims_create! — (2,1,1), (2,2,1), (2,2,2) for 2, 4, 8 ranks — so a drop-in for the unmodified script never sees
// z-slabs, and solve_cart's one sweep + one update_halo! per iteration was all it got.  Here every rank keeps its solve state in a
// BOX extended by G = depth−1 ghost cells on each side that has a neighbour, in x, y and z; a pass advances `its` ≤ G+1
// iterations on the whole box as if it were a grid of its own (the boundary rule is folded in on every box face: on a physical
// face that is the reference's rule, on a ghost face it writes values that are wrong — and are at least `its` cells away from the
// rank's own cells by the end of the pass, Jacobi sweeps move information one cell per iteration), then the G+1 outermost OWN layers
// of Pr and the G outermost of dPrdτ travel to the neighbours, dimension by dimension with the FULL extents of the other two
// (x, then y — which carries the fresh x ghosts — then z: edges and corners arrive in two / three hops).  x and y layers are
// strided and go through k_subbox_copy on both ends; z layers are contiguous.  Same iterates as the global solve, bit for bit.
// No overlap of exchange and sweep yet: the win is the pass itself (one pass over memory per `its` iterations, one round of
// exchanges per pass instead of per iteration).
// =====================================================================================================================
template <class T>
struct Box {                       // geometry of one rank's extended box
    int n[3], g[3][2], e[3], G;
    size_t px, pl, dpx, dpl, cells, dcells;
    Box(const ns3d_mgpu *m, const MRank &r)
    {
        n[0] = m->nx; n[1] = m->ny; n[2] = m->nz; G = m->G;
        for (int d = 0; d < 3; ++d) {
            g[d][0] = r.nbr[d][0] >= 0 ? G : 0;
            g[d][1] = r.nbr[d][1] >= 0 ? G : 0;
            e[d] = n[d] + g[d][0] + g[d][1];
        }
        px = (size_t)e[0]; pl = px * e[1]; cells = pl * e[2];
        dpx = (size_t)(e[0] - 2); dpl = dpx * (e[1] - 2); dcells = dpl * (e[2] - 2);
    }
};
struct BoxArr { void *base; bool dshape; int layers; };   // an array of the box and how many layers per seam travel

int box_ghost_depth(const ns3d_mgpu *m)
{
    int d = m->depth;
    const int n[3] = {m->nx, m->ny, m->nz};
    for (int q = 0; q < 3; ++q)
        if (m->dims[q] > 1) d = std::min(d, n[q] - 2);      // a rank sends its G+1 outermost OWN layers
    return d - 1;
}
bool box_enabled(const ns3d_mgpu *m, const ns3d_pt_params *p)
{
    const char *ev = std::getenv("NS3D_CART_DEEP");
    if (ev && std::atoi(ev) == 0) return false;
    return p && p->bc_kind == NS3D_BC_MULTI && m->P > 1 && box_ghost_depth(m) >= 1;
}
template <class T>
ns3d_pt_params box_params(const ns3d_mgpu *m, const MRank &r)
{
    const Box<T> b(m, r);
    ns3d_pt_params pe = m->p;
    pe.nx = b.e[0]; pe.ny = b.e[1]; pe.nz = b.e[2];
    pe.owns_outlet = (m->p.owns_outlet && r.nbr[0][1] < 0) ? 1 : 0;       // multi.jl:179: the ranks on the outlet plane
    pe.z_lo_is_halo = pe.z_hi_is_halo = 0;
    return pe;
}

// ghost layers of `arrs[l]` (same list on every rank) from the neighbours' own layers: x, y, z in turn
// on_comm: the whole chain — pack, exchange, unpack, dimension after dimension — runs on the ranks' communication streams (box_pass
// with the shells swept there first, the core sweep on the compute streams meanwhile); the caller joins the streams afterwards
template <class T>
int box_exchange(ns3d_mgpu *m, const std::vector<std::vector<BoxArr>> &arrs, bool on_comm = false)
{
    const int n = (int)m->loc.size();
    auto lane = [&](MRank &r) { return on_comm ? r.comm : compute(r); };
    // message buffer: the largest dimension's faces
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        const Box<T> b(m, r);
        size_t need = 0;
        for (int d = 0; d < 2; ++d) {
            if (m->dims[d] == 1) continue;
            size_t nd = 0;
            for (const BoxArr &a : arrs[l]) {
                const int s0 = b.e[0] - (a.dshape ? 2 : 0), s1 = b.e[1] - (a.dshape ? 2 : 0), s2 = b.e[2] - (a.dshape ? 2 : 0);
                nd += 4 * (size_t)a.layers * (d == 0 ? s1 : s0) * s2 * sizeof(T);
            }
            need = std::max(need, nd);
        }
        if (need > r.bbuf_bytes) {
            ns3d_device_guard g(r.device);
            if (r.bbuf) {
                int rc = sync_all(m);                  // a neighbour may still be pulling from the old buffer
                if (rc) return rc;
                HIPCHK(0, hipFree(r.bbuf));
                r.bbuf = nullptr; r.bbuf_bytes = 0;
            }
            HIPCHK(0, hipMalloc(&r.bbuf, need));
            r.bbuf_bytes = need;
        }
    }
    for (int d = 0; d < 3; ++d) {
        if (m->dims[d] == 1) continue;
        std::vector<std::vector<Block>> blocks(n);
        struct Piece { T *base; size_t px, pl; int c[3]; int recv_lo, recv_hi; };      // for the unpack
        std::vector<std::vector<Piece>> pieces(n);
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            const Box<T> b(m, r);
            ns3d_device_guard g(r.device);
            size_t off = 0;
            ns3d_subbox_batch<T> pack;                  // every array's layers for both neighbours: one launch
            for (const BoxArr &a : arrs[l]) {
                const int sh = a.dshape ? 2 : 0, L = a.layers;
                const int s[3] = {b.e[0] - sh, b.e[1] - sh, b.e[2] - sh};
                const size_t px = (size_t)s[0], pl = px * s[1];
                const size_t pitch[3] = {1, px, pl};
                const int k0 = b.g[d][0] + 1, k1 = b.g[d][0] + b.n[d] - 1;               // own inner layers [k0,k1) of the P-shaped arrays
                const int send_lo = a.dshape ? k0 - 1 : k0, send_hi = (a.dshape ? k1 - 1 : k1) - L;
                const int recv_lo = 0, recv_hi = s[d] - L;
                T *A = (T *)a.base;
                if (d == 2) {
                    blocks[l].push_back({A + pl * send_lo, A + pl * recv_lo, A + pl * send_hi, A + pl * recv_hi, pl * L * sizeof(T)});
                    continue;
                }
                int c[3] = {s[0], s[1], s[2]};
                c[d] = L;
                const size_t face = (size_t)c[0] * c[1] * c[2];
                T *b0 = (T *)((char *)r.bbuf + off);
                off += 4 * face * sizeof(T);
                blocks[l].push_back({b0, b0 + 2 * face, b0 + face, b0 + 3 * face, face * sizeof(T)});
                pieces[l].push_back({A, px, pl, {c[0], c[1], c[2]}, recv_lo, recv_hi});
                bool ok = true;
                if (r.nbr[d][0] >= 0)
                    ok = ok && pack.add(b0, c[0], (long)c[0] * c[1], A + pitch[d] * send_lo, (long)px, (long)pl, c[0], c[1], c[2]);
                if (r.nbr[d][1] >= 0)
                    ok = ok && pack.add(b0 + face, c[0], (long)c[0] * c[1], A + pitch[d] * send_hi, (long)px, (long)pl, c[0], c[1], c[2]);
                if (!ok) return fail(NS3D_ERR_STATE, "box_exchange: more than %d pieces in one pack", NS3D_SUBBOX_MAX);
            }
            if (d < 2) {
                hipError_t e = ns3d_enqueue_subbox_copy<T>(r.ctx, lane(r), pack);
                if (e != hipSuccess) return fail(NS3D_ERR_HIP, "ghost pack launch: %s", hipGetErrorString(e));
            }
        }
        int rc = exchange_begin(m, blocks, d, on_comm);
        if (rc) return rc;
        if (!on_comm) { if ((rc = exchange_end(m, d))) return rc; }
        else if (!m->rccl)      // the next pack overwrites the message buffer: not before the neighbours have pulled this dimension's
            for (int l = 0; l < n; ++l) {
                MRank &r = m->loc[l];
                ns3d_device_guard g(r.device);
                for (int side = 0; side < 2; ++side)
                    if (r.nbr[d][side] >= 0) HIPCHK(0, hipStreamWaitEvent(r.comm, m->loc[r.nbr[d][side]].ev_landed, 0));
            }
        if (d == 2) continue;
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            ns3d_subbox_batch<T> unpack;
            bool ok = true;
            for (size_t q = 0; q < pieces[l].size(); ++q) {
                const Piece &pc = pieces[l][q];
                const Block &bk = blocks[l][q];
                const size_t pitch = d == 0 ? 1 : pc.px;
                if (r.nbr[d][0] >= 0)
                    ok = ok && unpack.add(pc.base + pitch * pc.recv_lo, (long)pc.px, (long)pc.pl, (const T *)bk.recv_lo, pc.c[0],
                                          (long)pc.c[0] * pc.c[1], pc.c[0], pc.c[1], pc.c[2]);
                if (r.nbr[d][1] >= 0)
                    ok = ok && unpack.add(pc.base + pitch * pc.recv_hi, (long)pc.px, (long)pc.pl, (const T *)bk.recv_hi, pc.c[0],
                                          (long)pc.c[0] * pc.c[1], pc.c[0], pc.c[1], pc.c[2]);
            }
            if (!ok) return fail(NS3D_ERR_STATE, "box_exchange: more than %d pieces in one unpack", NS3D_SUBBOX_MAX);
            hipError_t e = ns3d_enqueue_subbox_copy<T>(r.ctx, lane(r), unpack);
            if (e != hipSuccess) return fail(NS3D_ERR_HIP, "ghost unpack launch: %s", hipGetErrorString(e));
        }
    }
    return NS3D_OK;
}

template <class T>
int box_load(ns3d_mgpu *m, const T *const *Pr, const T *const *D, const T *const *divV, const ns3d_pt_params *p)
{
    m->p = *p;
    m->G = box_ghost_depth(m);
    m->esize = (int)sizeof(T);
    std::vector<std::vector<BoxArr>> arrs;
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        if (!Pr[l] || !D[l] || !divV[l]) return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: null field pointer (local rank %zu)", l);
        const Box<T> b(m, r);
        const size_t bp = b.cells * sizeof(T), bd = b.dcells * sizeof(T);
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        if (r.st.bytes_P != bp || r.st.bytes_D != bd) {
            int rc = sync_all(m);
            if (rc) return rc;
            free_slab(r);
            for (int q = 0; q < 2; ++q) {
                HIPCHK(0, hipMalloc(&r.st.Pa[q], bp));
                HIPCHK(0, hipMalloc(&r.st.Da[q], bd));
            }
            HIPCHK(0, hipMalloc(&r.st.Ra, bp));
            r.st.bytes_P = bp; r.st.bytes_D = bd;
        }
        for (int q = 0; q < 2; ++q) {
            r.st.P[q] = r.st.Pa[q]; r.st.D[q] = r.st.Da[q];
            HIPCHK(0, hipMemsetAsync(r.st.P[q], 0, bp, s));
            HIPCHK(0, hipMemsetAsync(r.st.D[q], 0, bd, s));
        }
        r.st.R = r.st.Ra;
        HIPCHK(0, hipMemsetAsync(r.st.R, 0, bp, s));
        r.st.ip = r.st.id = 0;
        const size_t op = b.g[0][0] + b.px * b.g[1][0] + b.pl * b.g[2][0], od = b.g[0][0] + b.dpx * b.g[1][0] + b.dpl * b.g[2][0];
        const long nx = b.n[0], ny = b.n[1];
        ns3d_subbox_batch<T> in;
        in.add((T *)r.st.P[0] + op, (long)b.px, (long)b.pl, Pr[l], nx, nx * ny, b.n[0], b.n[1], b.n[2]);
        in.add((T *)r.st.R + op, (long)b.px, (long)b.pl, divV[l], nx, nx * ny, b.n[0], b.n[1], b.n[2]);
        in.add((T *)r.st.D[0] + od, (long)b.dpx, (long)b.dpl, D[l], nx - 2, (nx - 2) * (ny - 2), b.n[0] - 2, b.n[1] - 2, b.n[2] - 2);
        hipError_t e = ns3d_enqueue_subbox_copy<T>(r.ctx, s, in);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "box load launch: %s", hipGetErrorString(e));
        arrs.push_back({{r.st.P[0], false, m->G + 1}, {r.st.D[0], true, m->G}, {r.st.R, false, m->G + 1}});
    }
    return box_exchange<T>(m, arrs);
}

template <class T>
int box_store(ns3d_mgpu *m, T *const *Pr, T *const *D)
{
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        const Box<T> b(m, r);
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        const size_t op = b.g[0][0] + b.px * b.g[1][0] + b.pl * b.g[2][0], od = b.g[0][0] + b.dpx * b.g[1][0] + b.dpl * b.g[2][0];
        const long nx = b.n[0], ny = b.n[1];
        ns3d_subbox_batch<T> out;
        out.add(Pr[l], nx, nx * ny, (const T *)r.st.P[r.st.ip] + op, (long)b.px, (long)b.pl, b.n[0], b.n[1], b.n[2]);
        out.add(D[l], nx - 2, (nx - 2) * (ny - 2), (const T *)r.st.D[r.st.id] + od, (long)b.dpx, (long)b.dpl, b.n[0] - 2, b.n[1] - 2, b.n[2] - 2);
        hipError_t e = ns3d_enqueue_subbox_copy<T>(r.ctx, s, out);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "box store launch: %s", hipGetErrorString(e));
    }
    return NS3D_OK;
}

// The shells first (round 4; the reference reserves b_width = (8,8,4) "for comm / comp overlap", multi.jl:326, and never
// uses it).  A pass of `its` ≥ 2 iterations is issued in pieces: on the COMMUNICATION stream the tiles / planes that produce every own
// cell within G+1 layers of a decomposed face — two plane ranges in z, two strips of tile rows, two strips of tile columns (sub-rectangles
// of the plane's tile grid: ns3d_tile_window) — then the boundary cells whose source cell lies in those shells, then the whole exchange
// chain x → y → z (packs and unpacks on that stream too); on the COMPUTE stream, at the same time, the core: the remaining tiles of the
// remaining planes.  The streams join, and the boundary cells whose source lies in the core are completed (k_pt_faces_region: nothing an
// unpack has written is touched — a boundary cell on a ghost side has its source in a shell).  Same launches' arithmetic, same bits.
// Taken from ≈40 M cells per rank on (below, and with NS3D_BOX_OVERLAP=0: the whole box, then the exchange — round 3's order).
template <class T>
int box_pass_overlapped(ns3d_mgpu *m, int its, bool &done)
{
    done = false;
    // Default by size: the pieces are seven launches instead of one, and on one GPU (virtual ranks, tools/ab/box_overlap_ab.sh) that costs
    // +44 % at 130³ per rank, +16 % at 258³ on (2,2,2), +0.7 % at 386³, +3 % at 512³ — below ≈40 M cells per rank the pass is bound by
    // launch and event latencies, not by the bytes the exchange moves, and keeps round 3's order.  NS3D_BOX_OVERLAP=1 / 0 forces it.
    const char *ev = std::getenv("NS3D_BOX_OVERLAP");
    const int forced = ev ? std::atoi(ev) : -1;
    if (forced == 0 || its < 2 || m->P == 1) return NS3D_OK;
    if (forced < 0 && (long long)m->nx * m->ny * m->nz < 40ll * 1000 * 1000) return NS3D_OK;
    const int ip = m->loc[0].st.ip, idd = m->loc[0].st.id, idd_out = idd ^ 1;
    const int n = (int)m->loc.size();
    struct Split { int c0[3], c1[3], tx0, tx1, ty0, ty1; ns3d_tile_geom ge; ns3d_pt_params pe; };
    std::vector<Split> sp((size_t)n);
    // geometry first (no launch yet): every rank must be able to split, or none does
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        const Box<T> b(m, r);
        Split &q = sp[(size_t)l];
        q.pe = box_params<T>(m, r);
        ns3d_tile_window w{0, 0, 0, 0, &q.ge};
        ns3d_device_guard g(r.device);
        hipError_t e = ns3d_enqueue_pass<T>(r.ctx, compute(r), its, (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (const T *)r.st.D[idd],
                                            (T *)r.st.D[idd_out], (const T *)r.st.R, &q.pe, 1, q.pe.nz - 1, r.plan_v2, r.plan_vn, &w, 1);
        if (e != hipSuccess) { (void)hipGetLastError(); return NS3D_OK; }         // a shape without the query: the plain order
        const int S[2] = {q.ge.TX - q.ge.OV, q.ge.TY - q.ge.OV}, nt[2] = {q.ge.ntx, q.ge.nty}, half = q.ge.OV / 2;
        int tlo[2], thi[2];
        for (int d = 0; d < 3; ++d) {
            const int k0 = b.g[d][0] + 1, k1 = b.g[d][0] + b.n[d] - 1;            // own inner cells [k0,k1) of the box
            int lo = r.nbr[d][0] >= 0 ? std::min(k0 + b.G + 1, b.e[d] - 1) : 1, hi = r.nbr[d][1] >= 0 ? std::max(k1 - (b.G + 1), lo) : b.e[d] - 1;
            if (d < 2) {     // whole tiles: tile t produces the cells [begin(t), begin(t+1))
                auto begin = [&](int t) { return t <= 0 ? 1 : (t >= nt[d] ? b.e[d] - 1 : 1 + t * S[d] + half); };
                int a = 0;
                while (a < nt[d] && begin(a) < lo) ++a;
                int z = nt[d];
                if (r.nbr[d][1] >= 0) { z = nt[d] - 1; while (z > a && begin(z) > hi) --z; }
                if (z < a) z = a;
                tlo[d] = a; thi[d] = z;
                lo = begin(a); hi = begin(z);
            }
            q.c0[d] = lo; q.c1[d] = std::max(lo, hi);
        }
        q.tx0 = tlo[0]; q.tx1 = thi[0]; q.ty0 = tlo[1]; q.ty1 = thi[1];
    }
    std::vector<std::vector<BoxArr>> arrs;
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        Split &q = sp[(size_t)l];
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipEventRecord(r.ev_pass, compute(r)));
        HIPCHK(0, hipStreamWaitEvent(r.comm, r.ev_pass, 0));
        auto piece = [&](hipStream_t st, int k0, int k1, int x0, int x1, int y0, int y1) -> int {
            if (k1 <= k0 || x1 <= x0 || y1 <= y0) return NS3D_OK;
            ns3d_tile_window w{x0, x1, y0, y1, nullptr};
            // the planned tile SHAPE with the z-chunking left to the launcher: a strip of a few tiles must be cut into enough z-chunks to
            // occupy the chip (a planned "one chunk per tile column" would march 9 tiles through the whole z range on 9 CUs)
            const int v2 = r.plan_v2 >= 0 ? r.plan_v2 / 100 * 100 : -1, vn = r.plan_vn >= 0 ? r.plan_vn / 100 * 100 : -1;
            hipError_t e = ns3d_enqueue_pass<T>(r.ctx, st, its, (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (const T *)r.st.D[idd],
                                                (T *)r.st.D[idd_out], (const T *)r.st.R, &q.pe, k0, k1, v2, vn, &w, 1);
            return e == hipSuccess ? NS3D_OK : fail(NS3D_ERR_HIP, "box sweep launch: %s", hipGetErrorString(e));
        };
        const int nzb = q.pe.nz, ntx = q.ge.ntx, nty = q.ge.nty;
        int rc;
        if ((rc = piece(r.comm, 1, q.c0[2], 0, ntx, 0, nty))) return rc;                       // z shells: whole planes
        if ((rc = piece(r.comm, q.c1[2], nzb - 1, 0, ntx, 0, nty))) return rc;
        if ((rc = piece(r.comm, q.c0[2], q.c1[2], 0, ntx, 0, q.ty0))) return rc;               // y shells: strips of tile rows
        if ((rc = piece(r.comm, q.c0[2], q.c1[2], 0, ntx, q.ty1, nty))) return rc;
        if ((rc = piece(r.comm, q.c0[2], q.c1[2], 0, q.tx0, q.ty0, q.ty1))) return rc;         // x shells: strips of tile columns
        if ((rc = piece(r.comm, q.c0[2], q.c1[2], q.tx1, ntx, q.ty0, q.ty1))) return rc;
        hipError_t e = ns3d_enqueue_faces_region<T>(r.ctx, r.comm, (T *)r.st.P[ip ^ 1], &q.pe, q.c0, q.c1, 0);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "box boundary-cell launch: %s", hipGetErrorString(e));
        if ((rc = piece(compute(r), q.c0[2], q.c1[2], q.tx0, q.tx1, q.ty0, q.ty1))) return rc;  // the core, meanwhile
        arrs.push_back({{r.st.P[ip ^ 1], false, m->G + 1}, {r.st.D[idd_out], true, m->G}});
    }
    int rc = box_exchange<T>(m, arrs, true);
    if (rc) return rc;
    for (int l = 0; l < n; ++l) {       // join: the core's stream sees the ghosts; nobody overwrites what a neighbour still pulls
        MRank &r = m->loc[l];
        ns3d_device_guard g(r.device);
        HIPCHK(0, hipEventRecord(r.ev_landed, r.comm));
        HIPCHK(0, hipStreamWaitEvent(compute(r), r.ev_landed, 0));
    }
    if (!m->rccl)
        for (int l = 0; l < n; ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            for (int d = 0; d < 3; ++d)
                for (int side = 0; side < 2; ++side)
                    if (r.nbr[d][side] >= 0) HIPCHK(0, hipStreamWaitEvent(compute(r), m->loc[r.nbr[d][side]].ev_landed, 0));
        }
    for (int l = 0; l < n; ++l) {
        MRank &r = m->loc[l];
        Split &q = sp[(size_t)l];
        ns3d_device_guard g(r.device);
        hipError_t e = ns3d_enqueue_faces_region<T>(r.ctx, compute(r), (T *)r.st.P[ip ^ 1], &q.pe, q.c0, q.c1, 1);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "box boundary-cell launch: %s", hipGetErrorString(e));
        r.st.ip = ip ^ 1; r.st.id = idd_out;
    }
    done = true;
    return NS3D_OK;
}

// one pass: `its` (1 … G+1) PT iterations on every local rank's box, then the ghosts of the new state
template <class T>
int box_pass(ns3d_mgpu *m, int its)
{
    {
        bool done = false;
        const int rc = box_pass_overlapped<T>(m, its, done);
        if (rc || done) return rc;
    }
    const int ip = m->loc[0].st.ip, idd = m->loc[0].st.id;
    const int idd_out = its >= 2 ? idd ^ 1 : idd;
    std::vector<std::vector<BoxArr>> arrs;
    for (MRank &r : m->loc) {
        const ns3d_pt_params pe = box_params<T>(m, r);
        ns3d_device_guard g(r.device);
        hipError_t e;
        if (its >= 2)
            e = ns3d_enqueue_pass<T>(r.ctx, compute(r), its, (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (const T *)r.st.D[idd],
                                     (T *)r.st.D[idd_out], (const T *)r.st.R, &pe, 1, pe.nz - 1, r.plan_v2, r.plan_vn);
        else
            e = ns3d_enqueue_pt1<T>(r.ctx, compute(r), (const T *)r.st.P[ip], (T *)r.st.P[ip ^ 1], (T *)r.st.D[idd], (const T *)r.st.R,
                                    &pe, 1, pe.nz - 1);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "box sweep launch: %s", hipGetErrorString(e));
        arrs.push_back({{r.st.P[ip ^ 1], false, m->G + 1}, {r.st.D[idd_out], true, m->G}});
    }
    const int rc = box_exchange<T>(m, arrs);
    if (rc) return rc;
    for (MRank &r : m->loc) { r.st.ip = ip ^ 1; r.st.id = idd_out; }
    return NS3D_OK;
}

template <class T>
int box_plan(ns3d_mgpu *m)
{
    int depth = m->G + 1;
    if (depth >= 2)
        for (MRank &r : m->loc) {
            const ns3d_pt_params pe = box_params<T>(m, r);
            ns3d_device_guard g(r.device);
            // A pass costs its sweep plus one round of exchanges (three dimensions, pack and unpack kernels on both ends), so the
            // deepest pass the ghosts allow wins whatever the sweep alone would prefer (tools/cart_rates.py: (2,2,2) ranks of 130³
            // 0.85 / 0.70 / 0.64 ms per iteration with two / three / four per pass): the tile shapes are measured FOR that depth,
            // unless the caller pinned one (ns3d_set_pt_depth).  Outputs go to the buffers the next pass overwrites anyway.
            const int pinned = r.ctx->pt_depth;
            if (pinned <= 0) r.ctx->pt_depth = m->G + 1;
            const int d = ns3d_plan_pt_internal<T>(r.ctx, (const T *)r.st.P[r.st.ip], (T *)r.st.P[r.st.ip ^ 1], (const T *)r.st.D[r.st.id],
                                                   (T *)r.st.D[r.st.id ^ 1], (const T *)r.st.R, &pe, 1, pe.nz - 1);
            r.ctx->pt_depth = pinned;
            r.plan_v2 = ns3d_last_pt2_variant(r.ctx); r.plan_vn = ns3d_last_ptn_variant(r.ctx);
            depth = std::min(depth, std::max(2, d));
        }
    const int rc = agree_min_depth(m, depth);
    if (rc) return rc;
    m->pass_depth = std::max(1, std::min(depth, m->G + 1));
    return NS3D_OK;
}

template <class T>
int solve_box(ns3d_mgpu *m, T *const *Pr, T *const *D, const T *const *divV, const ns3d_pt_params *p, double eps, int niter,
              int nchk, double err_mul, double err_div, int *iters_done, double *err_hist, int max_checks, int *n_checks)
{
    int rc = ns3d_check_pt_params(p, "ns3d_pt_solve_slab");
    if (rc) return rc;
    if (p->nx != m->nx || p->ny != m->ny || p->nz != m->nz)
        return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: params grid %dx%dx%d differs from the grid of ns3d_mgpu_create %dx%dx%d", p->nx,
                    p->ny, p->nz, m->nx, m->ny, m->nz);
    if ((rc = box_load<T>(m, Pr, D, divV, p))) return rc;
    if ((rc = box_plan<T>(m))) return rc;
    int checks = 0, iter = 0, done = niter;
    while (iter < niter) {
        const int n = nchk > 0 ? std::min(nchk - iter % nchk, niter - iter) : niter - iter;   // multi.jl:464
        for (int it = 0; it < n;) {
            const int rem = n - it, d = m->pass_depth;
            const int its = rem >= d ? ((rem == d + 1 && d >= 3) ? d - 1 : d) : rem;            // 4 = 2+2, not 3+1
            if ((rc = box_pass<T>(m, its))) return rc;
            it += its;
        }
        iter += n;
        if (nchk > 0 && iter % nchk == 0) {                                                     // multi.jl:465-469
            for (MRank &r : m->loc) {
                const ns3d_pt_params pe = box_params<T>(m, r);
                ns3d_device_guard g(r.device);
                hipError_t e = ns3d_enqueue_residual_key<T>(r.ctx, compute(r), (const T *)r.st.P[r.st.ip], (const T *)r.st.R, &pe,
                                                            r.ctx->key_dev);
                if (e != hipSuccess) return fail(NS3D_ERR_HIP, "residual launch: %s", hipGetErrorString(e));
            }
            double mx;
            if ((rc = reduce_keys(m, &mx))) return rc;
            const double err = mx * err_mul / err_div;
            if (err_hist && checks < max_checks) err_hist[checks] = err;
            ++checks;
            if (eps >= 0 && (err < eps || !std::isfinite(err))) { done = iter; break; }
        }
    }
    if ((rc = box_store<T>(m, Pr, D))) return rc;
    if (iters_done) *iters_done = done;
    if (n_checks) *n_checks = checks;
    return NS3D_OK;
}

// {X_o .= X; advect!; update_halo!} (multi.jl:475-477) with a z halo of width TWO for the old fields — an option outside the
// reference's multi-rank semantics (SURVEY §7 "hard parts"; VERDICT r2 missing #5): backtrack! clamps its departure indices to
// the LOCAL array (multi.jl:192-195), so with CFL_adv = 1 a departure point that crosses the one-plane halo is clamped on a
// z-slab rank where the one-rank run reads the real neighbour — multi-rank results differ from one-rank results in the cells
// next to a seam.  Here every rank advects on copies of the old fields extended by ONE MORE plane per seam (the neighbour's
// plane sz−ol resp. ol+1, exchanged like halos), so departure points up to two planes away read what the global array holds;
// the new fields' own planes are copied back, then ALL FOUR get their one-plane halo (the reference updates Vx, Vy, Vz only and
// leaves C's halo planes to the clamped local computation).  With it the whole time step is decomposition-independent: P z-slab
// ranks reproduce the one-rank run bit for bit while |δz| < 2 cells (tests/test_gpu_mgpu.py).
template <class T>
int advect_wide(ns3d_mgpu *m, T *const *V[4], T *const *Vo[4], double dt, double dx, double dy, double dz, int faithful)
{
    if (!z_slabs(m)) return fail(NS3D_ERR_STATE, "ns3d_advect_wide: z-slab topologies only (dims = (1,1,P))");
    const int nx = m->nx, ny = m->ny, nz = m->nz;
    const int ext[4][3] = {{nx + 1, ny, nz}, {nx, ny + 1, nz}, {nx, ny, nz + 1}, {nx, ny, nz}};     // Vx, Vy, Vz, C
    if (m->P > 1 && nz < 5) return fail(NS3D_ERR_ARG, "ns3d_advect_wide: slabs of %d planes are too thin for a two-plane halo", nz);
    size_t plane[4], off_o[4], off_n[4], total = 0;
    for (int f = 0; f < 4; ++f) {
        plane[f] = (size_t)ext[f][0] * ext[f][1];
        off_o[f] = total; total += plane[f] * (ext[f][2] + 2);
        off_n[f] = total; total += plane[f] * (ext[f][2] + 2);
    }
    std::vector<std::vector<Block>> blocks(m->loc.size());
    int rc;
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        for (int f = 0; f < 4; ++f)
            if (!V[f][l] || !Vo[f][l]) return fail(NS3D_ERR_ARG, "ns3d_advect_wide: null field pointer (local rank %zu)", l);
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        if (r.wbuf_bytes < total * sizeof(T)) {
            if ((rc = sync_all(m))) return rc;           // a neighbour may still pull from the old buffer
            if (r.wbuf) HIPCHK(0, hipFree(r.wbuf));
            r.wbuf = nullptr; r.wbuf_bytes = 0;
            HIPCHK(0, hipMalloc(&r.wbuf, total * sizeof(T)));
            r.wbuf_bytes = total * sizeof(T);
        }
        const int elo = has_lower(m, r) ? 1 : 0;
        T *W = (T *)r.wbuf;
        for (int f = 0; f < 4; ++f) {
            const int sz = ext[f][2], ol = 2 + (sz - nz);
            T *old = W + off_o[f];
            HIPCHK(0, hipMemcpyAsync(Vo[f][l], V[f][l], plane[f] * sz * sizeof(T), hipMemcpyDeviceToDevice, s));          // X_o .= X
            HIPCHK(0, hipMemcpyAsync(old + plane[f] * elo, V[f][l], plane[f] * sz * sizeof(T), hipMemcpyDeviceToDevice, s));
            // 1-based planes of the local array: ol+1 → the lower neighbour's extra plane above its array, sz−ol → the upper
            // neighbour's extra plane below its array; in the extended copy local plane q sits at index q−1+elo
            blocks[l].push_back({old + plane[f] * (ol + elo), old, old + plane[f] * (sz - ol - 1 + elo), old + plane[f] * (sz + elo),
                                 plane[f] * sizeof(T)});
        }
    }
    if ((rc = exchange_begin(m, blocks))) return rc;
    if ((rc = exchange_end(m))) return rc;
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        const int elo = has_lower(m, r) ? 1 : 0, ehi = has_upper(m, r) ? 1 : 0;
        T *W = (T *)r.wbuf;
        hipError_t e = ns3d_enqueue_advect<T>(r.ctx, s, W + off_n[0], W + off_o[0], W + off_n[1], W + off_o[1], W + off_n[2], W + off_o[2],
                                              W + off_n[3], W + off_o[3], dt, dx, dy, dz, nx, ny, nz + elo + ehi, (faithful ? 1 : 0) | 2,
                                              r.coords[2] * (nz - 2) - elo, m->dims[2] * (nz - 2) + 2);    // departure indices from GLOBAL plane numbers
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "advect launch: %s", hipGetErrorString(e));
        for (int f = 0; f < 4; ++f)
            HIPCHK(0, hipMemcpyAsync(V[f][l], W + off_n[f] + plane[f] * elo, plane[f] * ext[f][2] * sizeof(T), hipMemcpyDeviceToDevice, s));
    }
    // update_halo!(Vx, Vy, Vz) (multi.jl:477) — and C, whose halo planes were computed above from clamped departure points
    std::vector<T *> flat(4 * m->loc.size());
    int extents[12];
    for (int f = 0; f < 4; ++f) {
        for (size_t l = 0; l < m->loc.size(); ++l) flat[(size_t)f * m->loc.size() + l] = V[f][l];
        for (int d = 0; d < 3; ++d) extents[3 * f + d] = ext[f][d];
    }
    return update_halo_core<T>(m, flat.data(), extents, 4);
}

// rank block (bx×by×bz, packed) → its place in the global halo-stripped array (column-major, dims·block entries per side)
template <class T>
void place_block(const T *blk, T *out, const int c[3], int bx, int by, int bz, const int dims[3])
{
    const size_t gx = (size_t)bx * dims[0], gy = (size_t)by * dims[1];
    for (int k = 0; k < bz; ++k)
        for (int j = 0; j < by; ++j)
            std::memcpy(out + ((size_t)(c[2] * bz + k) * gy + (size_t)(c[1] * by + j)) * gx + (size_t)c[0] * bx,
                        blk + ((size_t)k * by + j) * bx, (size_t)bx * sizeof(T));
}

template <class T>
int gather_impl(ns3d_mgpu *m, const T *const *A, int sx, int sy, int sz, T *out_host)
{
    if (sx < 3 || sy < 3 || sz < 3) return fail(NS3D_ERR_ARG, "ns3d_gather: extents %dx%dx%d too small", sx, sy, sz);
    const size_t blk = (size_t)(sx - 2) * (sy - 2) * (sz - 2), bytes = blk * sizeof(T);
    const bool direct = z_slabs(m);                 // z blocks of a column-major array are contiguous: no host placement
    for (size_t l = 0; l < m->loc.size(); ++l) {
        MRank &r = m->loc[l];
        if (!A[l]) return fail(NS3D_ERR_ARG, "ns3d_gather: null pointer (local rank %zu)", l);
        ns3d_device_guard g(r.device);
        if (r.gbuf_bytes < bytes) {
            if (r.gbuf) { HIPCHK(0, hipStreamSynchronize(compute(r))); HIPCHK(0, hipFree(r.gbuf)); r.gbuf = nullptr; r.gbuf_bytes = 0; }
            HIPCHK(0, hipMalloc(&r.gbuf, bytes));
            r.gbuf_bytes = bytes;
        }
        hipError_t e = ns3d_enqueue_strip_inner<T>(r.ctx, compute(r), A[l], (T *)r.gbuf, sx, sy, sz);
        if (e != hipSuccess) return fail(NS3D_ERR_HIP, "strip_inner launch: %s", hipGetErrorString(e));
    }
    const bool root = !m->rccl || m->loc[0].rank == 0;
    if (root && !out_host) return fail(NS3D_ERR_ARG, "ns3d_gather: null output on the process that holds rank 0");
    T *land = out_host;                             // where the rank blocks arrive in rank order
    if (root && !direct) {
        m->hstage.resize(bytes * m->P);
        land = (T *)m->hstage.data();
    }
    if (!m->rccl) {
        for (size_t l = 0; l < m->loc.size(); ++l) {
            MRank &r = m->loc[l];
            ns3d_device_guard g(r.device);
            HIPCHK(0, hipMemcpyAsync(land + blk * r.rank, r.gbuf, bytes, hipMemcpyDeviceToHost, compute(r)));
        }
        int rc = sync_all(m);
        if (rc) return rc;
    } else {
        MRank &r = m->loc[0];
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        if (r.rank != 0) {
            if (m->P > 1) NCCLCHK(g_rccl.Send(r.gbuf, bytes, ncclUint8, 0, m->comm, s));
            HIPCHK(0, hipStreamSynchronize(s));
            return NS3D_OK;
        }
        if (m->stage_bytes < bytes * m->P) {
            if (m->stage) { HIPCHK(0, hipStreamSynchronize(s)); HIPCHK(0, hipFree(m->stage)); m->stage = nullptr; m->stage_bytes = 0; }
            HIPCHK(0, hipMalloc(&m->stage, bytes * m->P));
            m->stage_bytes = bytes * m->P;
        }
        HIPCHK(0, hipMemcpyAsync(m->stage, r.gbuf, bytes, hipMemcpyDeviceToDevice, s));
        if (m->P > 1) {
            NCCLCHK(g_rccl.GroupStart());
            for (int q = 1; q < m->P; ++q) NCCLCHK(g_rccl.Recv((char *)m->stage + bytes * q, bytes, ncclUint8, q, m->comm, s));
            NCCLCHK(g_rccl.GroupEnd());
        }
        HIPCHK(0, hipMemcpyAsync(land, m->stage, bytes * m->P, hipMemcpyDeviceToHost, s));
        HIPCHK(0, hipStreamSynchronize(s));
    }
    if (!direct)
        for (int q = 0; q < m->P; ++q) {
            int c[3];
            cart_coords(q, m->dims, c);
            place_block<T>(land + blk * q, out_host, c, sx - 2, sy - 2, sz - 2, m->dims);
        }
    return NS3D_OK;
}

} // namespace

#define CHECK_M(m)                                                                                          \
    do {                                                                                                    \
        if (!(m)) return fail(NS3D_ERR_ARG, "%s: null ns3d_mgpu", __func__);                                \
    } while (0)

extern "C" {

// MPI_Dims_create as ImplicitGlobalGrid calls it (init_global_grid's dimx=dimy=dimz=0 default, multi.jl:325): entries > 0
// are kept, the zeros are filled with a factorisation of P / (product of the fixed entries) that is as balanced as possible,
// in non-increasing order.
int ns3d_dims_create(int P, int *dims)
{
    if (!dims || P < 1) return fail(NS3D_ERR_ARG, "ns3d_dims_create: P = %d, dims %p", P, (void *)dims);
    long fixed = 1;
    int nfree = 0;
    for (int d = 0; d < 3; ++d) {
        if (dims[d] < 0) return fail(NS3D_ERR_ARG, "ns3d_dims_create: dims[%d] = %d", d, dims[d]);
        if (dims[d] > 0) fixed *= dims[d]; else ++nfree;
    }
    if (P % fixed) return fail(NS3D_ERR_ARG, "ns3d_dims_create: %d ranks are not a multiple of the fixed dimensions (%ld)", P, fixed);
    const int Q = (int)(P / fixed);
    if (nfree == 0) return Q == 1 ? NS3D_OK : fail(NS3D_ERR_ARG, "ns3d_dims_create: fixed dims hold %ld ranks, not %d", fixed, P);
    int best[3] = {Q, 1, 1};
    for (int a = 1; a <= Q; ++a) {                     // a ≥ b ≥ c, a·b·c = Q, the trailing factors 1 when fewer are free
        if (Q % a) continue;
        for (int b = 1; b <= a; ++b) {
            if ((Q / a) % b) continue;
            const int c = Q / a / b;
            if (c > b) continue;
            if ((nfree < 3 && c != 1) || (nfree < 2 && b != 1)) continue;
            if (a < best[0] || (a == best[0] && b < best[1])) { best[0] = a; best[1] = b; best[2] = c; }
        }
    }
    for (int d = 0, q = 0; d < 3; ++d)
        if (dims[d] == 0) dims[d] = best[q++];
    return NS3D_OK;
}

// init_global_grid(nx,ny,nz; dimx,dimy,dimz) (multi.jl:325), one process driving every rank's device (rank order =
// MPI_Cart: the last dimension varies fastest)
ns3d_mgpu *ns3d_mgpu_create_cart(const int *dims, const int *devices, int nx, int ny, int nz, int flags)
{
    ns3d_mgpu *m = new_mgpu(dims, nx, ny, nz, flags, "ns3d_mgpu_create_cart");
    if (!m) return nullptr;
    const int P = m->P;
    if (!devices) { fail(NS3D_ERR_ARG, "ns3d_mgpu_create_cart: null device list"); delete m; return nullptr; }
    m->loc.resize(P);
    for (int l = 0; l < P; ++l)
        if (init_rank(m, m->loc[l], l, devices[l], flags)) { ns3d_mgpu_destroy(m); return nullptr; }
    for (int l = 0; l < P; ++l)            // xGMI peer access between neighbours on different devices
        for (int d = 0; d < 3; ++d) {
            const int q = m->loc[l].nbr[d][1];
            if (q < 0) continue;
            const int a = devices[l], b = devices[q];
            if (a == b) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, a, b) != hipSuccess || !can) {
                (void)hipGetLastError();
                fail(NS3D_ERR_HIP, "ns3d_mgpu_create_cart: devices %d and %d have no peer access", a, b);
                ns3d_mgpu_destroy(m);
                return nullptr;
            }
            for (int dir = 0; dir < 2; ++dir) {
                ns3d_device_guard g(dir ? b : a);
                hipError_t e = hipDeviceEnablePeerAccess(dir ? a : b, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    fail(NS3D_ERR_HIP, "hipDeviceEnablePeerAccess(%d→%d): %s", dir ? b : a, dir ? a : b, hipGetErrorString(e));
                    ns3d_mgpu_destroy(m);
                    return nullptr;
                }
                (void)hipGetLastError();
            }
        }
    return m;
}
// … with dims = (1,1,P): z-slabs
ns3d_mgpu *ns3d_mgpu_create(int P, const int *devices, int nx, int ny, int nz_local, int flags)
{
    if (P < 1) { fail(NS3D_ERR_ARG, "ns3d_mgpu_create: P = %d", P); return nullptr; }
    const int dims[3] = {1, 1, P};
    return ns3d_mgpu_create_cart(dims, devices, nx, ny, nz_local, flags);
}

int ns3d_mgpu_unique_id(void *id_out)
{
    if (!id_out) return fail(NS3D_ERR_ARG, "ns3d_mgpu_unique_id: null output");
    int rc = load_rccl();
    if (rc) return rc;
    ncclUniqueId id;
    NCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, &id, sizeof id);
    return NS3D_OK;
}

// init_global_grid for one rank of prod(dims) processes (one per GPU); collective over the ranks
ns3d_mgpu *ns3d_mgpu_create_rank_cart(const int *dims, int rank, int device, const void *unique_id, int nx, int ny, int nz,
                                      int flags)
{
    ns3d_mgpu *m = new_mgpu(dims, nx, ny, nz, flags, "ns3d_mgpu_create_rank_cart");
    if (!m) return nullptr;
    const int P = m->P;
    if (rank < 0 || rank >= P || !unique_id) {
        fail(NS3D_ERR_ARG, "ns3d_mgpu_create_rank_cart: rank %d of %d, unique_id %p", rank, P, unique_id);
        delete m;
        return nullptr;
    }
    if (load_rccl()) { delete m; return nullptr; }
    m->rccl = true;
    m->loc.resize(1);
    if (init_rank(m, m->loc[0], rank, device, flags)) { ns3d_mgpu_destroy(m); return nullptr; }
    ns3d_device_guard g(device);
    ncclUniqueId id;
    std::memcpy(&id, unique_id, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&m->comm, P, id, rank);
    if (r != ncclSuccess) {
        fail(NS3D_ERR_RCCL, "ncclCommInitRank(rank %d of %d, device %d): %s", rank, P, device, g_rccl.GetErrorString(r));
        m->comm = nullptr;
        ns3d_mgpu_destroy(m);
        return nullptr;
    }
    if (g_rccl.CommCount(m->comm, &m->rccl_ranks) != ncclSuccess) m->rccl_ranks = 0;
    return m;
}
ns3d_mgpu *ns3d_mgpu_create_rank(int P, int rank, int device, const void *unique_id, int nx, int ny, int nz_local, int flags)
{
    if (P < 1) { fail(NS3D_ERR_ARG, "ns3d_mgpu_create_rank: P = %d", P); return nullptr; }
    const int dims[3] = {1, 1, P};
    return ns3d_mgpu_create_rank_cart(dims, rank, device, unique_id, nx, ny, nz_local, flags);
}

// finalize_global_grid() (multi.jl:534)
void ns3d_mgpu_destroy(ns3d_mgpu *m)
{
    if (!m) return;
    for (MRank &r : m->loc) {
        if (!r.ctx) continue;
        ns3d_device_guard g(r.device);
        (void)hipStreamSynchronize(compute(r));
        if (r.comm) (void)hipStreamSynchronize(r.comm);
    }
    if (m->comm) (void)g_rccl.CommDestroy(m->comm);
    for (MRank &r : m->loc) {
        if (!r.ctx) continue;              // a rank that never came up (failed create) owns nothing
        ns3d_device_guard g(r.device);
        free_slab(r);
        if (r.gbuf) (void)hipFree(r.gbuf);
        if (r.hbuf) (void)hipFree(r.hbuf);
        if (r.cbuf) (void)hipFree(r.cbuf);
        if (r.bbuf) (void)hipFree(r.bbuf);
        if (r.wbuf) (void)hipFree(r.wbuf);
        if (r.ev_ready) (void)hipEventDestroy(r.ev_ready);
        if (r.ev_landed) (void)hipEventDestroy(r.ev_landed);
        if (r.ev_pass) (void)hipEventDestroy(r.ev_pass);
        if (r.comm) (void)hipStreamDestroy(r.comm);
        if (r.ctx) ns3d_destroy(r.ctx);
    }
    if (m->stage) (void)hipFree(m->stage);
    delete m;
}

int ns3d_mgpu_world(const ns3d_mgpu *m) { return m ? m->P : -1; }
int ns3d_mgpu_nlocal(const ns3d_mgpu *m) { return m ? (int)m->loc.size() : -1; }
int ns3d_mgpu_rank(const ns3d_mgpu *m, int local) { return (m && local >= 0 && local < (int)m->loc.size()) ? m->loc[local].rank : -1; }
ns3d_ctx *ns3d_mgpu_ctx(ns3d_mgpu *m, int local) { return (m && local >= 0 && local < (int)m->loc.size()) ? m->loc[local].ctx : nullptr; }
int ns3d_mgpu_nz_g(const ns3d_mgpu *m) { return m ? m->dims[2] * (m->nz - 2) + 2 : -1; }
int ns3d_mgpu_dims(const ns3d_mgpu *m, int *dims_out)
{
    CHECK_M(m);
    if (!dims_out) return fail(NS3D_ERR_ARG, "ns3d_mgpu_dims: null output");
    for (int d = 0; d < 3; ++d) dims_out[d] = m->dims[d];
    return NS3D_OK;
}
int ns3d_mgpu_coords(const ns3d_mgpu *m, int local, int *coords_out)
{
    CHECK_M(m);
    if (local < 0 || local >= (int)m->loc.size() || !coords_out) return fail(NS3D_ERR_ARG, "ns3d_mgpu_coords: local rank %d, output %p", local, (void *)coords_out);
    for (int d = 0; d < 3; ++d) coords_out[d] = m->loc[local].coords[d];
    return NS3D_OK;
}
// nx_g(), ny_g(), nz_g() (multi.jl:328-329,338): dims·(n−2)+2
int ns3d_mgpu_n_g(const ns3d_mgpu *m, int *n_g_out)
{
    CHECK_M(m);
    if (!n_g_out) return fail(NS3D_ERR_ARG, "ns3d_mgpu_n_g: null output");
    n_g_out[0] = m->dims[0] * (m->nx - 2) + 2; n_g_out[1] = m->dims[1] * (m->ny - 2) + 2; n_g_out[2] = m->dims[2] * (m->nz - 2) + 2;
    return NS3D_OK;
}
const char *ns3d_mgpu_transport(const ns3d_mgpu *m) { return !m ? "" : (m->rccl ? "rccl" : "peer"); }
int ns3d_mgpu_rccl_ranks(const ns3d_mgpu *m) { return m ? m->rccl_ranks : -1; }
int ns3d_mgpu_pass_depth(const ns3d_mgpu *m) { return m ? m->pass_depth : -1; }
int ns3d_mgpu_ghost_depth(const ns3d_mgpu *m) { return (m && m->loaded) ? m->G : -1; }

int ns3d_mgpu_reserve_cus(ns3d_mgpu *m, int n_cus)
{
    CHECK_M(m);
    for (MRank &r : m->loc) {
        const int rc = ns3d_reserve_cus(r.ctx, n_cus);
        if (rc) return rc;
    }
    return NS3D_OK;
}
int ns3d_mgpu_set_interior_chunks(ns3d_mgpu *m, int chunks)
{
    CHECK_M(m);
    if (chunks < 1 || chunks > 16) return fail(NS3D_ERR_ARG, "ns3d_mgpu_set_interior_chunks: %d (1 … 16)", chunks);
    m->interior_chunks = chunks;
    return NS3D_OK;
}
int ns3d_mgpu_set_temporal(ns3d_mgpu *m, int depth)
{
    CHECK_M(m);
    if (depth < 1 || depth > 4) return fail(NS3D_ERR_ARG, "ns3d_mgpu_set_temporal: depth %d (1 … 4)", depth);
    if (m->loaded && depth != m->depth) m->loaded = false;      // ghost depth changes: the state must be loaded again
    m->depth = depth;
    m->pass_depth = std::min(2, depth);                         // until ns3d_slab_plan measures
    return NS3D_OK;
}

int ns3d_mgpu_sync(ns3d_mgpu *m)
{
    CHECK_M(m);
    return sync_all(m);
}

// max_g(A) = MPI.Allreduce(maximum(A), MPI.MAX) (multi.jl:21) over every rank's local maximum; NaN-propagating
int ns3d_max_g(ns3d_mgpu *m, const double *local_max, double *out)
{
    CHECK_M(m);
    if (!local_max || !out) return fail(NS3D_ERR_ARG, "ns3d_max_g: null argument");
    bool nan = false;
    double best = -INFINITY;
    for (size_t l = 0; l < m->loc.size(); ++l) {
        if (local_max[l] != local_max[l]) nan = true;
        else best = std::max(best, local_max[l]);
    }
    if (m->rccl) {
        // max over doubles through two monotone unsigned keys: [is-NaN flag, order-preserving image of the double]
        MRank &r = m->loc[0];
        ns3d_device_guard g(r.device);
        hipStream_t s = compute(r);
        unsigned long long bits;
        std::memcpy(&bits, &best, sizeof bits);
        bits = (bits & 0x8000000000000000ull) ? ~bits : (bits | 0x8000000000000000ull);
        r.ctx->key_host[1] = nan ? 1ull : 0ull;
        r.ctx->key_host[2] = bits;
        HIPCHK(0, hipMemcpyAsync(r.ctx->key_dev + 1, r.ctx->key_host + 1, 2 * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
        NCCLCHK(g_rccl.AllReduce(r.ctx->key_dev + 1, r.ctx->key_dev + 1, 2, ncclUint64, ncclMax, m->comm, s));
        HIPCHK(0, hipMemcpyAsync(r.ctx->key_host + 1, r.ctx->key_dev + 1, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
        HIPCHK(0, hipStreamSynchronize(s));
        nan = r.ctx->key_host[1] != 0ull;
        bits = r.ctx->key_host[2];
        bits = (bits & 0x8000000000000000ull) ? (bits & 0x7FFFFFFFFFFFFFFFull) : ~bits;
        std::memcpy(&best, &bits, sizeof best);
    }
    *out = nan ? NAN : best;
    return NS3D_OK;
}

// n × { update_dPrdτ!; update_Pr!; set_bc_Pr!; update_halo!(Pr) }  (multi.jl:459-463) on the loaded state
int ns3d_slab_iterate(ns3d_mgpu *m, int n_iters)
{
    CHECK_M(m);
    if (!m->loaded) return fail(NS3D_ERR_STATE, "ns3d_slab_iterate: no state loaded (ns3d_slab_load)");
    if (n_iters < 0) return fail(NS3D_ERR_ARG, "ns3d_slab_iterate: negative count");
    int rc = m->esize == 8 ? slab_iterate<double>(m, n_iters) : slab_iterate<float>(m, n_iters);
    return rc ? rc : finish_m(m);
}
// the tile shape of the interior sweeps, measured now (ns3d_plan_pt on every local rank)
int ns3d_slab_plan(ns3d_mgpu *m)
{
    CHECK_M(m);
    if (!m->loaded) return fail(NS3D_ERR_STATE, "ns3d_slab_plan: no state loaded (ns3d_slab_load)");
    return m->esize == 8 ? slab_plan<double>(m) : slab_plan<float>(m);
}
// max_g(maximum(abs.(Rp))) of the loaded state's current iterate (multi.jl:465-466,21)
int ns3d_slab_residual(ns3d_mgpu *m, double *out)
{
    CHECK_M(m);
    if (!m->loaded) return fail(NS3D_ERR_STATE, "ns3d_slab_residual: no state loaded (ns3d_slab_load)");
    if (!out) return fail(NS3D_ERR_ARG, "ns3d_slab_residual: null output");
    return m->esize == 8 ? slab_residual<double>(m, out) : slab_residual<float>(m, out);
}

} // extern "C"

#define NS3D_MGPU_DEFINE(T, S)                                                                               \
    extern "C" int ns3d_update_halo_##S(ns3d_mgpu *m, T *const *fields, const int *extents, int nfields)     \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (nfields < 0 || (nfields > 0 && (!fields || !extents)))                                           \
            return fail(NS3D_ERR_ARG, "ns3d_update_halo: bad field list");                                   \
        return update_halo_impl<T>(m, fields, extents, nfields);                                             \
    }                                                                                                        \
    extern "C" int ns3d_advect_wide_##S(ns3d_mgpu *m, T *const *Vx, T *const *Vx_o, T *const *Vy, T *const *Vy_o, T *const *Vz, \
                                        T *const *Vz_o, T *const *C, T *const *C_o, double dt, double dx, double dy, double dz, \
                                        int faithful)                                                        \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (!Vx || !Vx_o || !Vy || !Vy_o || !Vz || !Vz_o || !C || !C_o) return fail(NS3D_ERR_ARG, "ns3d_advect_wide: null field list"); \
        T *const *V[4] = {Vx, Vy, Vz, C};                                                                    \
        T *const *Vo[4] = {Vx_o, Vy_o, Vz_o, C_o};                                                           \
        int rc = advect_wide<T>(m, V, Vo, dt, dx, dy, dz, faithful);                                         \
        return rc ? rc : finish_m(m);                                                                        \
    }                                                                                                        \
    extern "C" int ns3d_gather_##S(ns3d_mgpu *m, const T *const *A, int sx, int sy, int sz, T *out_host)     \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (!A) return fail(NS3D_ERR_ARG, "ns3d_gather: null field list");                                   \
        return gather_impl<T>(m, A, sx, sy, sz, out_host);                                                   \
    }                                                                                                        \
    extern "C" int ns3d_slab_load_##S(ns3d_mgpu *m, const T *const *Pr, const T *const *dPrdtau,             \
                                      const T *const *divV, const ns3d_pt_params *p)                         \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (!Pr || !dPrdtau || !divV) return fail(NS3D_ERR_ARG, "ns3d_slab_load: null field list");          \
        m->loaded = false;                                                                                   \
        int rc = slab_load<T>(m, Pr, dPrdtau, divV, p);                                                      \
        return rc ? rc : finish_m(m);                                                                        \
    }                                                                                                        \
    extern "C" int ns3d_slab_store_##S(ns3d_mgpu *m, T *const *Pr, T *const *dPrdtau)                        \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (!m->loaded || m->esize != (int)sizeof(T))                                                        \
            return fail(NS3D_ERR_STATE, "ns3d_slab_store: no state of this element type loaded");            \
        if (!Pr || !dPrdtau) return fail(NS3D_ERR_ARG, "ns3d_slab_store: null field list");                  \
        int rc = slab_store<T>(m, Pr, dPrdtau);                                                              \
        return rc ? rc : finish_m(m);                                                                        \
    }                                                                                                        \
    extern "C" int ns3d_pt_solve_slab_##S(ns3d_mgpu *m, T *const *Pr, T *const *dPrdtau, const T *const *divV,\
                                          const ns3d_pt_params *p, double eps, int niter, int nchk,          \
                                          double err_mul, double err_div, int *iters_done, double *err_hist, \
                                          int max_checks, int *n_checks)                                     \
    {                                                                                                        \
        CHECK_M(m);                                                                                          \
        if (!Pr || !dPrdtau || !divV) return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: null field list");      \
        if (niter < 0 || nchk < 0) return fail(NS3D_ERR_ARG, "ns3d_pt_solve_slab: negative niter/nchk");     \
        m->loaded = false;                                                                                   \
        int rc = z_slabs(m) ? solve_slab<T>(m, Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul, err_div,     \
                                            iters_done, err_hist, max_checks, n_checks)                      \
                 : box_enabled(m, p) ? solve_box<T>(m, Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul,      \
                                                    err_div, iters_done, err_hist, max_checks, n_checks)     \
                            : solve_cart<T>(m, Pr, dPrdtau, divV, p, eps, niter, nchk, err_mul, err_div,     \
                                            iters_done, err_hist, max_checks, n_checks);                     \
        return rc ? rc : finish_m(m);                                                                        \
    }

NS3D_MGPU_DEFINE(double, f64)
NS3D_MGPU_DEFINE(float, f32)
