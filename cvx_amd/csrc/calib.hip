// Camera-calibration bundle adjustment on MI355X: handle, HIP evaluator, C ABI.
// Replaces the body of CameraCalibration::refine between "parameters packed" and
// "parameters unpacked" (/root/reference/modules/calib/src/calibration.cpp:359-442).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <future>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <vector>

#include "calib_kernels.h"
#include "common.h"
#include "lm_driver.h"
#include "solve_kernels.h"
#include "step_kernels.h"
#include "inner_kernels.h"

using namespace cvxk;

namespace {

// CVX_CREATE_PROF=1: where cvx_calib_create / cvx_calib_destroy spend their time, on stderr (development aid)
struct PhaseClock {
    bool on;
    std::chrono::steady_clock::time_point t;
    PhaseClock() : on(std::getenv("CVX_CREATE_PROF") != nullptr), t(std::chrono::steady_clock::now()) {}
    void mark(const char *what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cvx prof] %-34s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};

// One freed arena per device is kept for the next cvx_calib_create (a drop-in refine() is create -> solve -> destroy, and
// freeing / mapping a few hundred MB costs 0.3-0.4 ms of a 6.5 ms call): at most ARENA_KEEP_MAX bytes, handed back to the
// runtime by cvx_release_cached_memory() or when a larger block takes its place; CVX_ARENA_CACHE=0 switches it off.
// Nothing relies on what a block holds (tests run with the arena poisoned, CVX_POISON_ARENA=1).  The caller has
// synchronised the stream of the handle that gives a block back, so no kernel still uses it.
constexpr size_t ARENA_KEEP_MAX = (size_t)1 << 30;
struct ArenaCache {
    std::mutex mu;
    char *p[16] = {};
    size_t cap[16] = {};
} g_arena_cache;
bool arena_cache_on() { return cvx::handle_cache_on(); }
// a kept block of at least `need` bytes (and not more than twice that: a small problem does not sit on a large block)
char *arena_take(int dev, size_t need, size_t *cap) {
    if (!arena_cache_on() || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_arena_cache.mu);
    if (!g_arena_cache.p[dev] || g_arena_cache.cap[dev] < need || g_arena_cache.cap[dev] > 2 * need + ((size_t)8 << 20)) return nullptr;
    char *p = g_arena_cache.p[dev];
    *cap = g_arena_cache.cap[dev];
    g_arena_cache.p[dev] = nullptr; g_arena_cache.cap[dev] = 0;
    return p;
}
// the device of `dev` is current
void arena_give(int dev, char *p, size_t cap) {
    if (!p) return;
    if (arena_cache_on() && dev >= 0 && dev < 16 && cap <= ARENA_KEEP_MAX) {
        std::lock_guard<std::mutex> lock(g_arena_cache.mu);
        if (g_arena_cache.cap[dev] < cap) { std::swap(p, g_arena_cache.p[dev]); std::swap(cap, g_arena_cache.cap[dev]); }
    }
    if (p) (void)hipFree(p);
}

template <class Tp>
int dev_alloc(Tp **p, size_t n) {
    *p = nullptr;
    if (n == 0) n = 1;
    CVX_HIP(hipMalloc((void **)p, n * sizeof(Tp)));
    return CVX_OK;
}

template <class Tp>
int dev_upload(Tp **p, const Tp *src, size_t n, hipStream_t s) {
    int st = dev_alloc(p, n);
    if (st) return st;
    if (n) CVX_HIP(hipMemcpyAsync(*p, src, n * sizeof(Tp), hipMemcpyHostToDevice, s));
    return CVX_OK;
}

__global__ void k_fill(double *p, long long n, double v) {
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void k_widen(const float *__restrict__ xy, double *__restrict__ x, double *__restrict__ y, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { x[i] = (double)xy[i]; y[i] = (double)xy[n + i]; }
}

__global__ void __launch_bounds__(64) k_view_ids(const int *__restrict__ view_off, int *__restrict__ view_id) {
    const int v = blockIdx.x;
    for (int o = view_off[v] + threadIdx.x; o < view_off[v + 1]; o += 64) view_id[o] = v;
}

// cvx_calib_create's checks of the observation lists, on the device (a host pass over 1 M observations costs 0.3 ms of
// a 1.5 ms create).  flags[0]: a point id out of range; flags[1]: a view lists a point twice.  One block per view.
__global__ void __launch_bounds__(64) k_check_views(const int *__restrict__ view_off, const int *__restrict__ point_id, int P,
                                                     int *__restrict__ flags) {
    __shared__ unsigned seen[(CVX_MAX_POINTS + 31) / 32];
    const int v = blockIdx.x;
    if (threadIdx.x < (CVX_MAX_POINTS + 31) / 32) seen[threadIdx.x] = 0u;
    __syncthreads();
    for (int o = view_off[v] + threadIdx.x; o < view_off[v + 1]; o += 64) {
        const int pid = point_id[o];
        if (pid < 0 || pid >= P) { flags[0] = 1; continue; }
        const unsigned bit = 1u << (pid & 31);
        if (atomicOr(&seen[pid >> 5], bit) & bit) flags[1] = 1;
    }
}
// flags[2]: some group's views do not all list the same points in the same order.  One block per group of views.
__global__ void __launch_bounds__(128) k_check_aligned(const int *__restrict__ view_off, const int *__restrict__ point_id,
                                                        const int *__restrict__ grp_v0, int *__restrict__ flags) {
    const int g = blockIdx.x, a0 = grp_v0[g], a1 = grp_v0[g + 1];
    const int o0 = view_off[a0], n0 = view_off[a0 + 1] - o0;
    for (int v = a0 + 1; v < a1; ++v) {
        const int o = view_off[v];
        if (view_off[v + 1] - o != n0) { if (threadIdx.x == 0) flags[2] = 1; return; }
        for (int i = threadIdx.x; i < n0; i += 128)
            if (point_id[o + i] != point_id[o0 + i]) flags[2] = 1;
    }
}

// Jacobi scales of the shared unknowns, s_j^2 = 1 / (1 + sqrt(d_j))^2 (or 1), from the diagonal the assembly left in E:
// made on the device so that a solve's set-up uploads nothing
__global__ void k_shared_scales(const double *__restrict__ dU, double *__restrict__ s2, int n, int jacobi) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = 1.0;
    if (jacobi) { const double s = 1.0 / (1.0 + sqrt(dU[i])); v = s * s; }
    s2[i] = v;
}

enum { PH_NORMAL = 0, PH_SCHUR, PH_ASSEMBLE, PH_BACKSUBST, PH_COST, PH_SOLVE, PH_COUNT };

}  // namespace

struct cvx_calib {
    // problem
    int C = 0, V = 0, P = 0;
    int64_t N = 0;
    int fix_k3 = 1, fix_intr = 0, fix_pts = 0, device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int ncf = 0, cam_free = 0, pts_free = 0, ns = 0, ZC = 0, camcol0 = 0, ycol = 0, nb = 0, nblk = 0, parts = 1, ZCp = 0, KB = 16;
    size_t k3_lds = 0;
    int T = 64;
    bool aligned = false;   // every group's views list the same points in the same order
    size_t k2_lds = 0;
    int n_groups = 0, n_chunks = 0, n_cost_blocks = 0;
    std::vector<int> h_cam_of_view, h_view_off, h_point_id;
    // device: static
    int *d_view_off = nullptr, *d_cam_of_view = nullptr, *d_point_id = nullptr, *d_view_id = nullptr;
    bool A_ready = false;            // the last linearisation wrote the damped matrix A itself (one rank)
    K4Task *d_k4tasks = nullptr;     // task list of k_schur_syrk4 (null: the 16x16x4 kernel, ZC > K4_PITCH)
    int *d_grp_v0 = nullptr, *d_chk_row0 = nullptr, *d_clist = nullptr, *d_clist_off = nullptr, *d_glist = nullptr, *d_glist_off = nullptr;
    double *d_obs_x = nullptr, *d_obs_y = nullptr;
    // device: parameters (current / candidate)
    double *d_intr[2] = {nullptr, nullptr}, *d_points[2] = {nullptr, nullptr}, *d_poses[2] = {nullptr, nullptr};
    int cur = 0;
    // the two buffers of each kind are one allocation, buffer 1 `*_step` doubles behind buffer 0 (multiples of 32), so
    // that a speculated step's kernels can pick the current one from a device-side index (LmDev::cur)
    long long intr_step = 0, points_step = 0, poses_step = 0, rt_step = 0;
    LmDev *d_lm = nullptr, *h_lm = nullptr;      // the speculated steps' record and its pinned staging copy
    long long n_spec_used = 0, n_spec_dropped = 0;   // speculated steps the loop went on to use / that were switched off
    // device: work
    double *d_rt[2] = {nullptr, nullptr};   // per-view R, dR/dw, t of the current / candidate poses
    bool rt_valid[2] = {false, false};
    double *d_Z = nullptr, *d_Linv = nullptr, *d_Dv = nullptr, *d_gv = nullptr, *d_s2pose = nullptr;
    double *d_slab_pt = nullptr, *d_slab_cam = nullptr, *d_grp_scal = nullptr, *d_slab_tile = nullptr;
    double *d_Tpt = nullptr, *d_Tt = nullptr, *d_Gcam = nullptr;
    double *d_E = nullptr, *h_E = nullptr;
    double *d_E_res = nullptr, *h_E_res = nullptr;   // E's places in the arena / the pinned block, good for up to E_res_count doubles
    size_t E_res_count = 0;
    double *h_params = nullptr;      // pinned: 9 C + 3 P + 6 V doubles, set_params / get_params stage through it
    double *d_X = nullptr;           // ranks > 1: the exchanged buffer (packed lower triangle of S + the tail of E); h_E stages it
    size_t X_count = 0, X_tail = 0;
    size_t E_count = 0, E_scal = 0;
    double *d_ds_emb = nullptr, *h_ds_emb = nullptr, *d_dposes = nullptr, *d_out_view = nullptr;
    double *d_cost_partial = nullptr, *d_E2 = nullptr, *h_E2 = nullptr;
    hipEvent_t ev_step = nullptr;
    hipEvent_t ev_params = nullptr;  // behind set_params' copies out of h_params: the next writer of the staging block waits for it
    bool params_pending = false;
    double *d_E2_map = nullptr;      // device view of h_E2 (host-mapped, fine-grained)
    long long step_seq = 0;
    double *d_eval = nullptr;
    double *d_best = nullptr;        // intr | points | poses of the lowest-cost point (non-monotonic runs)
    // device: damped reduced solve (solve_kernels.h)
    int npad = 32, nT = 1;
    double *d_A = nullptr, *d_s2sh = nullptr, *d_Ds = nullptr, *d_delta_s = nullptr;
    int *d_fail = nullptr;
    int *d_flags = nullptr;          // cvx_calib_create's device-side checks: id out of range | point twice in a view | groups not aligned
    size_t fin_lds = 0;
    std::vector<double> h_A;
    // inner iterations (inner_kernels.h), made on first use
    bool inner_ready = false;
    int inner_sets = 0;
    std::vector<int> inner_types;    // per set: bit t set when the set holds blocks of type t (camera, point, pose)
    int *d_ib_iblock = nullptr;      // one allocation each for the integers and the doubles below
    double *d_ib_dblock = nullptr;
    int *d_pt_off = nullptr, *d_pt_obs = nullptr, *d_blk_group = nullptr, *d_ib_flag = nullptr;
    double *d_ib_active = nullptr, *d_ib_X = nullptr;   // blocks that go on; exchange buffer of the shared blocks' sums
    std::vector<double> x_stage;     // host staging of an exchange over a host-buffer hook
    double *d_ib_H = nullptr, *d_ib_st = nullptr, *d_ib_xc = nullptr, *d_ib_cc = nullptr, *d_ib_slab = nullptr, *d_ib_out = nullptr;
    double *d_ib_ptslab = nullptr, *d_pt_ox = nullptr, *d_pt_oy = nullptr;
    int *d_pt_view = nullptr, *d_pt_cam = nullptr;
    int pt_slices = 1;               // workgroups per point block
    double *h_ib_active = nullptr;   // pinned: [0] blocks that go on, [1..2] step norm parts, [4..5] the polled count and its sequence number
    long long ib_seq = 0;
    // the host half of the first use (ordering, the points' observation lists), possibly made ahead on a host thread
    struct InnerHostPrep {
        int status = CVX_OK, sets = 0;
        std::vector<int> group, types, pt_off, pt_obs;
    };
    std::future<InnerHostPrep> inner_ahead;
    // exchange
    cvx_allreduce_fn xfn = nullptr;
    void *xctx = nullptr;
    int rank = 0, nranks = 1, xdev = 0;
    // timing
    bool timing = false;
    hipEvent_t ev[PH_COUNT][2];
    bool ev_made = false;
    double ph_ms[PH_COUNT] = {0, 0, 0, 0, 0, 0};
    long long ph_n[PH_COUNT] = {0, 0, 0, 0, 0, 0};
    bool ph_used[PH_COUNT] = {false, false, false, false, false, false};

    // One device allocation for everything cvx_calib_create makes (a hipMalloc / hipFree pair per buffer costs ~40 us:
    // 2 ms of a destroy, 0.3 ms of a create at 45 buffers); buffers that later calls re-size (E, X, the inner
    // iterations' blocks, the best point, the evaluation output) stay allocations of their own.  A request the
    // arena cannot hold falls back to hipMalloc, so its size estimate is a matter of speed, not of correctness.
    char *arena = nullptr;
    size_t arena_cap = 0, arena_used = 0;
    int arena_fallbacks = 0;
    bool in_arena(const void *p) const { return arena && (const char *)p >= arena && (const char *)p < arena + arena_cap; }
    template <class Tp>
    int dalloc(Tp **p, size_t n) {
        const size_t bytes = (std::max<size_t>(n, 1) * sizeof(Tp) + 255) / 256 * 256;
        if (arena && arena_used + bytes <= arena_cap) {
            *p = reinterpret_cast<Tp *>(arena + arena_used);
            arena_used += bytes;
            return CVX_OK;
        }
        ++arena_fallbacks;
        return dev_alloc(p, n);
    }
    template <class Tp>
    int dupload(Tp **p, const Tp *src, size_t n, hipStream_t s) {
        const int st = dalloc(p, n);
        if (st) return st;
        if (n) CVX_HIP(hipMemcpyAsync(*p, src, n * sizeof(Tp), hipMemcpyHostToDevice, s));
        return CVX_OK;
    }

    ~cvx_calib();
};

cvx_calib::~cvx_calib() {
    if (inner_ahead.valid()) inner_ahead.wait();     // the thread reads this handle's host tables
    (void)hipSetDevice(device);
    PhaseClock pc;
    // nothing of this handle in flight when its block and its stream go to the next one (a failed create may leave work)
    bool idle = true;
    if (stream && hipStreamSynchronize(stream) != hipSuccess) { (void)hipGetLastError(); idle = false; }
    if (d_E == d_E_res) d_E_res = nullptr;       // (one buffer, listed once below)
    void *ptrs[] = {d_E_res, d_view_off, d_cam_of_view, d_point_id, d_view_id, d_grp_v0, d_chk_row0, d_clist, d_clist_off, d_glist,
                    d_glist_off, d_obs_x, d_obs_y, d_intr[0], d_points[0], d_poses[0], d_rt[0], d_lm, d_Z, d_Linv, d_Dv, d_gv, d_s2pose, d_slab_pt, d_slab_cam, d_grp_scal,
                    d_slab_tile, d_Tpt, d_Tt, d_Gcam, d_E, d_ds_emb, d_dposes, d_out_view, d_cost_partial,
                    d_E2, d_eval, d_best, d_A, d_s2sh, d_Ds, d_delta_s, d_fail, d_flags, d_k4tasks, d_X, d_ib_iblock, d_ib_dblock};
    for (void *p : ptrs)
        if (p && !in_arena(p)) (void)hipFree(p);
    arena_give(device, arena, arena_cap);
    pc.mark("  device frees");
    if (h_E && h_E != h_E_res) (void)hipHostFree(h_E);
    if (h_E2) (void)hipHostFree(h_E2);          // heads the handle's one pinned block: h_lm, h_ib_active, h_ds_emb, h_params behind it
    pc.mark("  pinned frees");
    if (ev_made)
        for (int i = 0; i < PH_COUNT; ++i) { (void)hipEventDestroy(ev[i][0]); (void)hipEventDestroy(ev[i][1]); }
    if (ev_step) (void)hipEventDestroy(ev_step);
    if (ev_params) (void)hipEventDestroy(ev_params);
    if (own_stream && stream) {
        if (idle) cvx::stream_give(device, stream);
        else (void)hipStreamDestroy(stream);
    }
    pc.mark("  events, stream");
}

namespace {

inline void ph_begin(cvx_calib *h, int ph) {
    if (h->timing) { (void)hipEventRecord(h->ev[ph][0], h->stream); }
}
inline void ph_end(cvx_calib *h, int ph) {
    if (h->timing) { (void)hipEventRecord(h->ev[ph][1], h->stream); h->ph_used[ph] = true; }
}
// call after a stream synchronisation
inline void ph_collect(cvx_calib *h) {
    if (!h->timing) return;
    for (int i = 0; i < PH_COUNT; ++i)
        if (h->ph_used[i]) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, h->ev[i][0], h->ev[i][1]) == hipSuccess) { h->ph_ms[i] += ms; h->ph_n[i]++; }
            h->ph_used[i] = false;
        }
}

static_assert(CVX_MAX_MARKERS_PER_VIEW <= 512, "k_normal_eq runs one observation per lane in workgroups of at most 512");

template <int MODE>
int launch_normal_eq(cvx_calib *h, const K2Args &a) {
    dim3 grid(h->n_groups);
#define CVX_LAUNCH(TT, AL)                                                                                    \
    do {                                                                                                      \
        /* the limit is per function and process: raised when this handle needs more than any before it (the   \
           call sits on the host's path between a step's result and the next step's first launch) */          \
        static std::atomic<size_t> lds_limit[16];                                                             \
        std::atomic<size_t> &lim = lds_limit[h->device & 15];                                                 \
        if (h->k2_lds > lim.load(std::memory_order_relaxed)) {                                                \
            CVX_HIP(hipFuncSetAttribute((const void *)k_normal_eq<TT, MODE, AL>,                              \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k2_lds));         \
            lim.store(h->k2_lds, std::memory_order_relaxed);                                                  \
        }                                                                                                     \
        hipLaunchKernelGGL((k_normal_eq<TT, MODE, AL>), grid, dim3(TT), h->k2_lds, h->stream, a);             \
    } while (0)
#define CVX_CASE(TT)                                                      \
    case TT:                                                              \
        if (h->aligned) CVX_LAUNCH(TT, true); else CVX_LAUNCH(TT, false); \
        break;
    switch (h->T) {
        CVX_CASE(64)
        CVX_CASE(128)
        CVX_CASE(256)
        CVX_CASE(512)
        default:
            return CVX_ERR_UNSUPPORTED;
    }
#undef CVX_CASE
#undef CVX_LAUNCH
    CVX_HIP(hipGetLastError());
    return CVX_OK;
}

K2Args make_k2(cvx_calib *h, const cvx_damping &d, bool spec = false) {
    K2Args a;
    a.view_off = h->d_view_off; a.cam_of_view = h->d_cam_of_view; a.point_id = h->d_point_id;
    a.obs_x = h->d_obs_x; a.obs_y = h->d_obs_y;
    a.intr = h->d_intr[h->cur]; a.points = h->d_points[h->cur]; a.rt = h->d_rt[h->cur]; a.s2pose = h->d_s2pose;
    a.grp_v0 = h->d_grp_v0;
    a.P = h->P; a.ZC = h->ZC; a.camcol0 = h->camcol0; a.ycol = h->ycol; a.cam_free = h->cam_free; a.pts_free = h->pts_free;
    a.inv_radius = d.inv_radius; a.mu = d.mu; a.min_diag = d.min_diag; a.max_diag = d.max_diag;
    a.Z = h->d_Z; a.Lfac = h->d_Linv; a.Dv = h->d_Dv; a.gv = h->d_gv; a.s2pose_out = nullptr;
    a.slab_pt = h->d_slab_pt; a.slab_cam = h->d_slab_cam; a.grp_scal = h->d_grp_scal;
    a.st = nullptr; a.intr_step = h->intr_step; a.points_step = h->points_step; a.rt_step = h->rt_step;
    if (spec) {         // buffer 0's pointers: the kernel picks the current buffer and the damping from the record
        a.st = h->d_lm; a.intr = h->d_intr[0]; a.points = h->d_points[0]; a.rt = h->d_rt[0];
    }
    return a;
}

// rotation tables of parameter buffer `which` (skipped while they are still valid for it)
int launch_pose_prepare(cvx_calib *h, int which) {
    if (h->V == 0 || h->rt_valid[which]) return CVX_OK;
    h->rt_valid[which] = true;
    hipLaunchKernelGGL(k_pose_prepare, dim3((h->V + 127) / 128), dim3(128), 0, h->stream, h->d_poses[which], h->d_rt[which], h->V);
    CVX_HIP(hipGetLastError());
    return CVX_OK;
}

// Enqueue one linearisation at the current point: pose tables, normal-equation blocks, Schur
// contraction, fixed-order slab sums, assembly of E, and the exchange step.  Nothing is copied to
// the host (a host-buffer exchange hook is the exception: E makes the round trip the hook asks for).
// K3: slab[chunk] = Z_chunk^T Z_chunk over the 16x16 blocks of the upper triangle (reads Z, writes the slab only)
int launch_schur(cvx_calib *h) {
    K3Args k;
    k.Z = h->d_Z; k.chk_row0 = h->d_chk_row0; k.slab = h->d_slab_tile;
    k.ZC = h->ZC; k.ZCp = h->ZCp; k.KB = h->KB; k.n_chunks = h->n_chunks; k.nb = h->nb; k.nblk = h->nblk; k.parts = h->parts;
    const int blocks = 8 * ((h->n_chunks + 7) / 8) * h->parts;
    if (h->d_k4tasks) {
        K4Args q;
        q.Z = h->d_Z; q.chk_row0 = h->d_chk_row0; q.tasks = h->d_k4tasks; q.slab = h->d_slab_tile;
        q.ZC = h->ZC; q.ZCp = K4_PITCH; q.KB = K4_KB; q.n_chunks = h->n_chunks; q.nb = h->nb; q.nblk = h->nblk; q.parts = h->parts;
        q.zero_row = 6 * h->V;      // d_Z ends in eight rows that stay zero
        CVX_HIP(hipFuncSetAttribute((const void *)k_schur_syrk4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
        hipLaunchKernelGGL(k_schur_syrk4, dim3(blocks), dim3(64 * K4_WAVES), h->k3_lds, h->stream, q);
        CVX_HIP(hipGetLastError());
        return CVX_OK;
    }
    CVX_HIP(hipFuncSetAttribute((const void *)k_schur_syrk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->k3_lds));
    hipLaunchKernelGGL(k_schur_syrk, dim3(blocks), dim3(64 * K3_WAVES), h->k3_lds, h->stream, k);
    CVX_HIP(hipGetLastError());
    return CVX_OK;
}

// spec: a speculated step (calib_kernels.h, LmDev) -- the current point's tables are there by construction (the step
// before made its candidate's), K2 and the assembly read the record
int linearize(cvx_calib *h, const cvx_damping &d, bool init_mode, bool write_scale, bool spec = false) {
    if (h->V == 0) { cvx::set_last_error("the handle holds no views: nothing to linearise"); return CVX_ERR_INVALID_ARG; }
    int st = spec ? CVX_OK : launch_pose_prepare(h, h->cur);
    if (st) return st;
    K2Args a = make_k2(h, d, spec);
    ph_begin(h, PH_NORMAL);
    if (init_mode) {
        a.s2pose_out = write_scale ? h->d_s2pose : nullptr;
        st = launch_normal_eq<1>(h, a);
    } else {
        st = launch_normal_eq<0>(h, a);
    }
    if (st) return st;
    ph_end(h, PH_NORMAL);
    const bool schur = !init_mode && h->ns > 0;
    if (schur) {
        ph_begin(h, PH_SCHUR);
        st = launch_schur(h);
        if (st) return st;
        ph_end(h, PH_SCHUR);
    }
    ph_begin(h, PH_ASSEMBLE);
    double *scal = h->d_E + h->E_scal;
    {
        // fixed-order slab sums per camera and the scalars of this linearisation, one launch
        ReduceArgs ra;
        ra.njobs = 0; ra.C = h->C;
        auto add = [&](const double *src, int n_out, const int *list, const int *off, double *dst, bool wide = false) {
            SumJob &j = ra.job[ra.njobs++];
            j.wide = wide ? 1 : 0;
            j.src = src; j.stride = (size_t)n_out; j.n_out = n_out; j.blocks_x = wide ? (n_out + 511) / 512 : (n_out + 127) / 128;   // n_out is even (36 per point, 256 per tile)
            j.list = list; j.list_off = off; j.dst = dst;
        };
        if (schur) add(h->d_slab_tile, h->nblk * 256, h->d_clist, h->d_clist_off, h->d_Tt, true);
        if (h->pts_free) add(h->d_slab_pt, h->P * PT_ACC, h->d_glist, h->d_glist_off, h->d_Tpt);
        add(h->d_slab_cam, 256, h->d_glist, h->d_glist_off, h->d_Gcam);
        int blocks = 1;
        for (int i = 0; i < ra.njobs; ++i) blocks += ra.job[i].blocks_x * h->C;
        ra.slab_cam = h->d_slab_cam; ra.grp_scal = h->d_grp_scal; ra.n_groups = h->n_groups;
        ra.scal = scal; ra.rank = h->rank; ra.nranks = h->nranks; ra.fail = h->d_fail;
        hipLaunchKernelGGL(k_reduce_all, dim3(blocks), dim3(256), 0, h->stream, ra);
        CVX_HIP(hipGetLastError());
    }
    if (h->ns > 0) {
        AsmArgs s;
        s.Tt = schur ? h->d_Tt : nullptr; s.Tpt = h->d_Tpt; s.Gcam = h->d_Gcam;
        s.ns = h->ns; s.C = h->C; s.P = h->P; s.ncf = h->ncf; s.fix_k3 = h->fix_k3; s.pts_free = h->pts_free;
        s.ZC = h->ZC; s.camcol0 = h->camcol0; s.ycol = h->ycol; s.nb = h->nb; s.nblk = h->nblk;
        s.E = h->d_E;
        // one rank: nothing lies between the assembly and the damped matrix of the solve, so it is made here
        h->A_ready = !init_mode && h->nranks == 1;
        s.A = h->A_ready ? h->d_A : nullptr; s.Ds = h->d_Ds; s.s2 = h->d_s2sh; s.npad = h->npad;
        s.inv_radius = d.inv_radius; s.mu = d.mu; s.min_diag = d.min_diag; s.max_diag = d.max_diag;
        s.X = (h->nranks > 1) ? h->d_X : nullptr; s.n_scal = 2 + 2 * h->nranks;
        s.st = spec ? h->d_lm : nullptr;
        const long long pitch = h->A_ready ? h->npad : h->ns;
        const long long n = pitch * pitch + h->ns;
        hipLaunchKernelGGL(k_assemble, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, s);
        CVX_HIP(hipGetLastError());
    }
    ph_end(h, PH_ASSEMBLE);
    // exchange step (view-sharded runs): one all-reduce of X = [lower triangle of S | gred | gs | diagU | scalars];
    // the reduced tail then goes back behind E's (local) S, where the solve and the host read it
    if (h->nranks > 1 && h->xfn) {
        if (h->ns == 0) {
            // no shared unknowns: X is the scalars only and k_assemble did not run
            CVX_HIP(hipMemcpyAsync(h->d_X, h->d_E, h->X_count * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        }
        if (h->xdev) {
            if (h->xfn(h->xctx, h->d_X, h->X_count, 1, (void *)h->stream)) return CVX_ERR_EXCHANGE;
        } else {
            CVX_HIP(hipMemcpyAsync(h->h_E, h->d_X, h->X_count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            CVX_HIP(hipStreamSynchronize(h->stream));
            if (h->xfn(h->xctx, h->h_E, h->X_count, 0, nullptr)) return CVX_ERR_EXCHANGE;
            CVX_HIP(hipMemcpyAsync(h->d_X, h->h_E, h->X_count * sizeof(double), hipMemcpyHostToDevice, h->stream));
        }
        CVX_HIP(hipMemcpyAsync(h->d_E + (size_t)h->ns * h->ns, h->d_X + h->X_tail, (h->X_count - h->X_tail) * sizeof(double),
                               hipMemcpyDeviceToDevice, h->stream));
    }
    return CVX_OK;
}

// 1/2 sum r^2 of parameter buffer `which` over this rank's observations
int cost_of(cvx_calib *h, int which, double *cost) {
    int st = launch_pose_prepare(h, which);
    if (st) return st;
    hipLaunchKernelGGL(k_cost, dim3(h->n_cost_blocks), dim3(256), 0, h->stream, h->d_view_id, h->d_cam_of_view,
                       h->d_point_id, h->d_obs_x, h->d_obs_y, h->d_intr[which], h->d_points[which], h->d_rt[which],
                       (long long)h->N, h->d_cost_partial);
    CVX_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_final_scalars, dim3(1), dim3(1024), 0, h->stream, h->d_cost_partial, h->n_cost_blocks,
                       h->d_out_view, 0, h->d_E2, (double *)nullptr, 0.0);
    CVX_HIP(hipGetLastError());
    CVX_HIP(hipMemcpyAsync(h->h_E2, h->d_E2, 4 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    *cost = 0.5 * h->h_E2[0];
    return CVX_OK;
}

// sum of `count` doubles of a device buffer over the ranks, in place, through the hook in the mode it was registered for
int exchange_sum(cvx_calib *h, double *d, size_t count) {
    if (h->nranks <= 1 || count == 0) return CVX_OK;
    if (h->xdev) return h->xfn(h->xctx, d, count, 1, (void *)h->stream) ? CVX_ERR_EXCHANGE : CVX_OK;
    h->x_stage.resize(count);
    CVX_HIP(hipMemcpyAsync(h->x_stage.data(), d, count * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    if (h->xfn(h->xctx, h->x_stage.data(), count, 0, nullptr)) return CVX_ERR_EXCHANGE;
    CVX_HIP(hipMemcpyAsync(d, h->x_stage.data(), count * sizeof(double), hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}
// the same for values that live on the host
int exchange_sum_host(cvx_calib *h, std::vector<double> &v) {
    if (h->nranks <= 1 || v.empty()) return CVX_OK;
    if (!h->xdev) return h->xfn(h->xctx, v.data(), v.size(), 0, nullptr) ? CVX_ERR_EXCHANGE : CVX_OK;
    // a device-buffer hook: through device memory (the pass's own four doubles when they suffice)
    const bool small = v.size() <= 4 && h->d_ib_out;
    double *d = small ? h->d_ib_out : nullptr;
    if (!small) {
        int st = dev_alloc(&d, v.size());
        if (st) return st;
    }
    hipError_t e = hipMemcpyAsync(d, v.data(), v.size() * sizeof(double), hipMemcpyHostToDevice, h->stream);
    int rc = 0;
    if (e == hipSuccess) rc = h->xfn(h->xctx, d, v.size(), 1, (void *)h->stream);
    if (e == hipSuccess && !rc) e = hipMemcpyAsync(v.data(), d, v.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess && !rc) e = hipStreamSynchronize(h->stream);
    if (!small) (void)hipFree(d);
    if (rc) return CVX_ERR_EXCHANGE;
    CVX_HIP(e);
    return CVX_OK;
}

void free_inner(cvx_calib *h) {
    if (h->d_ib_iblock && !h->in_arena(h->d_ib_iblock)) (void)hipFree(h->d_ib_iblock);
    if (h->d_ib_dblock && !h->in_arena(h->d_ib_dblock)) (void)hipFree(h->d_ib_dblock);
    h->d_ib_iblock = nullptr; h->d_ib_dblock = nullptr;
    h->d_pt_off = h->d_pt_obs = h->d_blk_group = h->d_ib_flag = h->d_pt_view = h->d_pt_cam = nullptr;
    h->d_ib_active = h->d_ib_H = h->d_ib_st = h->d_ib_xc = h->d_ib_cc = h->d_ib_slab = h->d_ib_out = h->d_ib_ptslab = h->d_ib_X = nullptr;
    h->d_pt_ox = h->d_pt_oy = nullptr;
    h->inner_ready = false;
}

// what a point block reads per observation, laid out in the order of its list
__global__ void k_inner_gather(const int *__restrict__ pt_obs, const int *__restrict__ view_id, const int *__restrict__ cam_of_view,
                               const double *__restrict__ obs_x, const double *__restrict__ obs_y, long long n,
                               int *__restrict__ pt_view, int *__restrict__ pt_cam, double *__restrict__ pt_ox, double *__restrict__ pt_oy) {
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int o = pt_obs[k], v = view_id[o];
    pt_view[k] = v; pt_cam[k] = cam_of_view[v]; pt_ox[k] = obs_x[o]; pt_oy[k] = obs_y[o];
}

// ---- inner iterations (calibration.cpp:424): ordering, per-point observation lists and block state, on first use.
// View-sharded runs (a collective: every rank gets here at its first pass): the ordering is a function of the WHOLE
// problem's graph, so the ranks first put their views' structure together -- camera, marker count and point ids of
// every view, each rank's part at its offset of a zero buffer, one all-reduce -- and every rank orders the whole
// problem; rank by rank the views follow each other as in the unsharded problem, so the sets are the same as there.
// the points' observation lists: ascending observation = ascending view
void inner_point_lists(const cvx_calib *h, std::vector<int> &pt_off, std::vector<int> &pt_obs) {
    const std::vector<int> &point_id = h->h_point_id;
    pt_off.assign((size_t)h->P + 1, 0); pt_obs.resize((size_t)h->N);
    for (int64_t o = 0; o < h->N; ++o) pt_off[point_id[o] + 1]++;
    for (int p = 0; p < h->P; ++p) pt_off[p + 1] += pt_off[p];
    std::vector<int> pos(pt_off.begin(), pt_off.end() - 1);
    for (int64_t o = 0; o < h->N; ++o) pt_obs[pos[point_id[o]]++] = (int)o;
}

// One-rank handles: everything of the first use that needs no device, a function of tables that do not change after
// cvx_calib_create -- so it can run on a host thread while the caller goes on (start_inner_ahead).
cvx_calib::InnerHostPrep inner_host_prep(const cvx_calib *h) {
    cvx_calib::InnerHostPrep r;
    PhaseClock pc;
    const int C = h->C, P = h->P, V = h->V, NB = C + P + V;
    r.group.resize(NB);
    // the two halves are independent (1.1 and 0.7 ms at config 4): the lists on a thread of their own.  The ordering
    // checks the tables; the lists index by point id, so they wait for nothing but are only used if it passed.
    std::future<void> lists;
    bool ids_ok = true;
    for (int64_t o = 0; o < h->N && ids_ok; ++o) ids_ok = h->h_point_id[o] >= 0 && h->h_point_id[o] < P;
    bool lists_inline = false;
    if (ids_ok) {
        try { lists = std::async(std::launch::async, [h, &r] { inner_point_lists(h, r.pt_off, r.pt_obs); }); }
        catch (...) { lists_inline = true; }        // no thread to be had: one after the other
    }
    r.status = cvx_calib_inner_ordering(C, V, P, h->h_cam_of_view.data(), h->h_view_off.data(), h->h_point_id.data(), h->fix_intr,
                                        h->fix_pts, r.group.data(), &r.sets);
    pc.mark("inner host: ordering");
    if (!r.status) {
        r.types.assign(r.sets, 0);
        for (int i = 0; i < NB; ++i)
            if (r.group[i] >= 0) r.types[r.group[i]] |= 1 << (i < C ? cvxi::IB_CAM : i < C + P ? cvxi::IB_POINT : cvxi::IB_POSE);
    }
    if (lists.valid()) lists.get();
    else if (lists_inline && !r.status) inner_point_lists(h, r.pt_off, r.pt_obs);
    pc.mark("inner host: point lists (wait)");
    return r;
}

void start_inner_ahead(cvx_calib *h) {
    if (h->inner_ready || h->inner_ahead.valid() || h->nranks > 1) return;
    try { h->inner_ahead = std::async(std::launch::async, inner_host_prep, (const cvx_calib *)h); }
    catch (...) {}      // no thread to be had: the first use makes it itself
}

int ensure_inner(cvx_calib *h) {
    if (h->inner_ready) return CVX_OK;
    PhaseClock pc;
    free_inner(h);
    const int C = h->C, P = h->P, V = h->V, NB = C + P + V;
    const std::vector<int> &point_id = h->h_point_id;
    std::vector<int> group(NB), pt_off, pt_obs;
    int st;
    if (h->nranks > 1) {
        if (h->inner_ahead.valid()) (void)h->inner_ahead.get();      // made for the unsharded problem: of no use now
        const int R = h->nranks;
        std::vector<double> cnt(2 * (size_t)R, 0.0);
        cnt[2 * h->rank] = V; cnt[2 * h->rank + 1] = (double)h->N;
        st = exchange_sum_host(h, cnt);
        if (st) return st;
        int64_t Vt = 0, Nt = 0, v0 = 0, o0 = 0;
        for (int r = 0; r < R; ++r) {
            if (r == h->rank) { v0 = Vt; o0 = Nt; }
            Vt += (int64_t)cnt[2 * r]; Nt += (int64_t)cnt[2 * r + 1];
        }
        if (Vt > 0x7fffffff || Nt > 0x7fffffff) return CVX_ERR_UNSUPPORTED;
        std::vector<double> all((size_t)(2 * Vt + Nt), 0.0);        // cameras | marker counts | point ids
        for (int v = 0; v < V; ++v) {
            all[(size_t)(v0 + v)] = h->h_cam_of_view[v];
            all[(size_t)(Vt + v0 + v)] = h->h_view_off[v + 1] - h->h_view_off[v];
        }
        for (int64_t o = 0; o < h->N; ++o) all[(size_t)(2 * Vt + o0 + o)] = point_id[o];
        st = exchange_sum_host(h, all);
        if (st) return st;
        std::vector<int> gcam((size_t)Vt), goff((size_t)Vt + 1, 0), gpid((size_t)Nt), ggroup((size_t)(C + P + Vt));
        for (int64_t v = 0; v < Vt; ++v) { gcam[v] = (int)all[(size_t)v]; goff[v + 1] = goff[v] + (int)all[(size_t)(Vt + v)]; }
        for (int64_t o = 0; o < Nt; ++o) gpid[o] = (int)all[(size_t)(2 * Vt + o)];
        st = cvx_calib_inner_ordering(C, (int)Vt, P, gcam.data(), goff.data(), gpid.data(), h->fix_intr, h->fix_pts,
                                      ggroup.data(), &h->inner_sets);
        if (st) return st;
        // what a set holds is decided on the whole problem: every rank then runs the same kernels and exchanges
        h->inner_types.assign(h->inner_sets, 0);
        for (int64_t i = 0; i < C + P + Vt; ++i)
            if (ggroup[i] >= 0) h->inner_types[ggroup[i]] |= 1 << (i < C ? cvxi::IB_CAM : i < C + P ? cvxi::IB_POINT : cvxi::IB_POSE);
        for (int i = 0; i < C + P; ++i) group[i] = ggroup[i];
        for (int v = 0; v < V; ++v) group[C + P + v] = ggroup[(size_t)(C + P + v0 + v)];
        inner_point_lists(h, pt_off, pt_obs);
    } else {
        cvx_calib::InnerHostPrep r = h->inner_ahead.valid() ? h->inner_ahead.get() : inner_host_prep(h);
        if (r.status) return r.status;
        h->inner_sets = r.sets;
        h->inner_types.swap(r.types); group.swap(r.group); pt_off.swap(r.pt_off); pt_obs.swap(r.pt_obs);
    }
    pc.mark("inner: host half (or its wait)");
    hipStream_t s = h->stream;
    // a point's list is cut so that a workgroup of 256 lanes sees about two observations per lane
    h->pt_slices = (int)std::min<int64_t>(64, std::max<int64_t>(1, h->N / P / 512));
    // two allocations, carved in 256-byte steps
    const size_t N = (size_t)h->N;
    auto up = [](size_t n, size_t unit) { return (n + unit - 1) / unit * unit; };
    const size_t i_sizes[] = {(size_t)P + 1, N, N, N, (size_t)NB, 4 * (size_t)NB};
    const size_t d_sizes[] = {8, (size_t)NB * cvxi::IB_STRIDE, (size_t)NB * cvxi::IB_STATE, 9 * (size_t)NB, (size_t)NB,
                              (size_t)std::max(1, h->n_groups) * cvxi::IB_STRIDE, (size_t)(C + P) * cvxi::IB_STRIDE,
                              (size_t)P * h->pt_slices * cvxi::PT_SLOT, N, N};
    size_t i_tot = 0, d_tot = 0;
    for (size_t n : i_sizes) i_tot += up(std::max<size_t>(n, 1), 64);
    for (size_t n : d_sizes) d_tot += up(std::max<size_t>(n, 1), 32);
    // from the arena when cvx_calib_create left room for them (desc.prepare_inner_iterations), else allocations of their own
    st = h->dalloc(&h->d_ib_iblock, i_tot); if (st) return st;
    st = h->dalloc(&h->d_ib_dblock, d_tot); if (st) return st;
    {
        int *ip = h->d_ib_iblock;
        int **iv[] = {&h->d_pt_off, &h->d_pt_obs, &h->d_pt_view, &h->d_pt_cam, &h->d_blk_group, &h->d_ib_flag};
        for (int k = 0; k < 6; ++k) { *iv[k] = ip; ip += up(std::max<size_t>(i_sizes[k], 1), 64); }
        double *dp = h->d_ib_dblock;
        double **dv[] = {&h->d_ib_active, &h->d_ib_H, &h->d_ib_st, &h->d_ib_xc, &h->d_ib_cc, &h->d_ib_slab, &h->d_ib_X,
                         &h->d_ib_ptslab, &h->d_pt_ox, &h->d_pt_oy};
        for (int k = 0; k < 10; ++k) { *dv[k] = dp; dp += up(std::max<size_t>(d_sizes[k], 1), 32); }
        h->d_ib_out = h->d_ib_active + 4;       // the first eight doubles: running count | pad | four of the pass's own
    }
    pc.mark("inner: allocations");
    CVX_HIP(hipMemcpyAsync(h->d_pt_off, pt_off.data(), pt_off.size() * sizeof(int), hipMemcpyHostToDevice, s));
    if (N) CVX_HIP(hipMemcpyAsync(h->d_pt_obs, pt_obs.data(), N * sizeof(int), hipMemcpyHostToDevice, s));
    CVX_HIP(hipMemcpyAsync(h->d_blk_group, group.data(), group.size() * sizeof(int), hipMemcpyHostToDevice, s));
    if (N) {
        hipLaunchKernelGGL(k_inner_gather, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, s, h->d_pt_obs, h->d_view_id,
                           h->d_cam_of_view, h->d_obs_x, h->d_obs_y, (long long)N, h->d_pt_view, h->d_pt_cam, h->d_pt_ox, h->d_pt_oy);
        CVX_HIP(hipGetLastError());
    }
    CVX_HIP(hipMemsetAsync(h->d_ib_flag, 0, 4 * (size_t)NB * sizeof(int), s));
    CVX_HIP(hipMemsetAsync(h->d_ib_st, 0, (size_t)NB * cvxi::IB_STATE * sizeof(double), s));
    pc.mark("inner: uploads, gather enqueued");
    CVX_HIP(hipStreamSynchronize(s));       // the uploads read host vectors that end here
    pc.mark("inner: synchronize");
    h->inner_ready = true;
    return CVX_OK;
}

// One pass of CoordinateDescentMinimizer::Minimize on parameter buffer `which`, in place: set after set, every
// block of the set advances one LM iteration per round until none goes on (at most 50 iterations each, plus the
// rounds a block spends on invalid steps: bounded by the round limit below).
// View-sharded runs: pose blocks are a rank's own; a camera's or a point's observations are spread over the ranks,
// so in a set that holds such blocks every round all-reduces their sums (H | g | cost rows, then the candidates'
// costs) and the count of blocks that go on -- the ranks then take the same decisions and leave the set together.
// A set of pose blocks only runs a rank-local number of rounds, so its round limit can strike on one rank alone: on a
// sharded handle that failure is handed back in *rank_local_status and the pass goes on through the remaining sets'
// collectives; the caller agrees on it with the other ranks (agree_on_status) so that all return the error together.
int inner_pass(cvx_calib *h, int which, int *rank_local_status) {
    *rank_local_status = CVX_OK;
    int st = ensure_inner(h);
    if (st) return st;
    const int NB = h->C + h->P + h->V, nsh = h->C + h->P;
    const bool sharded = h->nranks > 1 && h->xfn;
    cvxi::InnerArgs a;
    a.view_off = h->d_view_off; a.cam_of_view = h->d_cam_of_view; a.point_id = h->d_point_id; a.view_id = h->d_view_id;
    a.obs_x = h->d_obs_x; a.obs_y = h->d_obs_y; a.pt_off = h->d_pt_off; a.pt_obs = h->d_pt_obs;
    a.pt_view = h->d_pt_view; a.pt_cam = h->d_pt_cam; a.pt_ox = h->d_pt_ox; a.pt_oy = h->d_pt_oy;
    a.grp_v0 = h->d_grp_v0; a.glist = h->d_glist; a.glist_off = h->d_glist_off;
    a.intr = h->d_intr[which]; a.points = h->d_points[which]; a.poses = h->d_poses[which]; a.rt = h->d_rt[which];
    a.blk_group = h->d_blk_group; a.H = h->d_ib_H; a.st = h->d_ib_st; a.xc = h->d_ib_xc; a.cc = h->d_ib_cc;
    a.flag = h->d_ib_flag; a.cam_slab = h->d_ib_slab; a.active = h->d_ib_active;
    a.pt_slab = h->d_ib_ptslab; a.pt_slices = h->pt_slices;
    a.C = h->C; a.P = h->P; a.V = h->V; a.ncf = h->ncf; a.fix_k3 = h->fix_k3;
    const dim3 per_block((NB + 63) / 64), t64(64);
    for (int set = 0; set < h->inner_sets; ++set) {
        const int types = h->inner_types[set];
        const bool cams = types & (1 << cvxi::IB_CAM), pts = types & (1 << cvxi::IB_POINT), poses = (types & (1 << cvxi::IB_POSE)) && h->V > 0;
        const bool shared = cams || pts;
        if (shared) {                       // they read the rotation tables of the point being improved
            st = launch_pose_prepare(h, which);
            if (st) return st;
        }
        a.cur = set;
        hipLaunchKernelGGL(cvxi::k_inner_reset, per_block, t64, 0, h->stream, a);
        for (int round = 0;; ++round) {
            if (round > 60) {
                cvx::set_last_error("inner iterations: a block did not terminate");
                if (sharded && !shared) { *rank_local_status = CVX_ERR_NUMERIC; break; }
                return CVX_ERR_NUMERIC;     // a shared set's round count is the same on every rank
            }
            // blocks of the shared kinds: partial sums, then one fixed-order sum per block (ids 0 .. C + P - 1)
            const dim3 pt_grid(h->P, h->pt_slices), shared_grid(nsh);
            if (poses) hipLaunchKernelGGL(cvxi::k_inner_pose<0>, dim3((h->V + 3) / 4), dim3(256), 0, h->stream, a);
            if (pts) hipLaunchKernelGGL(cvxi::k_inner_point<0>, pt_grid, dim3(cvxi::IP_THREADS), 0, h->stream, a);
            if (cams && h->n_groups) hipLaunchKernelGGL(cvxi::k_inner_cam<0>, dim3(h->n_groups), dim3(cvxi::IC_THREADS), 0, h->stream, a);
            if (shared) hipLaunchKernelGGL(cvxi::k_inner_sum<0>, shared_grid, dim3(cvxi::IS_THREADS), 0, h->stream, a);
            if (shared && sharded) {
                hipLaunchKernelGGL((cvxi::k_inner_xchg<0, 0>), shared_grid, t64, 0, h->stream, a, h->d_ib_X);
                st = exchange_sum(h, h->d_ib_X, (size_t)nsh * cvxi::IB_STRIDE);
                if (st) return st;
                hipLaunchKernelGGL((cvxi::k_inner_xchg<0, 1>), shared_grid, t64, 0, h->stream, a, h->d_ib_X);
            }
            hipLaunchKernelGGL(cvxi::k_inner_solve, per_block, t64, 0, h->stream, a);
            if (poses) hipLaunchKernelGGL(cvxi::k_inner_pose<1>, dim3((h->V + 3) / 4), dim3(256), 0, h->stream, a);
            if (pts) hipLaunchKernelGGL(cvxi::k_inner_point<1>, pt_grid, dim3(cvxi::IP_THREADS), 0, h->stream, a);
            if (cams && h->n_groups) hipLaunchKernelGGL(cvxi::k_inner_cam<1>, dim3(h->n_groups), dim3(cvxi::IC_THREADS), 0, h->stream, a);
            if (shared) hipLaunchKernelGGL(cvxi::k_inner_sum<1>, shared_grid, dim3(cvxi::IS_THREADS), 0, h->stream, a);
            if (shared && sharded) {
                hipLaunchKernelGGL((cvxi::k_inner_xchg<1, 0>), shared_grid, t64, 0, h->stream, a, h->d_ib_X);
                st = exchange_sum(h, h->d_ib_X, (size_t)nsh);
                if (st) return st;
                hipLaunchKernelGGL((cvxi::k_inner_xchg<1, 1>), shared_grid, t64, 0, h->stream, a, h->d_ib_X);
            }
            hipLaunchKernelGGL(cvxi::k_inner_decide, per_block, t64, 0, h->stream, a);
            CVX_HIP(hipGetLastError());
            // No set is through after one round but by accident (every block at a tolerance at once): the count is
            // not read then -- a round in which nothing runs costs less than the host's wait for the number.  (The same
            // on every rank: `round` is.)
            if (round == 0) continue;
            if (shared && sharded) {        // leave the set together (a set of pose blocks only has nothing to agree on)
                st = exchange_sum(h, h->d_ib_active, 1);
                if (st) return st;
            }
            if (h->d_E2_map && !h->timing && !(shared && sharded)) {
                // the count goes to host-mapped memory behind the round's last kernel; the host polls its sequence number
                double *slot = h->d_E2_map + (h->h_ib_active - h->h_E2) + 4;
                const double want = (double)++h->ib_seq;
                hipLaunchKernelGGL(cvxi::k_inner_publish, dim3(1), t64, 0, h->stream, h->d_ib_active, slot, want);
                CVX_HIP(hipGetLastError());
                volatile const double *flag = h->h_ib_active + 5;
                const auto t0 = std::chrono::steady_clock::now();
                long long spins = 0;
                while (*flag != want) {
                    if ((++spins & 0xffff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
                        CVX_HIP(hipStreamSynchronize(h->stream));       // surfaces a kernel fault as a HIP error
                        if (*flag != want) { cvx::set_last_error("inner iterations: a round's count never arrived in host memory"); return CVX_ERR_HIP; }
                    }
                }
                std::atomic_thread_fence(std::memory_order_acquire);
                if (h->h_ib_active[4] == 0.0) break;
                continue;
            }
            CVX_HIP(hipMemcpyAsync(h->h_ib_active, h->d_ib_active, sizeof(double), hipMemcpyDeviceToHost, h->stream));
            CVX_HIP(hipStreamSynchronize(h->stream));
            if (h->h_ib_active[0] == 0.0) break;
        }
        if (poses) h->rt_valid[which] = false;
    }
    return CVX_OK;
}

// max over the ranks of a status that may differ between them, riding as one more entry of an exchange the caller makes
// anyway (statuses are small non-negative integers; a sum over the ranks is non-zero iff any is)
int agreed_status(double summed) { return summed != 0.0 ? CVX_ERR_NUMERIC : CVX_OK; }

class CalibEvaluator : public cvx::Evaluator {
public:
    explicit CalibEvaluator(cvx_calib *h_) : h(h_) {}
    cvx_calib *h;
    // extra outputs of the last try_step, for cvx_calib_linear_step
    bool keep_dposes = false;
    double *want_S = nullptr;

    cvx::InitInfo init(bool jacobi) override {
        cvx::InitInfo out;
        if (!jacobi) {
            const long long n = 6LL * h->V;
            if (n) hipLaunchKernelGGL(k_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, h->d_s2pose, n, 1.0);
        }
        cvx_damping d0 = {0, 0, 0, 0};
        out.status = linearize(h, d0, true, jacobi);
        if (out.status) return out;
        // the tail of E: gred | gs | diagU | scalars -- into pinned memory (h_E is free between two exchanges); the
        // shared unknowns' scales are made from it on the device
        const size_t nn = (size_t)h->ns * h->ns, tail = h->E_count - nn;
        if (h->ns > 0) {
            hipLaunchKernelGGL(k_shared_scales, dim3((h->ns + 255) / 256), dim3(256), 0, h->stream, h->d_E + nn + 2 * (size_t)h->ns,
                               h->d_s2sh, h->ns, jacobi ? 1 : 0);
        }
        if (hipMemcpyAsync(h->h_E, h->d_E + nn, tail * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
            hipStreamSynchronize(h->stream) != hipSuccess) { out.status = CVX_ERR_HIP; return out; }
        ph_collect(h);
        const double *gs = h->h_E + h->ns, *dU = gs + h->ns, *scal = h->h_E + 3 * (size_t)h->ns;
        double gm = 0, dm = 0;
        for (int q = 0; q < h->nranks; ++q) { gm = std::max(gm, scal[2 + q]); dm = std::max(dm, scal[2 + h->nranks + q]); }
        for (int i = 0; i < h->ns; ++i) { gm = std::max(gm, std::fabs(gs[i])); dm = std::max(dm, dU[i]); }
        out.cost = 0.5 * scal[0]; out.grad_max = gm; out.diag_max = dm;
        return out;
    }

    cvx::StepInfo try_step(const cvx_damping &d, double gtol, bool gradient_only) override {
        cvx::StepInfo out;
        if (gradient_only) {
            // cost and gradient only (the test Ceres makes after the last accepted step): the undamped
            // linearisation without the Schur contraction, the reduced solve or a candidate
            cvx_damping d0 = {0, 0, 0, 0};
            out.status = linearize(h, d0, true, false);
            if (out.status) return out;
            const size_t nn = (size_t)h->ns * h->ns, tail = h->E_count - nn;
            if (hipMemcpyAsync(h->h_E, h->d_E + nn, tail * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
                hipStreamSynchronize(h->stream) != hipSuccess) { out.status = CVX_ERR_HIP; return out; }
            ph_collect(h);
            const double *gs = h->h_E + h->ns, *scal = h->h_E + 3 * (size_t)h->ns;
            double gm = 0;
            for (int q = 0; q < h->nranks; ++q) gm = std::max(gm, scal[2 + q]);
            for (int i = 0; i < h->ns; ++i) gm = std::max(gm, std::fabs(gs[i]));
            out.cost = 0.5 * scal[0];
            out.grad_max = gm;
            out.grad_converged = gm <= gtol;
            return out;
        }
        out.status = run_step(d);
        if (out.status) return out;
        // (the same reading of the step's 16 doubles as the device's: step_from_record)
        const cvx::TrStep ts = step_from_record(rec, gtol, &out.cost);
        out.grad_max = ts.grad_max;
        if (ts.grad_converged) { out.grad_converged = true; return out; }
        out.ok = ts.ok != 0;
        if (!out.ok && (rec[12] > 0 || rec[8] > 0)) return out;
        out.cand_cost = ts.cand_cost;
        out.model_change = ts.model_change;
        out.step_norm2 = ts.step_norm2;
        out.x_norm2 = ts.x_norm2;
        out.free_norm2 = rec[3] + rec[7];
        return out;
    }

    // strategy 0's state right before the try_step that follows: what a speculated step's record starts from
    void lm_state(const cvx::TrState &t, const cvx::TrOptions &o, const cvx_damping &d) override {
        snap_tr = t; snap_opt = o; snap_d = d; have_snap = true;
    }

    int accept() override {
        h->cur = 1 - h->cur;
        return CVX_OK;
    }

    // the candidate of the last try_step goes through one inner-iteration pass
    int inner_iterations(double *cost, double *step_norm2) override {
        const int nxt = 1 - h->cur;
        int local = CVX_OK;
        int st = inner_pass(h, nxt, &local);
        if (st) return st;
        hipLaunchKernelGGL(cvxi::k_inner_step_norm, dim3(1), dim3(1024), 0, h->stream, h->d_intr[h->cur], h->d_intr[nxt],
                           9LL * h->C, h->d_points[h->cur], h->d_points[nxt], 3LL * h->P, h->d_poses[h->cur], h->d_poses[nxt],
                           6LL * h->V, h->d_ib_out);
        CVX_HIP(hipGetLastError());
        CVX_HIP(hipMemcpyAsync(h->h_ib_active + 1, h->d_ib_out, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
        st = cost_of(h, nxt, cost);         // synchronises the stream
        if (st) return st;
        if (h->nranks > 1 && h->xfn) {      // the cost and the poses' part of the norm are this rank's share
            std::vector<double> v = {*cost, h->h_ib_active[2], local ? 1.0 : 0.0};
            st = exchange_sum_host(h, v);
            if (st) return st;
            *cost = v[0]; h->h_ib_active[2] = v[1];
            if (agreed_status(v[2])) {
                if (!local) cvx::set_last_error("inner iterations: a block did not terminate on another rank");
                return CVX_ERR_NUMERIC;
            }
        } else if (local) {
            return local;
        }
        *step_norm2 = h->h_ib_active[1] + h->h_ib_active[2];
        return CVX_OK;
    }

    // the lowest-cost point of a non-monotonic run: a third set of parameter buffers, made on first need
    int save_best() override {
        const size_t ni = 9 * (size_t)h->C, np = 3 * (size_t)h->P, nv = 6 * (size_t)h->V;
        if (!h->d_best) {
            int st = dev_alloc(&h->d_best, ni + np + nv);
            if (st) return st;
        }
        CVX_HIP(hipMemcpyAsync(h->d_best, h->d_intr[h->cur], ni * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        CVX_HIP(hipMemcpyAsync(h->d_best + ni, h->d_points[h->cur], np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (nv) CVX_HIP(hipMemcpyAsync(h->d_best + ni + np, h->d_poses[h->cur], nv * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        return CVX_OK;
    }
    int restore_best() override {
        if (!h->d_best) return CVX_ERR_INVALID_ARG;
        const size_t ni = 9 * (size_t)h->C, np = 3 * (size_t)h->P, nv = 6 * (size_t)h->V;
        CVX_HIP(hipMemcpyAsync(h->d_intr[h->cur], h->d_best, ni * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        CVX_HIP(hipMemcpyAsync(h->d_points[h->cur], h->d_best + ni, np * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        if (nv) CVX_HIP(hipMemcpyAsync(h->d_poses[h->cur], h->d_best + ni + np, nv * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
        h->rt_valid[h->cur] = false;
        CVX_HIP(hipStreamSynchronize(h->stream));
        return CVX_OK;
    }

private:
    // ---- speculated steps (calib_kernels.h, LmDev) ----
    cvx::TrState snap_tr;
    cvx::TrOptions snap_opt;
    cvx_damping snap_d = {0, 0, 0, 0};
    bool have_snap = false;
    bool spec_inflight = false;      // the NEXT step's kernels are enqueued already
    bool spec_runs = false;          // ... and the device said they would run
    long long spec_seq = 0;
    const double *rec = nullptr;     // the 16 doubles of the last step (host memory)

    // Opt-in (CVX_CALIB_SPECULATE=1 in the environment): measured on the box, the host's turn-around between two steps
    // is ~6 us, and taking the decision on the device puts ~9.5 us of its own on the critical path (the decision in the
    // last kernel 3.7 us, the candidate's rotation tables 5.8 us that the host's turn-around used to hide): 0.408 vs
    // 0.404 ms per iteration at config 4.  Otherwise: one rank, pose blocks present, strategy 0
    // past its inner iterations, no timing or debug outputs asked.
    bool can_speculate() const {
        const char *e = std::getenv("CVX_CALIB_SPECULATE");
        const bool on = e && e[0] == '1';
        return on && have_snap && !snap_tr.inner_enabled && h->nranks == 1 && h->V > 0 && h->ns > 0 && !h->timing &&
               h->d_E2_map && !want_S && !keep_dposes;
    }

    // all kernels of one linear step, enqueued: linearise, damp + factor + solve the reduced system, form the candidate,
    // back-substitute the poses, evaluate the candidate, publish 16 doubles, the candidate's rotation tables.
    // spec: every kernel that reads the point or the damping takes them from the record (and returns at once when the
    // record says the step is off); the 16 doubles go to host record `seq & 1`.
    int enqueue_step(const cvx_damping &d, bool spec, long long seq) {
        int st = linearize(h, d, false, false, spec);
        if (st) return st;
        const int ns = h->ns, nxt = 1 - h->cur;
        ph_begin(h, PH_SOLVE);
        if (ns > 0) {
            cvxs::DampArgs da;
            da.E = h->d_E; da.Sp = (h->nranks > 1) ? h->d_X : nullptr; da.s2 = h->d_s2sh; da.n = ns; da.npad = h->npad;
            da.inv_radius = d.inv_radius; da.mu = d.mu; da.min_diag = d.min_diag; da.max_diag = d.max_diag;
            da.A = h->d_A; da.Ds = h->d_Ds;
            const long long tot = (long long)h->npad * h->npad;
            if (!h->A_ready)
                hipLaunchKernelGGL(cvxs::k_damp_prepare, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, h->stream, da);
            if (want_S) {
                h->h_A.resize((size_t)tot);
                CVX_HIP(hipMemcpyAsync(h->h_A.data(), h->d_A, tot * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            }
            for (int k = 0; k < h->nT; ++k)
                hipLaunchKernelGGL(cvxs::k_chol_step, dim3(cvxs::chol_step_blocks(h->nT, k)), dim3(256), 0, h->stream, h->d_A,
                                   h->npad, ns, k, h->nT, h->d_fail, h->d_delta_s);
            CVX_HIP(hipGetLastError());
        }
        cvxs::FinishArgs f;
        f.n = ns;
        f.E = h->d_E; f.E_scal = (long long)h->E_scal; f.nranks = h->nranks; f.Ds = h->d_Ds; f.fail = h->d_fail;
        f.C = h->C; f.P = h->P; f.ncf = h->ncf; f.fix_k3 = h->fix_k3; f.cam_free = h->cam_free; f.pts_free = h->pts_free;
        f.ZC = h->ZC; f.camcol0 = h->camcol0; f.ycol = h->ycol;
        f.intr = h->d_intr[h->cur]; f.points = h->d_points[h->cur];
        f.intr_cand = h->d_intr[nxt]; f.points_cand = h->d_points[nxt];
        f.ds_emb = h->d_ds_emb; f.delta_s = h->d_delta_s; f.R = h->d_E2;
        f.st = nullptr; f.intr_step = h->intr_step; f.points_step = h->points_step; f.poses_step = h->poses_step;
        const double *poses_cur = h->d_poses[h->cur];
        double *poses_nxt = h->d_poses[nxt];
        if (spec) {
            f.st = h->d_lm; f.intr = h->d_intr[0]; f.points = h->d_points[0]; f.intr_cand = h->d_intr[0]; f.points_cand = h->d_points[0];
            poses_cur = h->d_poses[0]; poses_nxt = h->d_poses[0];
        }
        if (h->V == 0) {
            // no pose blocks: the step's bookkeeping is a kernel of its own
            if (h->fin_lds > 64 * 1024)
                CVX_HIP(hipFuncSetAttribute((const void *)cvxs::k_step_finish, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fin_lds));
            hipLaunchKernelGGL(cvxs::k_step_finish, dim3(1), dim3(cvxs::FIN_THREADS), h->fin_lds, h->stream, f);
            CVX_HIP(hipGetLastError());
        }
        ph_end(h, PH_SOLVE);
        ph_begin(h, PH_BACKSUBST);
        if (h->V > 0) {
            // pose steps, candidate poses and the candidate's cost per view in one pass; workgroup 0 also makes the
            // candidate's shared parameters and the scalars of the step (step_kernels.h)
            const size_t bsc_lds = h->fin_lds + sizeof(double) * 3 * (size_t)h->P;      // finish area | candidate points
            if (bsc_lds > 64 * 1024)
                CVX_HIP(hipFuncSetAttribute((const void *)k_backsubst_cost_step, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bsc_lds));
            hipLaunchKernelGGL(k_backsubst_cost_step, dim3((h->V + 3) / 4), dim3(BSC_THREADS), bsc_lds, h->stream, h->d_Z,
                               h->d_Linv, h->d_Dv, h->d_gv, h->d_cam_of_view, poses_cur, poses_nxt,
                               keep_dposes ? h->d_dposes : (double *)nullptr, h->d_out_view, h->V,
                               h->d_view_off, h->d_point_id, h->d_obs_x, h->d_obs_y, f);
            CVX_HIP(hipGetLastError());
        }
        ph_end(h, PH_BACKSUBST);
        // one rank, no phase timing: the result lands in host-mapped memory and the host polls its sequence number
        // (several ranks with a device-buffer hook: the same, published by a small kernel behind the all-reduce)
        const bool sharded_dev = h->nranks > 1 && h->xfn && h->xdev;
        polled = (h->nranks == 1 || sharded_dev) && !h->timing && h->d_E2_map;
        const double fseq = (double)seq;
        ph_begin(h, PH_COST);
        if (spec) {
            hipLaunchKernelGGL(k_final_scalars_decide, dim3(1), dim3(1024), 0, h->stream, h->d_cost_partial, 0, h->d_out_view, h->V,
                               h->d_E2, h->d_E2_map + 32 * (seq & 1), fseq, h->d_lm);
        } else {
            hipLaunchKernelGGL(k_final_scalars, dim3(1), dim3(1024), 0, h->stream, h->d_cost_partial, 0,
                               h->d_out_view, h->V, h->d_E2, (polled && !sharded_dev) ? h->d_E2_map : (double *)nullptr, fseq);
        }
        CVX_HIP(hipGetLastError());
        ph_end(h, PH_COST);
        if (sharded_dev) {
            if (h->xfn(h->xctx, h->d_E2, 4, 1, (void *)h->stream)) return CVX_ERR_EXCHANGE;
            if (polled) {
                hipLaunchKernelGGL(k_publish_scalars, dim3(1), dim3(64), 0, h->stream, h->d_E2, h->d_E2_map, fseq);
                CVX_HIP(hipGetLastError());
            }
        }
        if (!polled) {
            CVX_HIP(hipMemcpyAsync(h->h_E2, h->d_E2, 16 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
            CVX_HIP(hipEventRecord(h->ev_step, h->stream));
        }
        // the candidate's rotation tables (needed only if the step is accepted) are made while the host decides
        if (spec) {
            hipLaunchKernelGGL(k_pose_prepare_cand, dim3((h->V + 127) / 128), dim3(128), 0, h->stream, h->d_poses[0], h->d_rt[0], h->V,
                               h->poses_step, h->rt_step, h->d_lm);
            CVX_HIP(hipGetLastError());
        } else {
            h->rt_valid[nxt] = false;
            st = launch_pose_prepare(h, nxt);
            if (st) return st;
        }
        return CVX_OK;
    }
    bool polled = false;

    // wait for the 16 doubles of the step numbered seq (host record `slot`)
    int wait_step(long long seq, int slot) {
        rec = h->h_E2 + 32 * slot;
        if (polled) {
            volatile const double *flag = rec + 16;
            const double want = (double)seq;
            const auto t0 = std::chrono::steady_clock::now();
            long long spins = 0;
            while (*flag != want) {
                if ((++spins & 0xffff) == 0 &&
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
                    CVX_HIP(hipStreamSynchronize(h->stream));       // surfaces a kernel fault as a HIP error
                    if (*flag != want) { cvx::set_last_error("step result never arrived in host memory"); return CVX_ERR_HIP; }
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            CVX_HIP(hipEventSynchronize(h->ev_step));
        }
        ph_collect(h);
        if (h->nranks > 1 && h->xfn && !h->xdev) {
            if (h->xfn(h->xctx, h->h_E2, 4, 0, nullptr)) return CVX_ERR_EXCHANGE;
        }
        return CVX_OK;
    }

    // One linear step.  When the loop allows it (can_speculate) the step AFTER this one is enqueued before this one's
    // result is awaited: its kernels take the point and the damping from the device-side record, which this step's last
    // kernel moves on with the loop's own decision code.  The host then repeats that decision from the 16 doubles and
    // finds, next time round, its step already under way -- or, where the record said "off", enqueues it as ever.
    int run_step(const cvx_damping &d) {
        const bool spec = can_speculate();
        long long seq;
        int st;
        if (spec_inflight && spec_runs && spec) {
            // this step is the one enqueued behind the last: its damping was the device's to work out -- the same number
            seq = spec_seq;
            ++h->n_spec_used;
            const double *prev = h->h_E2 + 32 * ((seq - 1) & 1);
            if (prev[17] != d.inv_radius || (int)prev[18] != h->cur) {
                cvx::set_last_error("speculated step: the device-side trust-region state differs from the host's");
                return CVX_ERR_NUMERIC;
            }
        } else {
            if (spec_inflight) ++h->n_spec_dropped;
            seq = ++h->step_seq;
            if (spec) {
                // a chain of speculated steps starts here: the record is the host's state before this step's decision
                LmDev &r = *h->h_lm;
                r.inv_radius = d.inv_radius; r.mu = d.mu; r.min_diag = d.min_diag; r.max_diag = d.max_diag;
                r.cur = h->cur; r.active = 1; r.last_cand = -1; r.pad = 0;
                r.min_lm_diagonal = d.min_diag; r.max_lm_diagonal = d.max_diag;
                r.tr = snap_tr; r.opt = snap_opt;
                st = launch_pose_prepare(h, h->cur);
                if (st) return st;
                CVX_HIP(hipMemcpyAsync(h->d_lm, h->h_lm, sizeof(LmDev), hipMemcpyHostToDevice, h->stream));
            }
            st = enqueue_step(d, spec, seq);
            if (st) return st;
        }
        spec_inflight = false;
        if (spec && snap_tr.it + 1 < snap_opt.max_iterations) {
            spec_seq = ++h->step_seq;
            st = enqueue_step(d, true, spec_seq);      // (d is not read: the record's damping is)
            if (st) return st;
            spec_inflight = true;
        }
        st = wait_step(seq, spec ? (int)(seq & 1) : 0);
        if (st) return st;
        if (spec) {
            h->rt_valid[1 - h->cur] = true;             // k_pose_prepare_cand made the candidate's tables
            spec_runs = rec[15] != 0.0;
        }
        return CVX_OK;
    }
};


int alloc_exchange(cvx_calib *h) {
    int st = CVX_OK;
    if (h->d_E && h->d_E != h->d_E_res) (void)hipFree(h->d_E);
    if (h->h_E && h->h_E != h->h_E_res) (void)hipHostFree(h->h_E);
    h->d_E = nullptr; h->h_E = nullptr;
    h->E_scal = (size_t)h->ns * h->ns + 3 * (size_t)h->ns;
    h->E_count = h->E_scal + 2 + 2 * (size_t)h->nranks;
    if (h->d_E_res && h->h_E_res && h->E_count <= h->E_res_count) {
        h->d_E = h->d_E_res; h->h_E = h->h_E_res;       // (the usual case: no allocation)
    } else {
        st = dev_alloc(&h->d_E, h->E_count);
        if (st) return st;
        CVX_HIP(hipHostMalloc((void **)&h->h_E, h->E_count * sizeof(double), hipHostMallocDefault));
    }
    CVX_HIP(hipMemsetAsync(h->d_E, 0, h->E_count * sizeof(double), h->stream));
    if (h->d_X) { (void)hipFree(h->d_X); h->d_X = nullptr; }
    h->X_tail = (size_t)h->ns * (h->ns + 1) / 2;
    h->X_count = h->X_tail + (h->E_count - (size_t)h->ns * h->ns);
    if (h->nranks > 1) {
        st = dev_alloc(&h->d_X, h->X_count);
        if (st) return st;
        CVX_HIP(hipMemsetAsync(h->d_X, 0, h->X_count * sizeof(double), h->stream));
    }
    return CVX_OK;
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" int cvx_calib_create(const cvx_calib_desc *d, cvx_calib **out) {
    if (!d || !out) return CVX_ERR_INVALID_ARG;
    *out = nullptr;
    PhaseClock pc;
    if (d->n_cameras < 1 || d->n_cameras > CVX_MAX_CAMERAS || d->n_views < 0 || d->n_points < 1 || d->n_obs < 0)
        return CVX_ERR_INVALID_ARG;
    const bool obs_f32 = !d->obs_x && !d->obs_y && d->obs_xf && d->obs_yf;
    if (!d->cam_of_view || !d->view_off || (d->n_obs > 0 && (!d->point_id || (!obs_f32 && (!d->obs_x || !d->obs_y)))))
        return CVX_ERR_INVALID_ARG;
    if (d->n_points > CVX_MAX_POINTS) return CVX_ERR_UNSUPPORTED;
    const int V = d->n_views, P = d->n_points, C = d->n_cameras;
    if (d->view_off[0] != 0 || d->view_off[V] != d->n_obs) return CVX_ERR_INVALID_ARG;
    // the views here; the observation lists (point ids in range, no point twice in a view) on the device, below
    int maxn = 0;
    for (int v = 0; v < V; ++v) {
        if (d->cam_of_view[v] < 0 || d->cam_of_view[v] >= C) return CVX_ERR_INVALID_ARG;
        const int n = d->view_off[v + 1] - d->view_off[v];
        if (n < 0) return CVX_ERR_INVALID_ARG;
        maxn = std::max(maxn, n);
    }
    if (maxn > CVX_MAX_MARKERS_PER_VIEW) return CVX_ERR_UNSUPPORTED;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        cvx::set_last_error("no HIP device visible");
        return CVX_ERR_NO_DEVICE;
    }
    if (d->device < 0 || d->device >= ndev) return CVX_ERR_NO_DEVICE;
    CVX_HIP(hipSetDevice(d->device));

    pc.mark("validation");
    cvx_calib *h = new cvx_calib;
    std::unique_ptr<cvx_calib> guard(h);
    h->C = C; h->V = V; h->P = P; h->N = d->n_obs;
    h->fix_k3 = d->fix_k3 ? 1 : 0; h->fix_intr = d->fix_intrinsics ? 1 : 0; h->fix_pts = d->fix_points ? 1 : 0;
    h->device = d->device;
    if (d->stream) h->stream = (hipStream_t)d->stream;
    else {
        h->own_stream = true;
        const int sst = cvx::own_stream_begin(d->device, &h->stream);
        if (sst) return sst;
    }
    pc.mark("  stream");
    h->cam_free = h->fix_intr ? 0 : 1;
    h->pts_free = h->fix_pts ? 0 : 1;
    h->ncf = h->fix_intr ? 0 : (h->fix_k3 ? 8 : 9);
    h->ns = C * h->ncf + (h->pts_free ? 3 * P : 0);
    h->camcol0 = h->pts_free ? 3 * P : 0;
    h->ycol = h->camcol0 + (h->cam_free ? 9 : 0);
    h->ZC = ((h->ycol + 1 + 15) / 16) * 16;
    h->nb = h->ZC / 16;
    h->nblk = h->nb * (h->nb + 1) / 2;
    h->parts = (h->nblk + K3_CAP - 1) / K3_CAP;
    h->ZCp = h->ZC + ((h->ZC % 32 == 16) ? 0 : 16);        // LDS row pitch = 16 mod 32 doubles
    h->KB = 16;
    while (h->KB > 4 && (h->KB * h->ZC > 6144 || 2 * (size_t)h->KB * h->ZCp * sizeof(double) > 128 * 1024)) h->KB -= 4;
    h->k3_lds = 2 * (size_t)h->KB * h->ZCp * sizeof(double);
    if (h->KB * h->ZC > 6144 || h->k3_lds > 160 * 1024) return CVX_ERR_UNSUPPORTED;
    // the register-tile form of the contraction (v_mfma_f64_4x4x4_4b) wherever its fixed LDS pitch holds a row
    std::vector<K4Task> k4tasks;
    if (h->ZC <= K4_PITCH) {
        h->parts = k4_build_tasks(h->nb, k4tasks);
        h->k3_lds = K4_BUFS * (size_t)K4_KB * K4_PITCH * sizeof(double) + 512;   // + slack: groups past nb are read, not used
    }
    h->T = maxn <= 64 ? 64 : maxn <= 128 ? 128 : maxn <= 256 ? 256 : 512;
    const int W = h->T / 64;
    h->h_cam_of_view.assign(d->cam_of_view, d->cam_of_view + V);
    h->h_view_off.assign(d->view_off, d->view_off + V + 1);
    if (d->n_obs) h->h_point_id.assign(d->point_id, d->point_id + d->n_obs);
    pc.mark("  host copies of the lists");
    if (d->prepare_inner_iterations) start_inner_ahead(h);      // beside the rest of this function
    pc.mark("  inner preparation started");

    // groups (K2 workgroups) and chunks (K3 row ranges): contiguous views of one camera
    std::vector<int> grp_v0, grp_cam, chk_row0, chk_cam;
    {
        const int vpg = std::max(1, (V + 1023) / 1024);
        int v = 0;
        while (v < V) {
            const int c = d->cam_of_view[v];
            int e = v;
            while (e < V && e - v < vpg && d->cam_of_view[e] == c) ++e;
            grp_v0.push_back(v); grp_cam.push_back(c);
            v = e;
        }
        grp_v0.push_back(V);
        h->n_groups = (int)grp_cam.size();
        // (whether the groups are aligned -- thread tid meets the same point in every view of its group -- is found on
        // the device, below, and with it the normal-equation kernel's LDS need)
        const int target_chunks = std::max(1, 256 / std::max(1, h->parts));
        const int gpc = std::max(1, (h->n_groups + target_chunks - 1) / target_chunks);
        int g = 0;
        while (g < h->n_groups) {
            const int c = grp_cam[g];
            int e = g;
            while (e < h->n_groups && e - g < gpc && grp_cam[e] == c) ++e;
            chk_row0.push_back(6 * grp_v0[g]); chk_cam.push_back(c);
            g = e;
        }
        chk_row0.push_back(6 * V);
        h->n_chunks = (int)chk_cam.size();
    }
    std::vector<int> glist(h->n_groups), glist_off(C + 1, 0);
    for (int g = 0; g < h->n_groups; ++g) glist_off[grp_cam[g] + 1]++;
    for (int c = 0; c < C; ++c) glist_off[c + 1] += glist_off[c];
    {
        std::vector<int> pos(glist_off.begin(), glist_off.end() - 1);
        for (int g = 0; g < h->n_groups; ++g) glist[pos[grp_cam[g]]++] = g;
    }
    hipStream_t s = h->stream;
    int st;
    pc.mark("host tables");
#define CVX_TRY(x) do { st = (x); if (st) return st; } while (0)
    {
        // the arena: the sum of what follows, each buffer rounded up to 256 bytes (dalloc), plus slack
        auto up32 = [](size_t n) { return (std::max<size_t>(n, 1) + 31) / 32 * 32; };
        const size_t Nn = (size_t)h->N, Vn = (size_t)V, G = (size_t)h->n_groups, Q = (size_t)h->n_chunks;
        const size_t npad_ = (size_t)((h->ns + 1 + cvxs::TB - 1) / cvxs::TB) * cvxs::TB;
        const size_t ints = (Vn + 1) + Vn + 2 * Nn + (G + 1) + (Q + 1) + Q + G + 2 * ((size_t)C + 1) + 1;
        const size_t doubles = 2 * Nn + 2 * (up32(9 * (size_t)C) + up32(3 * (size_t)P) + up32(6 * Vn) + up32(Vn * RT_STRIDE)) +
                               (6 * Vn + 8) * h->ZC + (21 + 6 * 3 + 6 + 3 + 1) * Vn + G * P * PT_ACC + G * 256 + G * 4 +
                               Q * h->nblk * 256 + (size_t)C * P * PT_ACC + (size_t)C * h->nblk * 256 + (size_t)C * 256 +
                               (size_t)C * h->ZC + 2048 + 16 + 2 * npad_ * npad_ + 3 * (size_t)std::max(1, h->ns) +
                               ((size_t)h->ns * h->ns + 3 * (size_t)h->ns + 2 + 2 * 64);
        h->arena_cap = ints * sizeof(int) + doubles * sizeof(double) + k4tasks.size() * sizeof(K4Task) + sizeof(LmDev) + 64 * 256 + 65536;
        if (d->prepare_inner_iterations) {
            // room for the inner iterations' two blocks (ensure_inner: point-major copies of the observations, block state)
            const size_t NB = (size_t)C + P + Vn, slices = (size_t)std::min<int64_t>(64, std::max<int64_t>(1, h->N / P / 512));
            h->arena_cap += sizeof(int) * (((size_t)P + 1) + 3 * Nn + 5 * NB + 6 * 64) +
                            sizeof(double) * (8 + NB * (cvxi::IB_STRIDE + cvxi::IB_STATE + 10) + (G + 1 + C + P) * cvxi::IB_STRIDE +
                                              (size_t)P * slices * cvxi::PT_SLOT + 2 * Nn + 10 * 32) + 2 * 256;
        }
        size_t kept = 0;
        h->arena = arena_take(h->device, h->arena_cap, &kept);
        if (h->arena) h->arena_cap = kept;
        else if (hipMalloc((void **)&h->arena, h->arena_cap) != hipSuccess) {     // no arena: every buffer on its own
            (void)hipGetLastError();
            h->arena = nullptr; h->arena_cap = 0;
        }
        // CVX_POISON_ARENA=1 (development): every byte 0xFF -- NaN for doubles, -1 for integers -- so that a buffer read
        // before it is written shows in the tests instead of passing on memory that happens to come back zeroed
        if (h->arena && std::getenv("CVX_POISON_ARENA")) CVX_HIP(hipMemsetAsync(h->arena, 0xFF, h->arena_cap, s));
    }
    pc.mark("  arena");
    CVX_TRY(h->dupload(&h->d_view_off, d->view_off, (size_t)V + 1, s));
    CVX_TRY(h->dupload(&h->d_cam_of_view, d->cam_of_view, (size_t)V, s));
    CVX_TRY(h->dupload(&h->d_point_id, d->point_id, (size_t)h->N, s));
    CVX_TRY(h->dalloc(&h->d_view_id, (size_t)h->N));       // every observation's view: made on the device from the offsets
    if (V) hipLaunchKernelGGL(k_view_ids, dim3((unsigned)V), dim3(64), 0, s, h->d_view_off, h->d_view_id);
    if (!obs_f32) {
        CVX_TRY(h->dupload(&h->d_obs_x, d->obs_x, (size_t)h->N, s));
        CVX_TRY(h->dupload(&h->d_obs_y, d->obs_y, (size_t)h->N, s));
    } else {
        CVX_TRY(h->dalloc(&h->d_obs_x, (size_t)h->N));
        CVX_TRY(h->dalloc(&h->d_obs_y, (size_t)h->N));
    }
    CVX_TRY(h->dupload(&h->d_grp_v0, grp_v0.data(), grp_v0.size(), s));
    int *&d_flags = h->d_flags;
    CVX_TRY(h->dalloc(&d_flags, 4));
    CVX_HIP(hipMemsetAsync(d_flags, 0, 4 * sizeof(int), s));
    if (V) hipLaunchKernelGGL(k_check_views, dim3((unsigned)V), dim3(64), 0, s, h->d_view_off, h->d_point_id, P, d_flags);
    if (h->n_groups) hipLaunchKernelGGL(k_check_aligned, dim3((unsigned)h->n_groups), dim3(128), 0, s, h->d_view_off, h->d_point_id, h->d_grp_v0, d_flags);
    CVX_TRY(h->dupload(&h->d_chk_row0, chk_row0.data(), chk_row0.size(), s));
    if (!k4tasks.empty()) CVX_TRY(h->dupload(&h->d_k4tasks, k4tasks.data(), k4tasks.size(), s));
    {
        std::vector<int> clist(h->n_chunks), clist_off(C + 1, 0);
        for (int q = 0; q < h->n_chunks; ++q) clist_off[chk_cam[q] + 1]++;
        for (int c = 0; c < C; ++c) clist_off[c + 1] += clist_off[c];
        std::vector<int> pos(clist_off.begin(), clist_off.end() - 1);
        for (int q = 0; q < h->n_chunks; ++q) clist[pos[chk_cam[q]]++] = q;
        CVX_TRY(h->dupload(&h->d_clist, clist.data(), clist.size(), s));
        CVX_TRY(h->dupload(&h->d_clist_off, clist_off.data(), clist_off.size(), s));
    }
    CVX_TRY(h->dupload(&h->d_glist, glist.data(), glist.size(), s));
    CVX_TRY(h->dupload(&h->d_glist_off, glist_off.data(), glist_off.size(), s));
    pc.mark("uploads");
    {
        auto up32 = [](size_t n) { return (long long)((std::max<size_t>(n, 1) + 31) / 32 * 32); };
        h->intr_step = up32(9 * (size_t)C); h->points_step = up32(3 * (size_t)P);
        h->poses_step = up32(6 * (size_t)V); h->rt_step = up32((size_t)V * RT_STRIDE);
        CVX_TRY(h->dalloc(&h->d_intr[0], 2 * (size_t)h->intr_step));
        CVX_TRY(h->dalloc(&h->d_points[0], 2 * (size_t)h->points_step));
        CVX_TRY(h->dalloc(&h->d_poses[0], 2 * (size_t)h->poses_step));
        CVX_TRY(h->dalloc(&h->d_rt[0], 2 * (size_t)h->rt_step));
        h->d_intr[1] = h->d_intr[0] + h->intr_step; h->d_points[1] = h->d_points[0] + h->points_step;
        h->d_poses[1] = h->d_poses[0] + h->poses_step; h->d_rt[1] = h->d_rt[0] + h->rt_step;
        CVX_HIP(hipMemsetAsync(h->d_intr[0], 0, 2 * (size_t)h->intr_step * sizeof(double), s));
        CVX_HIP(hipMemsetAsync(h->d_points[0], 0, 2 * (size_t)h->points_step * sizeof(double), s));
        CVX_HIP(hipMemsetAsync(h->d_poses[0], 0, 2 * (size_t)h->poses_step * sizeof(double), s));
        CVX_TRY(h->dalloc(&h->d_lm, 1));
        CVX_HIP(hipMemsetAsync(h->d_lm, 0, sizeof(LmDev), s));
    }
    const size_t zn = ((size_t)6 * V + 8) * h->ZC;
    CVX_TRY(h->dalloc(&h->d_Z, zn));
    if (obs_f32 && h->N) {
        // float observations: staged in memory that is idle until the first linearisation (Z), widened on the device
        float *stage = reinterpret_cast<float *>(h->d_Z);
        float *own = nullptr;
        if (2 * (size_t)h->N * sizeof(float) > zn * sizeof(double)) { CVX_HIP(hipMalloc((void **)&own, 2 * (size_t)h->N * sizeof(float))); stage = own; }
        hipError_t e = hipMemcpyAsync(stage, d->obs_xf, (size_t)h->N * sizeof(float), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipMemcpyAsync(stage + h->N, d->obs_yf, (size_t)h->N * sizeof(float), hipMemcpyHostToDevice, s);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_widen, dim3((unsigned)((h->N + 255) / 256)), dim3(256), 0, s, stage, h->d_obs_x, h->d_obs_y, (long long)h->N);
            e = hipGetLastError();
        }
        if (own) { (void)hipStreamSynchronize(s); (void)hipFree(own); }
        CVX_HIP(e);
    }
    pc.mark("  parameters, Z, float observations");
    CVX_HIP(hipMemsetAsync(h->d_Z, 0, zn * sizeof(double), s));
    CVX_TRY(h->dalloc(&h->d_Linv, 21 * (size_t)V));
    CVX_TRY(h->dalloc(&h->d_Dv, 6 * (size_t)V));
    CVX_TRY(h->dalloc(&h->d_gv, 6 * (size_t)V));
    CVX_TRY(h->dalloc(&h->d_s2pose, 6 * (size_t)V));
    if (V) hipLaunchKernelGGL(k_fill, dim3((unsigned)((6LL * V + 255) / 256)), dim3(256), 0, s, h->d_s2pose, 6LL * V, 1.0);
    CVX_TRY(h->dalloc(&h->d_slab_pt, (size_t)h->n_groups * P * PT_ACC));
    CVX_HIP(hipMemsetAsync(h->d_slab_pt, 0, std::max<size_t>(1, (size_t)h->n_groups * P * PT_ACC) * sizeof(double), s));
    CVX_TRY(h->dalloc(&h->d_slab_cam, (size_t)h->n_groups * 256));
    CVX_TRY(h->dalloc(&h->d_grp_scal, (size_t)h->n_groups * 4));
    CVX_TRY(h->dalloc(&h->d_slab_tile, (size_t)h->n_chunks * h->nblk * 256));
    CVX_TRY(h->dalloc(&h->d_Tpt, (size_t)C * P * PT_ACC));
    CVX_HIP(hipMemsetAsync(h->d_Tpt, 0, (size_t)C * P * PT_ACC * sizeof(double), s));
    CVX_TRY(h->dalloc(&h->d_Tt, (size_t)C * h->nblk * 256));
    CVX_TRY(h->dalloc(&h->d_Gcam, (size_t)C * 256));
    pc.mark("allocations 1");
    {
        // one pinned, host-mapped block: the step's two records (polled) | the speculated steps' staging record | the
        // inner iterations' counters | the embedded step | the parameter staging of set_params / get_params | E's staging
        // copy (its size depends on the number of ranks: room for 64, beyond that alloc_exchange makes its own)
        const size_t n_lm = (sizeof(LmDev) + 7) / 8, n_emb = (size_t)C * h->ZC, n_par = 9 * (size_t)C + 3 * (size_t)P + 6 * (size_t)V;
        auto up8 = [](size_t n) { return (n + 7) / 8 * 8; };
        h->E_res_count = (size_t)h->ns * h->ns + 3 * (size_t)h->ns + 2 + 2 * 64;
        const size_t total = 64 + up8(n_lm) + 8 + up8(n_emb) + up8(std::max<size_t>(n_par, 1)) + up8(h->E_res_count);
        CVX_HIP(hipHostMalloc((void **)&h->h_E2, total * sizeof(double), hipHostMallocMapped));
        double *q = h->h_E2 + 64;
        h->h_lm = reinterpret_cast<LmDev *>(q); q += up8(n_lm);
        h->h_ib_active = q; q += 8;
        std::memset(h->h_ib_active, 0, 8 * sizeof(double));     // [5] is a polled sequence number
        h->h_ds_emb = q; q += up8(n_emb);
        h->h_params = q; q += up8(std::max<size_t>(n_par, 1));
        h->h_E_res = q;
        CVX_TRY(h->dalloc(&h->d_E_res, h->E_res_count));
    }
    pc.mark("  pinned block");
    CVX_TRY(alloc_exchange(h));
    pc.mark("alloc_exchange");
    CVX_TRY(h->dalloc(&h->d_ds_emb, (size_t)C * h->ZC));
    CVX_TRY(h->dalloc(&h->d_dposes, 6 * (size_t)V));
    CVX_TRY(h->dalloc(&h->d_out_view, 4 * (size_t)std::max(V, 1)));       // {cost, model terms} per view (step_kernels.h)
    h->n_cost_blocks = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (h->N + 255) / 256));
    CVX_TRY(h->dalloc(&h->d_cost_partial, (size_t)h->n_cost_blocks));
    CVX_HIP(hipEventCreateWithFlags(&h->ev_step, hipEventDisableTiming));
    CVX_HIP(hipEventCreateWithFlags(&h->ev_params, hipEventDisableTiming));
    CVX_TRY(h->dalloc(&h->d_E2, 16));
    CVX_HIP(hipMemsetAsync(h->d_E2, 0, 16 * sizeof(double), s));
    // two records of 32 doubles (a speculated step may publish while the host still reads the one before)
    std::memset(h->h_E2, 0, 64 * sizeof(double));
    if (hipHostGetDevicePointer((void **)&h->d_E2_map, h->h_E2, 0) != hipSuccess) h->d_E2_map = nullptr;
    // damped reduced solve: (ns + 1) rows (the right-hand side rides as row ns), padded to 32-tiles
    h->nT = (h->ns + 1 + cvxs::TB - 1) / cvxs::TB;
    h->npad = h->nT * cvxs::TB;
    // + npad extra rows below: the identity block whose elimination yields the solution (k_chol_step); never initialised
    CVX_TRY(h->dalloc(&h->d_A, 2 * (size_t)h->npad * h->npad));
    CVX_TRY(h->dalloc(&h->d_s2sh, (size_t)std::max(1, h->ns)));
    CVX_TRY(h->dalloc(&h->d_Ds, (size_t)std::max(1, h->ns)));
    CVX_HIP(hipMemsetAsync(h->d_Ds, 0, (size_t)std::max(1, h->ns) * sizeof(double), s));
    CVX_TRY(h->dalloc(&h->d_delta_s, (size_t)std::max(1, h->ns)));
    CVX_TRY(h->dalloc(&h->d_fail, 1));
    h->fin_lds = sizeof(double) * cvxs::finish_lds_doubles(h->ns);
#undef CVX_TRY
    hipLaunchKernelGGL(k_fill, dim3((std::max(1, h->ns) + 255) / 256), dim3(256), 0, s, h->d_s2sh, (long long)std::max(1, h->ns), 1.0);
    for (int i = 0; i < PH_COUNT; ++i) {
        CVX_HIP(hipEventCreate(&h->ev[i][0]));
        CVX_HIP(hipEventCreate(&h->ev[i][1]));
    }
    h->ev_made = true;
    pc.mark("allocations 2, events");
    {
        int flags[4] = {0, 0, 0, 0};
        CVX_HIP(hipMemcpyAsync(flags, d_flags, sizeof flags, hipMemcpyDeviceToHost, s));
        CVX_HIP(hipStreamSynchronize(s));
        if (flags[0]) return CVX_ERR_INVALID_ARG;
        if (flags[1]) return CVX_ERR_DUPLICATE_MARKER;
        h->aligned = flags[2] == 0;
        h->k2_lds = sizeof(double) * ((h->aligned ? 0 : (size_t)P * PT_ACC) + (size_t)W * 64 * AROW + (size_t)W * 256 + 256);
        if (h->k2_lds > 160 * 1024) return CVX_ERR_UNSUPPORTED;
    }
    if (pc.on) std::fprintf(stderr, "[cvx prof] arena: %.1f of %.1f MB used, %d buffers outside it\n", h->arena_used / 1048576.0, h->arena_cap / 1048576.0, h->arena_fallbacks);
    CVX_HIP(hipStreamSynchronize(s));
    pc.mark("stream synchronize");
    *out = guard.release();
    return CVX_OK;
}

extern "C" int cvx_release_cached_memory(void) {
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (int dev = 0; dev < 16; ++dev) {
        char *p = nullptr;
        {
            std::lock_guard<std::mutex> lock(g_arena_cache.mu);
            std::swap(p, g_arena_cache.p[dev]);
            g_arena_cache.cap[dev] = 0;
        }
        if (p) { (void)hipSetDevice(dev); (void)hipFree(p); }
    }
    cvx::stream_cache_release();
    if (have) (void)hipSetDevice(cur);
    return CVX_OK;
}

extern "C" void cvx_calib_destroy(cvx_calib *h) {
    PhaseClock pc;
    delete h;
    pc.mark("destroy");
}

extern "C" int cvx_calib_set_params(cvx_calib *h, const double *intr, const double *poses, const double *points) {
    if (!h || !intr || !points || (h->V > 0 && !poses)) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    const int c = h->cur;
    h->rt_valid[c] = false;
    // through the handle's pinned staging block: three asynchronous copies (copies from pageable memory wait one by one)
    // and NO wait -- whatever uses the parameters next is on the same stream; only the next writer of the staging block
    // has to know that the copies are through, and asks the event
    double *sp = h->h_params;
    const size_t ni = 9 * (size_t)h->C, np = 3 * (size_t)h->P, nv = 6 * (size_t)h->V;
    if (h->params_pending) { CVX_HIP(hipEventSynchronize(h->ev_params)); h->params_pending = false; }
    std::memcpy(sp, intr, sizeof(double) * ni);
    std::memcpy(sp + ni, points, sizeof(double) * np);
    if (nv) std::memcpy(sp + ni + np, poses, sizeof(double) * nv);
    CVX_HIP(hipMemcpyAsync(h->d_intr[c], sp, sizeof(double) * ni, hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipMemcpyAsync(h->d_points[c], sp + ni, sizeof(double) * np, hipMemcpyHostToDevice, h->stream));
    if (nv) CVX_HIP(hipMemcpyAsync(h->d_poses[c], sp + ni + np, sizeof(double) * nv, hipMemcpyHostToDevice, h->stream));
    CVX_HIP(hipEventRecord(h->ev_params, h->stream));
    h->params_pending = true;
    return CVX_OK;
}

extern "C" int cvx_calib_get_params(cvx_calib *h, double *intr, double *poses, double *points) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    const int c = h->cur;
    double *sp = h->h_params;
    const size_t ni = 9 * (size_t)h->C, np = 3 * (size_t)h->P, nv = 6 * (size_t)h->V;
    if (intr) CVX_HIP(hipMemcpyAsync(sp, h->d_intr[c], sizeof(double) * ni, hipMemcpyDeviceToHost, h->stream));
    if (points) CVX_HIP(hipMemcpyAsync(sp + ni, h->d_points[c], sizeof(double) * np, hipMemcpyDeviceToHost, h->stream));
    if (poses && nv) CVX_HIP(hipMemcpyAsync(sp + ni + np, h->d_poses[c], sizeof(double) * nv, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));       // (set_params' copies out of the block are ahead of these on the stream)
    h->params_pending = false;
    if (intr) std::memcpy(intr, sp, sizeof(double) * ni);
    if (points) std::memcpy(points, sp + ni, sizeof(double) * np);
    if (poses && nv) std::memcpy(poses, sp + ni + np, sizeof(double) * nv);
    return CVX_OK;
}

extern "C" int cvx_calib_num_shared(const cvx_calib *h) { return h ? h->ns : -1; }

extern "C" int cvx_calib_cost(cvx_calib *h, double *cost) {
    if (!h || !cost) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    return cost_of(h, h->cur, cost);
}

extern "C" int cvx_calib_inner_pass(cvx_calib *h, double *cost) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    int local = CVX_OK;
    int st = inner_pass(h, h->cur, &local);
    if (st) return st;
    if (h->nranks > 1 && h->xfn) {          // all ranks leave with the same status
        std::vector<double> v = {local ? 1.0 : 0.0};
        st = exchange_sum_host(h, v);
        if (st) return st;
        if (agreed_status(v[0])) return CVX_ERR_NUMERIC;
    } else if (local) {
        return local;
    }
    double c = 0;
    st = cost_of(h, h->cur, &c);
    if (cost) *cost = c;
    return st;
}

extern "C" int cvx_calib_evaluate_device(cvx_calib *h, void **device_buffer) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (!h->d_eval) {
        int st = dev_alloc(&h->d_eval, 38 * (size_t)h->N);
        if (st) return st;
    }
    int st = launch_pose_prepare(h, h->cur);
    if (st) return st;
    if (h->N > 0) {
        hipLaunchKernelGGL(k_eval, dim3((unsigned)((h->N + EVAL_THREADS - 1) / EVAL_THREADS)), dim3(EVAL_THREADS), 0, h->stream, h->d_view_id,
                           h->d_cam_of_view, h->d_point_id, h->d_obs_x, h->d_obs_y, h->d_intr[h->cur],
                           h->d_points[h->cur], h->d_rt[h->cur], (long long)h->N, h->d_eval);
        CVX_HIP(hipGetLastError());
    }
    if (device_buffer) *device_buffer = h->d_eval;
    return CVX_OK;
}

extern "C" int cvx_calib_evaluate(cvx_calib *h, double *cost, double *r, double *Jc, double *Jp, double *Jx) {
    if (!h) return CVX_ERR_INVALID_ARG;
    int st = cvx_calib_evaluate_device(h, nullptr);
    if (st) return st;
    const size_t N = (size_t)h->N;
    std::vector<double> buf(38 * N);
    if (N) CVX_HIP(hipMemcpyAsync(buf.data(), h->d_eval, sizeof(double) * 38 * N, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    long double s = 0;      // 10^7 terms in one running sum: the extended accumulator keeps it at 1e-16
    for (size_t o = 0; o < N; ++o) {
        const double r0 = buf[o], r1 = buf[N + o];
        s += (long double)(r0 * r0 + r1 * r1);
        if (r) { r[2 * o] = r0; r[2 * o + 1] = r1; }
        if (Jc) for (int i = 0; i < 18; ++i) Jc[18 * o + i] = buf[(2 + i) * N + o];
        if (Jp) for (int i = 0; i < 12; ++i) Jp[12 * o + i] = buf[(20 + i) * N + o];
        if (Jx) for (int i = 0; i < 6; ++i) Jx[6 * o + i] = buf[(32 + i) * N + o];
    }
    if (cost) *cost = (double)(0.5L * s);
    return CVX_OK;
}

extern "C" int cvx_calib_init_scaling(cvx_calib *h, int enable) {
    if (!h) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    CalibEvaluator ev(h);
    return ev.init(enable != 0).status;
}

extern "C" int cvx_calib_linear_step(cvx_calib *h, const cvx_damping *d, double *S, double *gred,
                                     double *delta_shared, double *delta_poses, double *cost,
                                     double *cand_cost, double *model_change) {
    if (!h || !d) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    CalibEvaluator ev(h);
    ev.keep_dposes = true;
    ev.want_S = S;
    cvx::StepInfo s = ev.try_step(*d, -1.0, false);
    if (s.status) return s.status;
    const int ns = h->ns;
    if (cost) *cost = s.cost;
    if (gred && ns) CVX_HIP(hipMemcpyAsync(gred, h->d_E + (size_t)ns * ns, sizeof(double) * ns, hipMemcpyDeviceToHost, h->stream));
    if (S)      // the damped matrix as it entered the factorisation (lower triangle mirrored)
        for (int i = 0; i < ns; ++i)
            for (int j = 0; j < ns; ++j) S[(size_t)i * ns + j] = h->h_A[(size_t)std::max(i, j) * h->npad + std::min(i, j)];
    if (delta_shared && ns)
        CVX_HIP(hipMemcpyAsync(delta_shared, h->d_delta_s, sizeof(double) * ns, hipMemcpyDeviceToHost, h->stream));
    if (delta_poses && h->V)
        CVX_HIP(hipMemcpyAsync(delta_poses, h->d_dposes, sizeof(double) * 6 * (size_t)h->V, hipMemcpyDeviceToHost, h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    if (!s.ok) return CVX_ERR_NUMERIC;
    if (cand_cost) *cand_cost = s.cand_cost;
    if (model_change) *model_change = s.model_change;
    return CVX_OK;
}

extern "C" int cvx_calib_time_schur(cvx_calib *h, int reps, double *ms_per_launch) {
    if (!h || reps < 1 || !ms_per_launch) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    *ms_per_launch = 0.0;
    if (h->ns <= 0 || h->V <= 0) return CVX_OK;
    const cvx_damping d = {1e-4, 0.0, 1e-6, 1e32};
    int st = linearize(h, d, false, false);         // makes Z current
    if (st) return st;
    struct Events {
        hipEvent_t e[2] = {nullptr, nullptr};
        ~Events() { for (hipEvent_t x : e) if (x) (void)hipEventDestroy(x); }
    } ev;
    CVX_HIP(hipEventCreate(&ev.e[0]));
    CVX_HIP(hipEventCreate(&ev.e[1]));
    st = launch_schur(h);                            // one untimed launch
    CVX_HIP(hipEventRecord(ev.e[0], h->stream));
    for (int i = 0; i < reps && !st; ++i) st = launch_schur(h);
    CVX_HIP(hipEventRecord(ev.e[1], h->stream));
    CVX_HIP(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    CVX_HIP(hipEventElapsedTime(&ms, ev.e[0], ev.e[1]));
    if (st) return st;
    *ms_per_launch = (double)ms / reps;
    return CVX_OK;
}

extern "C" int cvx_calib_set_exchange(cvx_calib *h, cvx_allreduce_fn fn, void *ctx, int rank, int nranks,
                                      int device_buffers) {
    if (!h || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn)) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    h->xfn = fn; h->xctx = ctx; h->rank = rank; h->nranks = nranks; h->xdev = device_buffers ? 1 : 0;
    h->inner_ready = false;     // the inner iterations' ordering is a function of the whole problem
    int st = alloc_exchange(h);
    if (st) return st;
    CVX_HIP(hipStreamSynchronize(h->stream));
    return CVX_OK;
}

extern "C" int cvx_calib_solve(cvx_calib *h, const cvx_lm_options *opt, cvx_lm_summary *summary) {
    if (!h || !opt || !summary) return CVX_ERR_INVALID_ARG;
    CVX_HIP(hipSetDevice(h->device));
    if (h->V == 0) {
        // no residual block at all: the solve of an empty problem, as for views without markers (on a sharded handle
        // this rank could not take part in the exchanges: refused)
        if (h->nranks > 1) { cvx::set_last_error("a rank of a sharded solve holds no views"); return CVX_ERR_INVALID_ARG; }
        std::memset(summary, 0, sizeof *summary);
        summary->termination = 1;
        return CVX_OK;
    }
    // the host half of the inner iterations' first use runs beside the initial linearisation and the first step
    if (opt->inner_iterations && opt->strategy == 0) start_inner_ahead(h);
    CalibEvaluator ev(h);
    return cvx::lm_minimize(ev, *opt, *summary);
}

extern "C" int cvx_calib_prepare_inner_iterations(cvx_calib *h) {
    if (!h) return CVX_ERR_INVALID_ARG;
    start_inner_ahead(h);
    return CVX_OK;
}

extern "C" double cvx_calib_refine_return(double total_cost, int64_t n_obs_total) {
    return std::sqrt(total_cost / (2.0 * (double)n_obs_total));
}

extern "C" int cvx_calib_enable_timing(cvx_calib *h, int enable) {
    if (!h) return CVX_ERR_INVALID_ARG;
    h->timing = enable != 0;
    for (int i = 0; i < PH_COUNT; ++i) { h->ph_ms[i] = 0; h->ph_n[i] = 0; h->ph_used[i] = false; }
    return CVX_OK;
}

extern "C" int cvx_calib_phase_times(cvx_calib *h, double ms[16]) {
    if (!h || !ms) return CVX_ERR_INVALID_ARG;
    for (int i = 0; i < PH_COUNT; ++i) { ms[i] = h->ph_ms[i]; ms[8 + i] = (double)h->ph_n[i]; }
    for (int i = PH_COUNT; i < 8; ++i) { ms[i] = 0; ms[8 + i] = 0; }
    return CVX_OK;
}

// geometry the bench needs to state algorithmic work: [0] ZC, [1] ns, [2] n_groups, [3] n_chunks,
// [4] upper-triangular tile pairs, [5] K2 threads per block, [6] Z rows
extern "C" int cvx_calib_speculation_counts(const cvx_calib *h, int64_t out[2]) {
    if (!h || !out) return CVX_ERR_INVALID_ARG;
    out[0] = h->n_spec_used; out[1] = h->n_spec_dropped;
    return CVX_OK;
}

extern "C" int cvx_calib_geometry(const cvx_calib *h, int64_t out[8]) {
    if (!h || !out) return CVX_ERR_INVALID_ARG;
    out[0] = h->ZC; out[1] = h->ns; out[2] = h->n_groups; out[3] = h->n_chunks; out[4] = h->nblk;
    out[5] = h->T; out[6] = 6LL * h->V; out[7] = h->ns > 0 ? h->nT : 0;
    return CVX_OK;
}

#if defined(K2_PROF) || defined(K2_WALL)
extern "C" int cvx_calib_debug_group_scalars(cvx_calib *h, double *out) {
    CVX_HIP(hipMemcpy(out, h->d_grp_scal, sizeof(double) * 4 * h->n_groups, hipMemcpyDeviceToHost));
    return h->n_groups;
}
#endif
