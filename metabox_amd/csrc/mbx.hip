// mbx.hip — libmbx.so: kernels' launch code and the C-ABI of include/mbx.h.
// Build: make -C metabox_amd/csrc  (six translation units: this file, mbx_run_rlepso*.hip, mbx_run_lde.hip); one file: hipcc ... -DMBX_SINGLE_TU -shared mbx.hip -o libmbx.so
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <memory>
#include <string>
#include <vector>

#include "mbx_device.hpp"
#include "mbx_rlepso.hpp"
#include "mbx_lde.hpp"
#ifndef MBX_LDE100_STEP_THREADS
#define MBX_LDE100_STEP_THREADS 512
#endif
#ifndef MBX_LDE50_STEP_THREADS
#define MBX_LDE50_STEP_THREADS 256     // workgroup size of k_lde_step<., 50, 30>: 38.8 KB of LDS let FOUR 256-thread workgroups share a CU (round 3, all 30 noisy
                                       // functions, 16 384 instances, one box: 128 / 256 / 384 / 512 / 1024 threads -> 658 / 466 / 662 / 483 / 849 us per generation)
#endif
#include "mbx_ddqn.hpp"
#include "mbx_rs.hpp"
#include "mbx_policy.hpp"
#include "mbx_lstm_policy.hpp"
#include "mbx_lde_run.hpp"
#include "mbx_qnet.hpp"
#include "mbx_rlpso.hpp"
#include "mbx_gleet.hpp"
#include "mbx_qlpso.hpp"
#include "mbx_gleet_policy.hpp"
#include "mbx_classic.hpp"
// k_rlepso_run / k_lde_run are compiled in translation units of their own (mbx_run_rlepso.hip, mbx_run_lde.hip) and only declared here;
// -DMBX_SINGLE_TU (instrumented builds: the phase counters are a __device__ array, one copy per translation unit) instantiates them in this file instead
#ifndef MBX_SINGLE_TU
#define MBX_RUN_KERNELS_EXTERN
#endif
#include "mbx_run_kernels.hpp"

using namespace mbx;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;

static int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) return fail(MBX_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)

// ------------------------------------------------------------------------------------------------ handles
struct mbx_suite {
    int n = 0, dim = 0;
    double* d_pool = nullptr;
    DevProblem* d_problems = nullptr;
    std::vector<DevProblem> h_problems;       // device pointers inside
    std::vector<double> optimum;
};

struct mbx_batch {
    mbx_suite* suite = nullptr;
    mbx_algo_cfg cfg{};
    int B = 0;
    int32_t* d_problem_idx = nullptr;
    std::vector<int32_t> h_problem_idx;    // host copy (mbx_debug_write_state validates an injected block against its problem's box)
    uint64_t* d_seeds = nullptr;
    double* d_state = nullptr;
    int32_t* d_order = nullptr;
    double* d_pci = nullptr;
    unsigned long long* d_clk = nullptr;   // mbx_debug_clock_slots: the caller's slot block for the NEXT resident RLEPSO launches (not owned)
    double* d_scratch = nullptr; // [B] per-generation rewards of mbx_rlepso_rollout's host-loop route (RLEPSO batches; allocated by mbx_batch_create)
    float* d_lstm_pack = nullptr;          // k-blocked copy of the PolicyNet weights for k_lde_run (k_lde_repack at every mbx_lde_rollout call)
    bool fdr_fast = false;                 // RLEPSO, MBX_F_FDR_FAST honoured: the kernels without the near-tie flag / second pass of the FDR scan (include/mbx.h); cfg.flags holds
                                           // the EFFECTIVE flags (environment overrides OR-ed in, MBX_F_FDR_FAST cleared where no fast instantiation exists)
    bool rl_run_kinds_ok = false;          // RLEPSO: every problem of the batch is one of the 24 BBOB kinds the per-kind k_rlepso_run geometries have a body for (rl_run_kind_ok)
    bool lde_run_kinds_ok = false;         // LDE: every problem of the batch has an objective kind the LEAN instantiations of k_lde_run build (one tile array: lde_run_kind_ok(.., two = false))
    bool lde_run_kinds_two = false;        // ... a kind the instantiations with the second tile array build (F1-F24)
    bool rollout_per_generation = false;   // MBX_F_ROLLOUT_PER_GENERATION: the mbx_*_rollout entry points take the host-loop route
    int64_t state_stride = 0;
    const double* d_tape = nullptr;
    size_t lds_bytes = 0;
    int64_t sc_off = 0;          // offset of the scalar block inside an instance's state
    int64_t tape_stride = 0;
    int state_dim = 0, action_dim = 0;
    int threads = kThreads;      // workgroup size of the RLEPSO generation kernels (512 for LDS-bound geometries)
    // compile-time-geometry instantiation of the generation / rollout kernels (0 = geometry read from the batch):
    //   1 = RLEPSO NP 100 / D 10 / 5 groups (BASELINE configs 1-2)      2 = RLEPSO NP 128 / D 40 / 5 groups (config 5)      7 = RLEPSO NP 100 / D 30 (bbob --dim 30)
    //   8 / 10 = RLEPSO NP 100 at D 12 (protein docking) / D 40 (bbob --dim 40): resident kernel only, mbx_step stays on the run-time-geometry kernel
    //   3 / 6 = LDE NP 50 / NP 100 at D 30 (config 3)      9 = LDE NP 50 / D 10 (resident kernel only)      4 = DE-DDQN NP 100 / D 12 (config 4)      5 = GLEET NP 100 / D 10
    int fixed_geometry = 0;
};

// per-algorithm geometry
struct AlgoGeom { int64_t state_doubles, sc_off, tape_stride, lds_doubles; int state_dim, action_dim; };

static AlgoGeom geom_of(const mbx_algo_cfg& c)
{
    AlgoGeom g{};
    if (c.algo == MBX_ALGO_RLEPSO) {
        g.state_doubles = MBX_RLEPSO_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_RLEPSO_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_RLEPSO_TAPE_STRIDE(c.np, c.dim);
        // compile-time geometries that read their maps from global memory need no LDS for them (unless the generic kernels are asked for)
        g.lds_doubles = rl_lds_doubles(c.np, c.dim, (c.flags & MBX_F_GENERIC_GEOMETRY) || c.n_group != 5 || rl_maps_in_lds(c.np, c.dim));
        g.state_dim = 1; g.action_dim = 7 * c.n_group;
    } else if (c.algo == MBX_ALGO_LDE) {
        g.state_doubles = MBX_LDE_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_LDE_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_LDE_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = lde_lds_doubles(c.np, c.dim, lde_maps_in_lds(c.np, c.dim) || (c.flags & MBX_F_GENERIC_GEOMETRY));
        g.state_dim = c.np + 2 * MBX_LDE_BINS; g.action_dim = 2 * c.np;
    } else if (c.algo == MBX_ALGO_DEDDQN) {
        g.state_doubles = MBX_DQ_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_DQ_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_DQ_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = dq_lds_doubles(c.np, c.np, c.dim);          // k_dq_reset; k_dq_step launches with the one-row layout
        g.state_dim = MBX_DQ_NFEAT; g.action_dim = 1;
    } else if (c.algo == MBX_ALGO_RANDOM_SEARCH) {
        g.state_doubles = MBX_RS_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_RS_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_RS_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = rs_lds_doubles(c.np, c.dim);
        g.state_dim = 1; g.action_dim = 0;
    } else if (c.algo == MBX_ALGO_RLPSO) {
        g.state_doubles = MBX_RLPSO_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_RLPSO_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_RLPSO_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = rp_lds_doubles(c.np, c.dim, 0);
        g.state_dim = 2 * c.dim; g.action_dim = 1;
    } else if (c.algo == MBX_ALGO_GLEET) {
        g.state_doubles = MBX_GLEET_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_GLEET_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_GLEET_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = gl_lds_doubles(c.np, c.dim);
        g.state_dim = 27 * c.np; g.action_dim = c.np;
    } else if (c.algo == MBX_ALGO_QLPSO) {
        g.state_doubles = MBX_QLPSO_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = MBX_QLPSO_ST_SCALARS(c.np, c.dim);
        g.tape_stride = MBX_QLPSO_TAPE_STRIDE(c.np, c.dim);
        g.lds_doubles = ql_lds_doubles(c.np, c.np, c.dim);
        g.state_dim = 1; g.action_dim = 1;
    } else if (c.algo == MBX_ALGO_DE || c.algo == MBX_ALGO_PSO || c.algo == MBX_ALGO_CMAES) {
        g.state_doubles = c.algo == MBX_ALGO_DE ? MBX_DE_STATE_DOUBLES(c.np, c.dim, c.n_logpoint)
                        : c.algo == MBX_ALGO_PSO ? MBX_PSO_STATE_DOUBLES(c.np, c.dim, c.n_logpoint) : MBX_CMA_STATE_DOUBLES(c.np, c.dim, c.n_logpoint);
        g.sc_off = c.algo == MBX_ALGO_DE ? MBX_DE_ST_SCALARS(c.np, c.dim) : c.algo == MBX_ALGO_PSO ? MBX_PSO_ST_SCALARS(c.np, c.dim) : MBX_CMA_ST_SCALARS(c.np, c.dim);
        g.tape_stride = 0;
        g.lds_doubles = cl_lds_doubles(c.np, c.np, c.dim, 0, c.algo == MBX_ALGO_CMAES);       // the largest carve-up of the family
        g.state_dim = 1; g.action_dim = 0;
    }
    return g;
}

extern "C" int mbx_suite_destroy(mbx_suite* s);
extern "C" int mbx_batch_destroy(mbx_batch* b);

// ------------------------------------------------------------------------------------------------ kernels
// Stand-alone evaluation: each block stages the problem's linear maps and evaluates up to `rows` rows.
__global__ __launch_bounds__(kThreads) void k_eval(const DevProblem* problems, int problem, const double* __restrict__ x, int n,
                                                   int rows, double* __restrict__ f, int noisy, uint64_t seed,
                                                   const double* __restrict__ draws)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DevProblem P = problems[problem];
    const int D = P.dim, tid = threadIdx.x;
    const int row0 = blockIdx.x * rows;
    const int m = min(rows, n - row0);
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D);
    const int64_t DV = align2(D);
    double* X = smem;
    double* T = X + NE;
    double* Z = T + eval_t_doubles(rows, D);
    double* M1T = Z + SC;
    double* M2T = M1T + DD;
    double* VEC = M2T + DD;
    double* F = VEC + 4 * DV;
    const EvalLds L{X, Z, T, M1T, M2T, VEC, VEC + DV, VEC + 2 * DV, VEC + 3 * DV, F};
    stage_problem(P, L);
    for (int e = tid; e < m * D; e += kThreads) X[e] = x[(int64_t)row0 * D + e];
    __syncthreads();
    eval_rows(P, L, m);
    for (int i = tid; i < m; i += kThreads) {
        double v = F[i];
        if (noisy && P.noise_kind != MBX_NOISE_NONE) {
            double a, b, c;
            const int r = row0 + i;
            if (draws) { a = draws[r]; b = draws[n + r]; c = draws[2 * (int64_t)n + r]; }
            else {
                const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), 0u, 0u};
                philox_noise(rng, (uint32_t)r, MBX_SITE_EVAL_A, MBX_SITE_EVAL_B, P.noise_kind, a, b, c);
            }
            v = apply_noise(P, v, a, b, c);
        }
        f[row0 + i] = v;
    }
}

static size_t eval_lds_bytes(int rows, int D)
{
    const int64_t NE = align2((int64_t)rows * D), SC = align2(NE > 2 * kThreads ? NE : 2 * kThreads), DD = align2((int64_t)D * D);
    return (size_t)(NE + eval_t_doubles(rows, D) + SC + 2 * DD + 4 * align2(D) + align2(rows)) * sizeof(double);
}

__global__ void k_init_state(double* state, int64_t stride, int64_t sc_off, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) state[(int64_t)b * stride + sc_off + MBX_SC_EPISODE] = -1.;
}

__global__ void k_set_optimum(DevProblem* problems, const double* opt, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) problems[i].optimum = opt[i];
}

// rollout_episode's return fields for every instance (rlepso_agent.py:303), cost padded with its last value
__global__ void k_results(const double* state, int64_t stride, int64_t sc_off, int B, int nlog, double* cost, double* fes,
                          double* ret, int32_t* steps, int32_t* cost_len)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double* sc = state + (int64_t)b * stride + sc_off;
    const int len = (int)sc[MBX_SC_COST_LEN];
    if (cost)
        for (int k = 0; k <= nlog; ++k) cost[(int64_t)b * (nlog + 1) + k] = sc[MBX_NSCALAR + (k < len ? k : len - 1)];
    if (fes) fes[b] = sc[MBX_SC_FES];
    if (ret) ret[b] = sc[MBX_SC_RETURN];
    if (steps) steps[b] = (int32_t)sc[MBX_SC_GEN];
    if (cost_len) cost_len[b] = len;
}

// ------------------------------------------------------------------------------------------------ suite
static int max_lds_bytes()
{
    int v = 0, dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || v <= 0) v = 64 * 1024;
    return v;
}

static int launch_eval(mbx_suite* s, int problem, const double* d_x, int n, double* d_f, int noisy, uint64_t seed,
                       const double* d_draws, hipStream_t st)
{
    if (n <= 0) return MBX_OK;
    const int D = s->dim;
    int rows = n < 128 ? n : 128;
    const int cap = max_lds_bytes();
    while (rows > 1 && eval_lds_bytes(rows, D) > (size_t)cap) rows /= 2;
    const size_t lds = eval_lds_bytes(rows, D);
    if (lds > (size_t)cap) return fail(MBX_E_UNSUPPORTED, "dim %d needs %zu B of LDS (> %d)", D, lds, cap);
    HIP_TRY(hipFuncSetAttribute((const void*)k_eval, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = (n + rows - 1) / rows;
    hipLaunchKernelGGL(k_eval, dim3(grid), dim3(kThreads), lds, st, s->d_problems, problem, d_x, n, rows, d_f, noisy, seed,
                       d_draws);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_suite_create(const mbx_problem_desc* descs, int n_problems, const double* opt, mbx_suite** out)
{
    if (!descs || n_problems <= 0 || !out) return fail(MBX_E_ARG, "mbx_suite_create: bad arguments");
    const int D = descs[0].dim;
    if (D < 2 || D > 64) return fail(MBX_E_ARG, "dim %d outside [2, 64]", D);
    std::vector<double> pool;
    std::vector<DevProblem> hp(n_problems);
    struct Off { int64_t o[11]; };
    std::vector<Off> offs(n_problems);
    auto push = [&](const double* p, size_t n) -> int64_t {
        if (!p) return -1;
        while (pool.size() % 2) pool.push_back(0.);
        const int64_t o = (int64_t)pool.size();
        pool.insert(pool.end(), p, p + n);
        return o;
    };
    for (int i = 0; i < n_problems; ++i) {
        const mbx_problem_desc& d = descs[i];
        if (d.dim != D) return fail(MBX_E_ARG, "problem %d has dim %d, suite dim is %d", i, d.dim, D);
        const bool bbob = d.kind >= 1 && d.kind <= 24, protein = d.kind == MBX_KIND_PROTEIN;
        if (!bbob && !protein) return fail(MBX_E_UNSUPPORTED, "problem %d: objective kind %d is not implemented", i, d.kind);
        if (bbob && (!d.dshift || !d.m1)) return fail(MBX_E_ARG, "problem %d: dshift/m1 missing", i);
        if (protein && (!d.v0 || !d.py || !d.pc || !d.pw || d.n_peaks < 2 || d.n_peaks > 160))
            return fail(MBX_E_ARG, "problem %d: protein tables missing (v0, py, pc, pw) or n_atoms outside [2,160]", i);
        if ((d.kind == 21 || d.kind == 22) && (!d.py || !d.pc || !d.pw || d.n_peaks <= 0))
            return fail(MBX_E_ARG, "problem %d: Gallagher tables missing", i);
        DevProblem& p = hp[i];
        memset(&p, 0, sizeof(p));
        p.func_id = d.func_id; p.kind = d.kind; p.dim = d.dim; p.n_peaks = d.n_peaks; p.noise_kind = d.noise_kind;
        p.bias = d.bias; p.lb = d.lb; p.ub = d.ub; p.pen_coef = d.pen_coef;
        for (int k = 0; k < 4; ++k) p.s[k] = d.s[k];
        p.noise_a = d.noise_a; p.noise_b = d.noise_b; p.optimum = NAN;
        const size_t DD = (size_t)D * D, NA = (size_t)d.n_peaks;
        const size_t PK = protein ? 3 * NA * D : NA * D;                // basis [D,3n]   | Gallagher y [n_peaks,D]
        const size_t PC = protein ? 3 * NA : NA * D;                    // coor_init [3n] | Gallagher C
        const size_t PW = protein ? 3 * NA * NA : NA;                   // sqrt(e)|q|r    | Gallagher w
        Off& o = offs[i];
        o.o[0] = push(d.dshift, D); o.o[1] = push(d.m1, DD); o.o[2] = push(d.m2, DD);
        o.o[3] = push(d.v0, D); o.o[4] = push(d.v1, D); o.o[5] = push(d.v2, D);
        o.o[6] = push(d.py, PK); o.o[7] = push(d.pc, PC); o.o[8] = push(d.pw, PW);
        o.o[9] = -1; o.o[10] = -1;
        if (d.kind == 21 || d.kind == 22) {
            std::vector<double> lw(NA);
            for (size_t k = 0; k < NA; ++k) lw[k] = std::log(d.pw[k]);
            o.o[10] = push(lw.data(), NA);
            // R y_k for every peak: z_k = R (x - y_k) = R x - R y_k lets the kernel rotate x once per row
            // instead of once per (row, peak) pair (10x fewer flops at D = 10); rounding differs by O(ulp |R x|).
            std::vector<double> ry(NA * D);
            for (int k = 0; k < d.n_peaks; ++k)
                for (int r = 0; r < D; ++r) {
                    double acc = 0.;
                    for (int j = 0; j < D; ++j) acc = std::fma(d.m1[(size_t)r * D + j], d.py[(size_t)k * D + j], acc);   // the kernels' matvec chain
                    ry[(size_t)k * D + r] = acc;
                }
            o.o[9] = push(ry.data(), NA * D);
        }
        if (protein) {
            // The energy kernel visits the atom pairs i < j only (eval_rows_protein), in the order of the folded rectangle
            // ceil(n / 2) x (n - 1): pair t = a (n - 1) + b is (a, a + 1 + b) for b < n - 1 - a, else (n - 1 - a, ...).  Its inputs are laid
            // out per pair: `pyr` <- [n_pairs][4] = sqrt(e) | q | r | 0 (one 32-byte record per pair: lane t reads record t, fully
            // coalesced), `plogw` <- the pair's atoms as int32 i | j << 16 (two per double), -1 for an empty slot.  Requires what the reference constructs
            // (protein_docking.py:175-181): symmetric tables.
            const int n = d.n_peaks, W = n - 1, n_pairs = ((n + 1) / 2) * W;
            const double *se = d.pw, *qm = d.pw + NA * NA, *rm = d.pw + 2 * NA * NA;
            for (int a = 0; a < n; ++a)
                for (int c = a + 1; c < n; ++c)
                    if (se[a * n + c] != se[c * n + a] || qm[a * n + c] != qm[c * n + a] || rm[a * n + c] != rm[c * n + a])
                        return fail(MBX_E_ARG, "problem %d: the protein tables sqrt(e) | q | r must be symmetric (atoms %d, %d)", i, a, c);
            // Pairs in ascending order of the SMALLEST distance they can reach: an atom moves by sum_k x_k v0_k basis[k][.] with |x_k| <= ub, i.e. by at most
            // bd_m = |ub sum_k v0_k |basis[k][3 m + c]||_2 (0.7 A on average over the 280 problems), so a pair never comes closer than d0 - bd_i - bd_j.
            // The ~52 % of the pairs (38-64 %) whose bound stays beyond the 9 A cut-off sit at the end of the list and are never visited for a candidate inside the
            // box (DevProblem::n_close; eval_rows_protein); of the rest, whole waves that see no pair inside the cut-off skip the arithmetic.
            std::vector<int> pi_(n_pairs), pj_(n_pairs), order;
            std::vector<double> d0(n_pairs, INFINITY);
            std::vector<double> bd(n, 0.);                             // per-atom displacement bound over the box
            for (int m = 0; m < n; ++m) {
                double q2 = 0.;
                for (int c = 0; c < 3; ++c) {
                    double bsum = 0.;
                    for (int k = 0; k < D; ++k) bsum += std::fabs(d.v0[k] * d.py[(size_t)k * 3 * n + 3 * m + c]);
                    bsum *= std::fmax(std::fabs(d.ub), std::fabs(d.lb));
                    q2 += bsum * bsum;
                }
                bd[m] = std::sqrt(q2) * (1. + 1e-12);
            }
            for (int t = 0; t < n_pairs; ++t) {
                const int a = t / W, b = t - a * W, La = W - a;
                const bool lower = b >= La;
                pi_[t] = lower ? W - a : a; pj_[t] = pi_[t] + 1 + (lower ? b - La : b);
                if (lower && pi_[t] == a) { pi_[t] = -1; continue; }   // odd n: the middle atom's row appears once; the empty slots go last
                double q2 = 0.;
                for (int c = 0; c < 3; ++c) { const double dd = d.pc[3 * pi_[t] + c] - d.pc[3 * pj_[t] + c]; q2 += dd * dd; }
                d0[t] = std::sqrt(q2) - bd[pi_[t]] - bd[pj_[t]];        // smallest reachable distance
            }
            int n_close = 0;
            for (int t = 0; t < n_pairs; ++t) n_close += pi_[t] >= 0 && d0[t] <= 9.0 + 1e-6;
            hp[i].n_close = n_close;
            order.resize(n_pairs);
            for (int t = 0; t < n_pairs; ++t) order[t] = t;
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return d0[x] < d0[y]; });
            std::vector<double> rec((size_t)n_pairs * 4, 0.);
            std::vector<double> ij((size_t)(n_pairs + 1) / 2, 0.);
            int32_t* ijw = reinterpret_cast<int32_t*>(ij.data());
            for (int t = 0; t < n_pairs; ++t) {
                const int pi = pi_[order[t]], pj = pj_[order[t]];
                if (pi < 0) { ijw[t] = -1; continue; }
                ijw[t] = pi | (pj << 16);
                rec[(size_t)t * 4] = se[pi * n + pj]; rec[(size_t)t * 4 + 1] = qm[pi * n + pj]; rec[(size_t)t * 4 + 2] = rm[pi * n + pj];
            }
            o.o[9] = push(rec.data(), rec.size());
            o.o[10] = push(ij.data(), ij.size());
        }
    }
    std::unique_ptr<mbx_suite, int (*)(mbx_suite*)> guard(new mbx_suite(), mbx_suite_destroy);   // freed on every error return
    mbx_suite* s = guard.get();
    s->n = n_problems; s->dim = D;
    HIP_TRY(hipMalloc(&s->d_pool, pool.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(s->d_pool, pool.data(), pool.size() * sizeof(double), hipMemcpyHostToDevice));
    for (int i = 0; i < n_problems; ++i) {
        const double** ptrs[11] = {&hp[i].dshift, &hp[i].m1, &hp[i].m2, &hp[i].v0, &hp[i].v1,
                                   &hp[i].v2, &hp[i].py, &hp[i].pc, &hp[i].pw, &hp[i].pyr, &hp[i].plogw};
        for (int k = 0; k < 11; ++k) *ptrs[k] = offs[i].o[k] < 0 ? nullptr : s->d_pool + offs[i].o[k];
    }
    HIP_TRY(hipMalloc(&s->d_problems, n_problems * sizeof(DevProblem)));
    HIP_TRY(hipMemcpy(s->d_problems, hp.data(), n_problems * sizeof(DevProblem), hipMemcpyHostToDevice));
    s->h_problems = hp;
    s->optimum.assign(n_problems, NAN);
    if (opt) {          // optimum_i = func_i(opt_i), noise-free, like BBOB_Basic_Problem.__init__ (bbob.py:42)
        double *d_x = nullptr, *d_f = nullptr;
        HIP_TRY(hipMalloc(&d_x, (size_t)n_problems * D * sizeof(double)));
        HIP_TRY(hipMalloc(&d_f, (size_t)n_problems * sizeof(double)));
        HIP_TRY(hipMemcpy(d_x, opt, (size_t)n_problems * D * sizeof(double), hipMemcpyHostToDevice));
        for (int i = 0; i < n_problems; ++i) {
            int rc = launch_eval(s, i, d_x + (size_t)i * D, 1, d_f + i, 0, 0, nullptr, nullptr);
            if (rc) return rc;
        }
        hipLaunchKernelGGL(k_set_optimum, dim3((n_problems + 63) / 64), dim3(64), 0, nullptr, s->d_problems, d_f, n_problems);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(s->optimum.data(), d_f, n_problems * sizeof(double), hipMemcpyDeviceToHost));
        for (int i = 0; i < n_problems; ++i) s->h_problems[i].optimum = s->optimum[i];
        (void)hipFree(d_x); (void)hipFree(d_f);
    }
    *out = guard.release();
    return MBX_OK;
}

extern "C" int mbx_suite_destroy(mbx_suite* s)
{
    if (!s) return MBX_OK;
    (void)hipFree(s->d_pool); (void)hipFree(s->d_problems);
    delete s;
    return MBX_OK;
}

extern "C" int mbx_suite_size(const mbx_suite* s) { return s ? s->n : fail(MBX_E_ARG, "null suite"); }

extern "C" int mbx_suite_optimum(const mbx_suite* s, double* optimum_out)
{
    if (!s || !optimum_out) return fail(MBX_E_ARG, "mbx_suite_optimum: bad arguments");
    memcpy(optimum_out, s->optimum.data(), s->n * sizeof(double));
    return MBX_OK;
}

extern "C" int mbx_suite_close_pairs(const mbx_suite* s, int problem)
{
    if (!s || problem < 0 || problem >= s->n) return fail(MBX_E_ARG, "mbx_suite_close_pairs: bad arguments");
    return s->h_problems[problem].kind == MBX_KIND_PROTEIN ? s->h_problems[problem].n_close : -1;
}

extern "C" int mbx_eval(mbx_suite* s, int problem, const double* d_x, int n, double* d_f, int noisy, uint64_t seed,
                        const double* d_noise_draws, void* stream)
{
    if (!s || problem < 0 || problem >= s->n || n < 0 || (n > 0 && (!d_x || !d_f)))
        return fail(MBX_E_ARG, "mbx_eval: bad arguments");
    return launch_eval(s, problem, d_x, n, d_f, noisy, seed, d_noise_draws, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------ batch
static int check_cfg(const mbx_algo_cfg* c)
{
    if (!c) return fail(MBX_E_ARG, "null cfg");
    if (c->algo < MBX_ALGO_RLEPSO || c->algo > MBX_ALGO_CMAES)
        return fail(MBX_E_UNSUPPORTED, "algo %d is not implemented in this build", c->algo);
    if (c->np < 4 || c->np > kThreads) return fail(MBX_E_ARG, "np %d outside [4, %d]", c->np, kThreads);
    if (c->dim < 2 || c->dim > 64) return fail(MBX_E_ARG, "dim %d outside [2, 64]", c->dim);
    if (c->algo == MBX_ALGO_RLEPSO && (c->n_group < 1 || c->n_group > 16 || c->np / c->n_group < 1))
        return fail(MBX_E_ARG, "bad n_group %d", c->n_group);
    if (c->max_fes <= 0 || c->log_interval <= 0 || c->n_logpoint <= 0) return fail(MBX_E_ARG, "bad budget/log settings");
    if (c->flags & ~(MBX_F_FDR_FAST | MBX_F_GENERIC_GEOMETRY | MBX_F_ROLLOUT_PER_GENERATION)) return fail(MBX_E_ARG, "unknown bits in cfg.flags 0x%x", c->flags);
    return MBX_OK;
}

extern "C" int mbx_state_dim(const mbx_algo_cfg* c)
{
    if (int rc = check_cfg(c)) return rc;
    return geom_of(*c).state_dim;
}

extern "C" int mbx_action_dim(const mbx_algo_cfg* c)
{
    if (int rc = check_cfg(c)) return rc;
    return geom_of(*c).action_dim;
}

extern "C" int64_t mbx_tape_stride(const mbx_algo_cfg* c)
{
    if (int rc = check_cfg(c)) return rc;
    return geom_of(*c).tape_stride;
}

// Longest-processing-time-first launch order: workgroups of the expensive objectives are dispatched first so that they do not form the
// tail of the launch.
static int upload_launch_order(mbx_batch* b, const int32_t* problem_idx)
{
    const mbx_suite* s = b->suite;
    const int n_instances = b->B;
    auto weight = [&](int pi) -> int {
        const int k = s->h_problems[pi].kind;
        // per-kind weights = ns per instance-generation x 10 of the resident RLEPSO kernel, tools/kbench_costs.py at the round-4 head, per-kind bodies (the table
        // metabox_amd/distributed.py: COST_NS holds for the inter-rank partition): D = 40 / NP = 128 for the large dimensions, D = 10 / NP = 100 otherwise
        if (b->cfg.algo == MBX_ALGO_LDE && s->dim >= 16 && k != MBX_KIND_PROTEIN) {
            // LDE at D = 30 (k_lde_run, pop 100; tools/exp/lde_run.py --functions, round 4): us per generation of 16 384 instances of that kind / 10.  The cost
            // ranking differs from RLEPSO's (no FDR scan to dilute the objective: Schaffers and the Gallagher search weigh 2-2.5x Sphere).
            switch (k) {
            case 21: case 22: return 155; case 17: case 18: return 121; case 16: case 23: return 110; case 15: case 3: case 4: return 100; case 2: case 10: case 11: return 92;
            case 14: case 12: return 81; case 7: return 78; case 19: case 24: case 20: return 77; case 8: case 9: case 6: return 66; default: return 62;
            }
        }
        if (s->dim >= 16 && k != MBX_KIND_PROTEIN) {
            switch (k) {
            case 21: return 3016; case 22: return 2378; case 16: return 2342; case 15: return 2285; case 18: return 2283; case 17: return 2255;
            case 3: return 2077; case 23: return 2049; case 4: return 1980; case 7: return 1885; case 12: return 1840; case 2: return 1787;
            case 11: return 1784; case 10: return 1783; case 24: return 1682; case 14: return 1656; case 19: return 1607; case 1: return 1588;
            case 6: return 1566; case 8: return 1533; case 9: return 1528; case 20: return 1526; case 13: return 1499; case 5: return 1334;
            default: return 1886;
            }
        }
        switch (k) {
        case MBX_KIND_PROTEIN: return 30000 + s->h_problems[pi].n_close;     // the energy walks the n_close atom pairs that can reach the 9 A cut-off (38-64 % of the 4950): most pairs first
        // (round 6: ns per instance-generation x 10 of the exact-FDR resident kernel over generations 3-152 of one-function batches, tools/kbench_costs.py --gens 150)
        case 21: return 416; case 3: return 332; case 15: return 326; case 16: return 325; case 22: return 323; case 23: return 316;
        case 18: return 310; case 17: return 310; case 4: return 309; case 2: return 282; case 12: return 282; case 11: return 275;
        case 10: return 270; case 6: return 254; case 14: return 252; case 1: return 251; case 20: return 250; case 19: return 247;
        case 24: return 246; case 8: return 244; case 13: return 239; case 7: return 237; case 9: return 233; case 5: return 153;
        default: return 276;
        }
    };
    b->lde_run_kinds_ok = true;
    b->rl_run_kinds_ok = true;
    for (int i = 0; i < n_instances; ++i) b->rl_run_kinds_ok = b->rl_run_kinds_ok && rl_run_kind_ok(s->h_problems[problem_idx[i]].kind, s->h_problems[problem_idx[i]].noise_kind);
    b->lde_run_kinds_two = true;
    for (int i = 0; i < n_instances; ++i) {
        b->lde_run_kinds_ok = b->lde_run_kinds_ok && lde_run_kind_ok(s->h_problems[problem_idx[i]].kind, s->dim, false);
        b->lde_run_kinds_two = b->lde_run_kinds_two && lde_run_kind_ok(s->h_problems[problem_idx[i]].kind, s->dim, true);
    }
    std::vector<int32_t> order(n_instances);
    for (int i = 0; i < n_instances; ++i) order[i] = i;
    // equal weights: by kind, so that neighbours in the launch order -- the workgroups that share a CU -- run the same per-kind body (k_rlepso_run, k_lde_run)
    std::stable_sort(order.begin(), order.end(), [&](int a, int c) {
        const int wa = weight(problem_idx[a]), wc = weight(problem_idx[c]);
        return wa != wc ? wa > wc : s->h_problems[problem_idx[a]].kind < s->h_problems[problem_idx[c]].kind;
    });
    if (!b->d_order) HIP_TRY(hipMalloc(&b->d_order, n_instances * sizeof(int32_t)));
    HIP_TRY(hipMemcpy(b->d_order, order.data(), n_instances * sizeof(int32_t), hipMemcpyHostToDevice));
    return MBX_OK;
}

// the environment's test overrides of mbx_algo_cfg.flags (include/mbx.h): read here, once per batch, and nowhere else
static uint32_t env_flags()
{
    auto on = [](const char* name) { const char* v = getenv(name); return v && v[0] == '1'; };
    return (on("MBX_FDR_FAST") ? MBX_F_FDR_FAST : 0u) | (on("MBX_GENERIC_GEOMETRY") ? MBX_F_GENERIC_GEOMETRY : 0u) |
           (on("MBX_ROLLOUT_PER_GENERATION") ? MBX_F_ROLLOUT_PER_GENERATION : 0u);
}

extern "C" int mbx_batch_create(mbx_suite* s, const mbx_algo_cfg* cfg_in, const int32_t* problem_idx, const uint64_t* seeds,
                                int n_instances, mbx_batch** out)
{
    if (!s || !problem_idx || !seeds || n_instances <= 0 || !out) return fail(MBX_E_ARG, "mbx_batch_create: bad arguments");
    if (int rc = check_cfg(cfg_in)) return rc;
    mbx_algo_cfg effective = *cfg_in;
    effective.flags |= env_flags();
    const mbx_algo_cfg* cfg = &effective;
    if (cfg->dim != s->dim) return fail(MBX_E_ARG, "cfg.dim %d != suite dim %d", cfg->dim, s->dim);
    for (int i = 0; i < n_instances; ++i)
        if (problem_idx[i] < 0 || problem_idx[i] >= s->n) return fail(MBX_E_ARG, "problem_idx[%d]=%d out of range", i, problem_idx[i]);
    const AlgoGeom g = geom_of(*cfg);
    const size_t lds = (size_t)g.lds_doubles * sizeof(double);
    if (lds > (size_t)max_lds_bytes())
        return fail(MBX_E_UNSUPPORTED, "np=%d dim=%d needs %zu B of LDS per workgroup (> %d)", cfg->np, cfg->dim, lds, max_lds_bytes());
    std::unique_ptr<mbx_batch, int (*)(mbx_batch*)> guard(new mbx_batch(), mbx_batch_destroy);   // freed on every error return
    mbx_batch* b = guard.get();
    b->suite = s; b->cfg = *cfg; b->B = n_instances; b->lds_bytes = lds;
    // More than half a CU's LDS per workgroup means one resident workgroup per CU: give it 8 or 16 waves instead of 4 (D >= 16 / 32 keeps the
    // evaluator's per-wave scratch inside its T region).
    if (cfg->algo == MBX_ALGO_LDE && (size_t)lde_lds_doubles(cfg->np, cfg->dim, true) * sizeof(double) > 40 * 1024 && cfg->dim >= 16) b->threads = 512;   // objective-bound at D = 30: 8 waves per workgroup, -21 %
    if (cfg->algo == MBX_ALGO_RLEPSO && (size_t)rl_lds_doubles(cfg->np, cfg->dim, true) * sizeof(double) > 80 * 1024 && cfg->dim >= 16) b->threads = cfg->dim >= 32 ? 1024 : 512;   // per-wave evaluator scratch needs D >= 2 x waves
    // MBX_F_GENERIC_GEOMETRY keeps the run-time-geometry kernel (the tests compare the two instantiations bit for bit)
    {
        const bool generic = (cfg->flags & MBX_F_GENERIC_GEOMETRY) != 0;
        if (cfg->algo == MBX_ALGO_RLEPSO && cfg->n_group == 5 && !generic) {
            if (b->threads == kThreads && cfg->np == 100 && cfg->dim == 10) b->fixed_geometry = 1;
            if (b->threads == 1024 && cfg->np == 128 && cfg->dim == 40) b->fixed_geometry = 2;
            if (b->threads == 512 && cfg->np == 100 && cfg->dim == 30) b->fixed_geometry = 7;
            if (b->threads == kThreads && cfg->np == 100 && cfg->dim == 12) b->fixed_geometry = 8;     // protein docking: resident rollout only, the one-generation kernel stays the run-time-geometry one
            if (b->threads == 1024 && cfg->np == 100 && cfg->dim == 40) b->fixed_geometry = 10;        // the reference's own NP at bbob --dim 40 (config.py:74, rlepso_optimizer.py:11): resident rollout only, likewise
        }
        // the fast FDR scan exists for the run-time-geometry kernels and the geometries of BASELINE configs 2 / 5; the other compile-time geometries keep the exact scan on
        // BOTH routes (so that the two stay bit-identical), and the batch's effective flags say so
        if (cfg->algo != MBX_ALGO_RLEPSO || b->fixed_geometry == 7 || b->fixed_geometry == 8 || b->fixed_geometry == 10) b->cfg.flags &= ~MBX_F_FDR_FAST;
        b->fdr_fast = (b->cfg.flags & MBX_F_FDR_FAST) != 0;
        // config 3 (LDE, NP 50 / D 30, 512 threads) and config 4 (DE-DDQN, NP 100 / D 12)
        if (cfg->algo == MBX_ALGO_LDE && b->threads == 512 && cfg->np == 50 && cfg->dim == 30 && !generic) b->fixed_geometry = 3;
        if (cfg->algo == MBX_ALGO_LDE && b->threads == 512 && cfg->np == 100 && cfg->dim == 30 && !generic) b->fixed_geometry = 6;
        // the reference's own LDE setting (lde_optimizer.py:10 NP = 50, bbob --dim 10): resident rollout only, the one-generation kernel stays the run-time-geometry one
        if (cfg->algo == MBX_ALGO_LDE && b->threads == kThreads && cfg->np == 50 && cfg->dim == 10 && !generic) b->fixed_geometry = 9;
        if (cfg->algo == MBX_ALGO_DEDDQN && cfg->np == 100 && cfg->dim == 12 && !generic) b->fixed_geometry = 4;
        if (cfg->algo == MBX_ALGO_GLEET && cfg->np == 100 && cfg->dim == 10 && !generic) b->fixed_geometry = 5;
        b->rollout_per_generation = (cfg->flags & MBX_F_ROLLOUT_PER_GENERATION) != 0;
    }
#ifdef MBX_LDS_PAD_EXPERIMENT
    if (const char* e = getenv("MBX_LDS_PAD")) b->lds_bytes += (size_t)atoi(e);      // occupancy experiments only
#endif
    b->state_stride = (g.state_doubles + 1) & ~(int64_t)1;
    b->sc_off = g.sc_off; b->tape_stride = g.tape_stride; b->state_dim = g.state_dim; b->action_dim = g.action_dim;
    HIP_TRY(hipMalloc(&b->d_problem_idx, n_instances * sizeof(int32_t)));
    HIP_TRY(hipMalloc(&b->d_seeds, n_instances * sizeof(uint64_t)));
    HIP_TRY(hipMalloc(&b->d_state, (size_t)n_instances * b->state_stride * sizeof(double)));
    HIP_TRY(hipMemcpy(b->d_problem_idx, problem_idx, n_instances * sizeof(int32_t), hipMemcpyHostToDevice));
    b->h_problem_idx.assign(problem_idx, problem_idx + n_instances);
    HIP_TRY(hipMemcpy(b->d_seeds, seeds, n_instances * sizeof(uint64_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(b->d_state, 0, (size_t)n_instances * b->state_stride * sizeof(double)));
    if (int rc = upload_launch_order(b, problem_idx)) return rc;
    hipLaunchKernelGGL(k_init_state, dim3((n_instances + 255) / 256), dim3(256), 0, nullptr, b->d_state, b->state_stride,
                       g.sc_off, n_instances);
    HIP_TRY(hipDeviceSynchronize());
    if (cfg->algo == MBX_ALGO_RLEPSO) {
        {   // RLEPSO learning-probability curve (rlepso_optimizer.py:23-24): pci_i = 0.05 + 0.45 exp(10 i/(NP-1)) / (e^10 - 1)
        std::vector<double> pci(cfg->np);
        for (int i = 0; i < cfg->np; ++i) pci[i] = 0.05 + 0.45 * std::exp(10. * i / (cfg->np - 1)) / (std::exp(10.) - 1);
        HIP_TRY(hipMalloc(&b->d_pci, cfg->np * sizeof(double)));
        HIP_TRY(hipMemcpy(b->d_pci, pci.data(), cfg->np * sizeof(double), hipMemcpyHostToDevice));
        // mbx_rlepso_rollout never allocates or reads the environment (it may run under stream capture): both happen here
        HIP_TRY(hipMalloc(&b->d_scratch, (size_t)n_instances * sizeof(double)));
    }
#define MBX_RL_LDS(...) HIP_TRY(hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds))
        MBX_RL_LDS(k_rlepso_reset<kThreads>); MBX_RL_LDS(k_rlepso_reset<512>); MBX_RL_LDS(k_rlepso_reset<1024>); MBX_RL_LDS(k_rlepso_reset<512, 100, 30>);
        // generation kernels: run-time geometry and the geometries of BASELINE configs 2 / 5 in both FDR forms, bbob --dim 30 exact only
        MBX_RL_LDS(k_rlepso_step<kThreads, 0, 0, 0, true>); MBX_RL_LDS(k_rlepso_step<512, 0, 0, 0, true>); MBX_RL_LDS(k_rlepso_step<1024, 0, 0, 0, true>);
        MBX_RL_LDS(k_rlepso_step<kThreads, 0, 0, 0, false>); MBX_RL_LDS(k_rlepso_step<512, 0, 0, 0, false>); MBX_RL_LDS(k_rlepso_step<1024, 0, 0, 0, false>);
        MBX_RL_LDS(k_rlepso_step<kThreads, 100, 10, 5, true>); MBX_RL_LDS(k_rlepso_step<kThreads, 100, 10, 5, false>);
        MBX_RL_LDS(k_rlepso_step<1024, 128, 40, 5, true>); MBX_RL_LDS(k_rlepso_step<1024, 128, 40, 5, false>);
        MBX_RL_LDS(k_rlepso_step<512, 100, 30, 5, true>);
        MBX_RL_LDS(k_rlepso_run<MBX_RUN10_THREADS, 100, 10, 5, true>); MBX_RL_LDS(k_rlepso_run<MBX_RUN10_THREADS, 100, 10, 5, false>);
        MBX_RL_LDS(k_rlepso_run<1024, 128, 40, 5, true>); MBX_RL_LDS(k_rlepso_run<1024, 128, 40, 5, false>);
        MBX_RL_LDS(k_rlepso_run<512, 100, 30, 5, true>); MBX_RL_LDS(k_rlepso_run<256, 100, 12, 5, true>); MBX_RL_LDS(k_rlepso_run<1024, 100, 40, 5, true>);
#undef MBX_RL_LDS
    } else if (cfg->algo == MBX_ALGO_LDE) {
        {   // mbx_lde_rollout: per-generation rewards / actions of the host-loop route
            HIP_TRY(hipMalloc(&b->d_scratch, (size_t)n_instances * (sizeof(double) + (size_t)g.action_dim * sizeof(float))));
            HIP_TRY(hipMalloc(&b->d_lstm_pack, (size_t)lde_run_pack_floats(g.state_dim, 64, g.action_dim) * sizeof(float)));      // hidden <= 64
            HIP_TRY(hipFuncSetAttribute((const void*)k_lde_run<100, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lde_run_lds_doubles(100, 30, 50, false) * sizeof(double))));
            HIP_TRY(hipFuncSetAttribute((const void*)k_lde_run<50, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lde_run_lds_doubles(50, 30, 50, false) * sizeof(double))));
            HIP_TRY(hipFuncSetAttribute((const void*)k_lde_run<50, 30, 50, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lde_run_lds_doubles(50, 30, 50, true) * sizeof(double))));
            HIP_TRY(hipFuncSetAttribute((const void*)k_lde_run<50, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lde_run_lds_doubles(50, 10, 50, true) * sizeof(double))));
        }
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_reset<kThreads>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_step<kThreads>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_reset<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_step<512>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_step<MBX_LDE50_STEP_THREADS, 50, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_step<MBX_LDE100_STEP_THREADS, 100, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_reset<512, 100, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_lde_reset<512, 50, 30>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (cfg->algo == MBX_ALGO_DEDDQN) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_dq_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_dq_step<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_dq_step<100, 12>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (cfg->algo == MBX_ALGO_DE || cfg->algo == MBX_ALGO_PSO || cfg->algo == MBX_ALGO_CMAES) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_classic_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_de_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_pso_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_cmaes_generation, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (cfg->algo == MBX_ALGO_QLPSO) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_qlpso_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_qlpso_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_qlpso_step<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (cfg->algo == MBX_ALGO_GLEET) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_gleet_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_gleet_step<>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_gleet_step<100, 10>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else if (cfg->algo == MBX_ALGO_RLPSO) {
        HIP_TRY(hipFuncSetAttribute((const void*)k_rlpso_reset, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_rlpso_step<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipFuncSetAttribute((const void*)k_rlpso_step<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    } else {
        HIP_TRY(hipFuncSetAttribute((const void*)k_rs_population, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    *out = guard.release();
    return MBX_OK;
}

extern "C" int mbx_batch_destroy(mbx_batch* b)
{
    if (!b) return MBX_OK;
    (void)hipFree(b->d_problem_idx); (void)hipFree(b->d_seeds); (void)hipFree(b->d_state); (void)hipFree(b->d_order); (void)hipFree(b->d_pci); (void)hipFree(b->d_scratch); (void)hipFree(b->d_lstm_pack);
    delete b;
    return MBX_OK;
}

extern "C" int mbx_batch_rebind(mbx_batch* b, const int32_t* problem_idx, const uint64_t* seeds)
{
    if (!b || !problem_idx || !seeds) return fail(MBX_E_ARG, "mbx_batch_rebind: bad arguments");
    for (int i = 0; i < b->B; ++i)
        if (problem_idx[i] < 0 || problem_idx[i] >= b->suite->n) return fail(MBX_E_ARG, "problem_idx[%d]=%d out of range", i, problem_idx[i]);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(b->d_problem_idx, problem_idx, b->B * sizeof(int32_t), hipMemcpyHostToDevice));
    b->h_problem_idx.assign(problem_idx, problem_idx + b->B);
    HIP_TRY(hipMemcpy(b->d_seeds, seeds, b->B * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (int rc = upload_launch_order(b, problem_idx)) return rc;
    hipLaunchKernelGGL(k_init_state, dim3((b->B + 255) / 256), dim3(256), 0, nullptr, b->d_state, b->state_stride, b->sc_off, b->B);   // episode counter back to -1
    HIP_TRY(hipDeviceSynchronize());
    return MBX_OK;
}

extern "C" int mbx_read_public(mbx_batch* b, int instance, double* host_out, void* stream)
{
    if (!b || instance < 0 || instance >= b->B || !host_out) return fail(MBX_E_ARG, "mbx_read_public: bad arguments");
    const size_t n = (size_t)(MBX_NSCALAR + b->cfg.n_logpoint + 1) * sizeof(double);
    HIP_TRY(hipMemcpyAsync(host_out, b->d_state + (int64_t)instance * b->state_stride + b->sc_off, n, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    return MBX_OK;
}

extern "C" int mbx_set_tape(mbx_batch* b, const double* d_tape)
{
    if (!b) return fail(MBX_E_ARG, "null batch");
    b->d_tape = d_tape;
    return MBX_OK;
}

static BatchParams make_params(const mbx_batch* b)
{
    BatchParams p;
    p.problems = b->suite->d_problems; p.problem_idx = b->d_problem_idx; p.seeds = b->d_seeds;
    p.state = b->d_state; p.state_stride = b->state_stride;
    p.tape = b->d_tape; p.tape_stride = b->tape_stride; p.order = b->d_order; p.pci = b->d_pci;
    p.NP = b->cfg.np; p.D = b->cfg.dim; p.max_fes = b->cfg.max_fes; p.log_interval = b->cfg.log_interval;
    p.n_logpoint = b->cfg.n_logpoint; p.early_stop = b->cfg.early_stop; p.n_group = b->cfg.n_group; p.B = b->B;
    p.sc_off = b->sc_off;
    p.clk = b->d_clk;
    return p;
}

// one RLEPSO generation: the instantiation of k_rlepso_step that fits the batch (workgroup size by LDS footprint, compile-time
// geometry for the reference's NP = 100 / D = 10 / 5 groups)
static void launch_rlepso_step(mbx_batch* b, hipStream_t stream, const float* d_actions, double* d_state_out, double* d_reward_out,
                               uint8_t* d_done_out, const float* d_table, int table_rows, float* d_actions_out)
{
#define MBX_RLEPSO_LAUNCH(...)                                                                                                   \
    hipLaunchKernelGGL((k_rlepso_step<__VA_ARGS__>), dim3(b->B), dim3(b->threads), b->lds_bytes, stream, make_params(b), d_actions, \
                       d_state_out, d_reward_out, d_done_out, d_table, table_rows, d_actions_out)
    const bool fast = b->fdr_fast;
    if (b->fixed_geometry == 1) { if (fast) MBX_RLEPSO_LAUNCH(kThreads, 100, 10, 5, false); else MBX_RLEPSO_LAUNCH(kThreads, 100, 10, 5, true); }
    else if (b->fixed_geometry == 2) { if (fast) MBX_RLEPSO_LAUNCH(1024, 128, 40, 5, false); else MBX_RLEPSO_LAUNCH(1024, 128, 40, 5, true); }
    else if (b->fixed_geometry == 7) MBX_RLEPSO_LAUNCH(512, 100, 30, 5, true);
    else if (b->threads == 1024) { if (fast) MBX_RLEPSO_LAUNCH(1024, 0, 0, 0, false); else MBX_RLEPSO_LAUNCH(1024, 0, 0, 0, true); }
    else if (b->threads == 512) { if (fast) MBX_RLEPSO_LAUNCH(512, 0, 0, 0, false); else MBX_RLEPSO_LAUNCH(512, 0, 0, 0, true); }
    else { if (fast) MBX_RLEPSO_LAUNCH(kThreads, 0, 0, 0, false); else MBX_RLEPSO_LAUNCH(kThreads, 0, 0, 0, true); }
#undef MBX_RLEPSO_LAUNCH
}

extern "C" int mbx_reset(mbx_batch* b, double* d_state_out, void* stream)
{
    if (!b) return fail(MBX_E_ARG, "null batch");
    if (b->cfg.algo == MBX_ALGO_RANDOM_SEARCH)
        hipLaunchKernelGGL(k_rs_population, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), 1, d_state_out,
                           (double*)nullptr, (uint8_t*)nullptr);
    else if (b->cfg.algo == MBX_ALGO_RLEPSO && b->fixed_geometry == 7)
        hipLaunchKernelGGL((k_rlepso_reset<512, 100, 30>), dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_RLEPSO && b->threads == 1024)
        hipLaunchKernelGGL(k_rlepso_reset<1024>, dim3(b->B), dim3(1024), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_RLEPSO && b->threads == 512)
        hipLaunchKernelGGL(k_rlepso_reset<512>, dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_RLEPSO)
        hipLaunchKernelGGL(k_rlepso_reset<kThreads>, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_RLPSO)
        hipLaunchKernelGGL(k_rlpso_reset, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_DE || b->cfg.algo == MBX_ALGO_PSO || b->cfg.algo == MBX_ALGO_CMAES)
        hipLaunchKernelGGL(k_classic_reset, dim3(b->B), dim3(kThreads), (size_t)cl_lds_doubles(b->cfg.np, b->cfg.np, b->cfg.dim, 0, 0) * sizeof(double),
                           (hipStream_t)stream, make_params(b), (int)b->cfg.algo, d_state_out);
    else if (b->cfg.algo == MBX_ALGO_QLPSO)
        hipLaunchKernelGGL(k_qlpso_reset, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else if (b->cfg.algo == MBX_ALGO_GLEET)
        hipLaunchKernelGGL(k_gleet_reset, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    else {
        if (!d_state_out) return fail(MBX_E_ARG, "mbx_reset: this algorithm needs d_state_out");
        if (b->cfg.algo == MBX_ALGO_LDE && b->fixed_geometry == 6)
            hipLaunchKernelGGL((k_lde_reset<512, 100, 30>), dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
        else if (b->cfg.algo == MBX_ALGO_LDE && b->fixed_geometry == 3)
            hipLaunchKernelGGL((k_lde_reset<512, 50, 30>), dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
        else if (b->cfg.algo == MBX_ALGO_LDE && b->threads == 512)
            hipLaunchKernelGGL(k_lde_reset<512>, dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
        else if (b->cfg.algo == MBX_ALGO_LDE)
            hipLaunchKernelGGL(k_lde_reset<kThreads>, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
        else
            hipLaunchKernelGGL(k_dq_reset, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), d_state_out);
    }
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_step(mbx_batch* b, const void* d_actions, double* d_state_out, double* d_reward_out, uint8_t* d_done_out,
                        void* stream)
{
    const bool no_agent = b && (b->cfg.algo == MBX_ALGO_RANDOM_SEARCH || b->cfg.algo == MBX_ALGO_DE || b->cfg.algo == MBX_ALGO_PSO ||
                                b->cfg.algo == MBX_ALGO_CMAES);
    if (!b || (!d_actions && !no_agent)) return fail(MBX_E_ARG, "mbx_step: bad arguments");
    if (b->cfg.algo == MBX_ALGO_RANDOM_SEARCH)
        hipLaunchKernelGGL(k_rs_population, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b), 0, d_state_out,
                           d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_RLEPSO)
        launch_rlepso_step(b, (hipStream_t)stream, (const float*)d_actions, d_state_out, d_reward_out, d_done_out, nullptr, 0, nullptr);
    else if (b->cfg.algo == MBX_ALGO_DE)
        hipLaunchKernelGGL(k_de_sweep, dim3(b->B), dim3(kThreads), (size_t)cl_lds_doubles(1, b->cfg.np, b->cfg.dim, 1, 0) * sizeof(double),
                           (hipStream_t)stream, make_params(b), d_state_out, d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_PSO)
        hipLaunchKernelGGL(k_pso_sweep, dim3(b->B), dim3(kThreads), (size_t)cl_lds_doubles(1, b->cfg.np, b->cfg.dim, 0, 0) * sizeof(double),
                           (hipStream_t)stream, make_params(b), d_state_out, d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_CMAES)
        hipLaunchKernelGGL(k_cmaes_generation, dim3(b->B), dim3(kThreads), (size_t)cl_lds_doubles(b->cfg.np, b->cfg.np, b->cfg.dim, 0, 1) * sizeof(double),
                           (hipStream_t)stream, make_params(b), d_state_out, d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_QLPSO)
        hipLaunchKernelGGL(k_qlpso_step<false>, dim3(b->B), dim3(kThreads), (size_t)ql_lds_doubles(1, b->cfg.np, b->cfg.dim) * sizeof(double),
                           (hipStream_t)stream, make_params(b), (const int32_t*)d_actions, (const double*)nullptr, 1, d_state_out,
                           d_reward_out, d_done_out, (int32_t*)nullptr);
    else if (b->cfg.algo == MBX_ALGO_GLEET && b->fixed_geometry == 5)
        hipLaunchKernelGGL((k_gleet_step<100, 10>), dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b),
                           (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_GLEET)
        hipLaunchKernelGGL(k_gleet_step<>, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b),
                           (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
    else if (b->cfg.algo == MBX_ALGO_RLPSO)
        hipLaunchKernelGGL(k_rlpso_step<false>, dim3(b->B), dim3(kThreads), (size_t)rp_lds_doubles(1, b->cfg.dim, 0) * sizeof(double),
                           (hipStream_t)stream, make_params(b), (const float*)d_actions, GaussMlp{}, 1, d_state_out, d_reward_out,
                           d_done_out, (float*)nullptr);
    else {
        if (!d_state_out) return fail(MBX_E_ARG, "mbx_step: this algorithm needs d_state_out");
        if (b->cfg.algo == MBX_ALGO_LDE && b->fixed_geometry == 3)
            hipLaunchKernelGGL((k_lde_step<MBX_LDE50_STEP_THREADS, 50, 30>), dim3(b->B), dim3(MBX_LDE50_STEP_THREADS), b->lds_bytes, (hipStream_t)stream, make_params(b),
                               (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
        else if (b->cfg.algo == MBX_ALGO_LDE && b->fixed_geometry == 6)
            hipLaunchKernelGGL((k_lde_step<MBX_LDE100_STEP_THREADS, 100, 30>), dim3(b->B), dim3(MBX_LDE100_STEP_THREADS), b->lds_bytes, (hipStream_t)stream, make_params(b),
                               (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
        else if (b->cfg.algo == MBX_ALGO_LDE && b->threads == 512)
            hipLaunchKernelGGL(k_lde_step<512>, dim3(b->B), dim3(512), b->lds_bytes, (hipStream_t)stream, make_params(b),
                               (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
        else if (b->cfg.algo == MBX_ALGO_LDE)
            hipLaunchKernelGGL(k_lde_step<kThreads>, dim3(b->B), dim3(kThreads), b->lds_bytes, (hipStream_t)stream, make_params(b),
                               (const float*)d_actions, d_state_out, d_reward_out, d_done_out);
        else if (b->fixed_geometry == 4)
            hipLaunchKernelGGL((k_dq_step<100, 12>), dim3(b->B), dim3(kDqStepThreads), (size_t)dq_lds_doubles(1, b->cfg.np, b->cfg.dim) * sizeof(double), (hipStream_t)stream, make_params(b),
                               (const int32_t*)d_actions, d_state_out, d_reward_out, d_done_out);
        else
            hipLaunchKernelGGL(k_dq_step<>, dim3(b->B), dim3(kDqStepThreads), (size_t)dq_lds_doubles(1, b->cfg.np, b->cfg.dim) * sizeof(double), (hipStream_t)stream, make_params(b),
                               (const int32_t*)d_actions, d_state_out, d_reward_out, d_done_out);
    }
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

static int check_gauss_mlp(const mbx_batch* b, const mbx_gauss_mlp* net, const char* who, size_t* lds)
{
    if (!b || !net || !net->d_weights) return fail(MBX_E_ARG, "%s: bad arguments", who);
    if (b->cfg.algo != MBX_ALGO_RLEPSO && b->cfg.algo != MBX_ALGO_RLPSO)
        return fail(MBX_E_UNSUPPORTED, "%s: the batch is neither an RLEPSO nor an RL-PSO batch", who);
    if (net->variant != (b->cfg.algo == MBX_ALGO_RLPSO ? MBX_POLICY_RLPSO : MBX_POLICY_RLEPSO))
        return fail(MBX_E_ARG, "%s: net.variant does not match the batch's algorithm", who);
    if (net->in_dim != b->state_dim || net->out_dim != b->action_dim || net->h1 < 1 || net->h2 < 1)
        return fail(MBX_E_ARG, "%s: network dimensions do not match the batch (state_dim -> h1 -> h2 -> action_dim)", who);
    *lds = gauss_mlp_lds_bytes(net->in_dim, net->h1, net->h2, net->out_dim);
    if (*lds > 64 * 1024) return fail(MBX_E_UNSUPPORTED, "%s: the weights do not fit the 64 KB LDS budget of this kernel", who);
    return MBX_OK;
}

static int policy_blocks(int rows)
{
    // every wave owns a row; two rows per wave amortise the weight staging without leaving CUs idle at 4096 rows
    const int blocks = (rows + 2 * kPolicyWaves - 1) / (2 * kPolicyWaves);
    return blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
}

extern "C" int mbx_gauss_policy(mbx_batch* b, const mbx_gauss_mlp* net, const double* d_state, float* d_actions, float* d_mu_sigma,
                                 void* stream)
{
    size_t lds = 0;
    if (const int rc = check_gauss_mlp(b, net, "mbx_gauss_policy", &lds)) return rc;
    if (!d_state || !d_actions) return fail(MBX_E_ARG, "mbx_gauss_policy: bad arguments");
    const GaussMlp g{net->d_weights, net->in_dim, net->h1, net->h2, net->out_dim, net->min_sigma, net->max_sigma, net->variant};
    hipLaunchKernelGGL(k_gauss_mlp_policy, dim3(policy_blocks(b->B)), dim3(kThreads), lds, (hipStream_t)stream, make_params(b), g,
                       d_state, d_actions, d_mu_sigma, 0);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

static int lde_policy_launch(mbx_batch* b, const mbx_lstm_policy* net, const double* d_state, float* d_h, float* d_c, float* d_actions,
                             float* d_mu_sigma, void* stream, int skip_done)
{
    if (!b || !net || !net->d_weights || !d_state || !d_h || !d_c) return fail(MBX_E_ARG, "mbx_lde_policy: bad arguments");
    if (b->cfg.algo != MBX_ALGO_LDE) return fail(MBX_E_UNSUPPORTED, "mbx_lde_policy: the batch is not an LDE batch");
    if (net->in_dim != b->state_dim || net->out_dim != b->action_dim || net->hidden < 1 || net->hidden > 64 || net->out_dim > 512)
        return fail(MBX_E_ARG, "mbx_lde_policy: network %d -> %d -> %d does not fit the batch (state %d, action %d; hidden <= 64)",
                    net->in_dim, net->hidden, net->out_dim, b->state_dim, b->action_dim);
    const LstmPolicy g{net->d_weights, net->in_dim, net->hidden, net->out_dim};
    {
        const size_t lds = lstm_policy_lds_bytes(net->in_dim, net->hidden);
        if (lds > (size_t)max_lds_bytes()) return fail(MBX_E_UNSUPPORTED, "mbx_lde_policy: %zu B of LDS needed", lds);
#define MBX_LSTM_LAUNCH(...)                                                                                                                          \
        do {                                                                                                                                          \
            HIP_TRY(hipFuncSetAttribute((const void*)k_lstm_policy<__VA_ARGS__>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));              \
            hipLaunchKernelGGL((k_lstm_policy<__VA_ARGS__>), dim3((b->B + kLstmTile - 1) / kLstmTile), dim3(kThreads), lds, (hipStream_t)stream,      \
                               make_params(b), g, d_state, d_h, d_c, d_actions, d_mu_sigma, skip_done);                                               \
        } while (0)
        // the reference's PolicyNet (lde_agent.py:8-29: LSTM NP + 10 -> 50, heads 50 -> 2 NP) at config 3's two populations: compile-time dimensions
        if (net->in_dim == 60 && net->hidden == 50 && net->out_dim == 100) MBX_LSTM_LAUNCH(kLstmTile, 60, 50, 100);
        else if (net->in_dim == 110 && net->hidden == 50 && net->out_dim == 200) MBX_LSTM_LAUNCH(kLstmTile, 110, 50, 200);
        else MBX_LSTM_LAUNCH(kLstmTile);
#undef MBX_LSTM_LAUNCH
    }
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_lde_policy(mbx_batch* b, const mbx_lstm_policy* net, const double* d_state, float* d_h, float* d_c, float* d_actions,
                              float* d_mu_sigma, void* stream)
{
    return lde_policy_launch(b, net, d_state, d_h, d_c, d_actions, d_mu_sigma, stream, 0);
}

extern "C" int mbx_ddqn_qnet(mbx_batch* b, const mbx_qnet* net, const double* d_state, int32_t* d_actions, float* d_q, void* stream)
{
    if (!b || !net || !net->d_weights || !d_state || !d_actions) return fail(MBX_E_ARG, "mbx_ddqn_qnet: bad arguments");
    if (b->cfg.algo != MBX_ALGO_DEDDQN) return fail(MBX_E_UNSUPPORTED, "mbx_ddqn_qnet: the batch is not a DE-DDQN batch");
    if (net->in_dim != b->state_dim) return fail(MBX_E_ARG, "mbx_ddqn_qnet: the network reads %d features, the batch writes %d", net->in_dim, b->state_dim);
    if (net->in_dim != 99 || net->width != 100 || net->depth != 4 || net->n_act != 4)
        return fail(MBX_E_UNSUPPORTED, "mbx_ddqn_qnet: only the reference architecture 99 -> 100 x 4 -> 4 is built (got %d -> %d x %d -> %d)",
                    net->in_dim, net->width, net->depth, net->n_act);
    const QNet g{net->d_weights, net->in_dim, net->width, net->depth, net->n_act};
    hipLaunchKernelGGL((k_qnet_argmax<99, 100, 4>), dim3((b->B + kQTile - 1) / kQTile), dim3(kQThreads), 0, (hipStream_t)stream, g, d_state, d_actions,
                       d_q, b->B);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_rlepso_policy_table_rows(const mbx_batch* b)
{
    if (!b || b->cfg.algo != MBX_ALGO_RLEPSO) return fail(MBX_E_ARG, "mbx_rlepso_policy_table_rows: not an RLEPSO batch");
    return b->cfg.max_fes + 2 * b->cfg.np + 1;       // fes < maxFEs before the last update, which bills at most 2 NP evaluations
}

extern "C" int mbx_rlepso_policy_table(mbx_batch* b, const mbx_gauss_mlp* net, float* d_table, void* stream)
{
    size_t lds = 0;
    if (const int rc = check_gauss_mlp(b, net, "mbx_rlepso_policy_table", &lds)) return rc;
    if (!d_table || net->in_dim != 1 || b->cfg.algo != MBX_ALGO_RLEPSO) return fail(MBX_E_ARG, "mbx_rlepso_policy_table: bad arguments");
    const int rows = mbx_rlepso_policy_table_rows(b);
    const GaussMlp g{net->d_weights, net->in_dim, net->h1, net->h2, net->out_dim, net->min_sigma, net->max_sigma, net->variant};
    hipLaunchKernelGGL(k_gauss_mlp_policy, dim3(policy_blocks(rows)), dim3(kThreads), lds, (hipStream_t)stream, make_params(b), g,
                       (const double*)nullptr, (float*)nullptr, d_table, rows);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_rlepso_act_step(mbx_batch* b, const float* d_table, float* d_actions_out, double* d_state_out,
                                   double* d_reward_out, uint8_t* d_done_out, void* stream)
{
    if (!b || !d_table) return fail(MBX_E_ARG, "mbx_rlepso_act_step: bad arguments");
    if (b->cfg.algo != MBX_ALGO_RLEPSO) return fail(MBX_E_UNSUPPORTED, "mbx_rlepso_act_step: the batch is not an RLEPSO batch");
    if (b->d_tape) return fail(MBX_E_ARG, "mbx_rlepso_act_step: a replay tape carries no policy draws; use mbx_step with recorded actions");
    launch_rlepso_step(b, (hipStream_t)stream, nullptr, d_state_out, d_reward_out, d_done_out, d_table, mbx_rlepso_policy_table_rows(b), d_actions_out);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_batch_flags(const mbx_batch* b)
{
    if (!b) return fail(MBX_E_ARG, "mbx_batch_flags: null batch");
    return (int)b->cfg.flags;
}

extern "C" int mbx_rlepso_rollout_resident(const mbx_batch* b)
{
    if (!b) return fail(MBX_E_ARG, "mbx_rlepso_rollout_resident: null batch");
    if (b->cfg.algo != MBX_ALGO_RLEPSO) return 0;
    return (b->fixed_geometry == 1 || b->fixed_geometry == 2 || b->fixed_geometry == 7 || b->fixed_geometry == 8 || b->fixed_geometry == 10) && !b->rollout_per_generation ? 1 : 0;
}

__global__ void k_sum_rewards(double* __restrict__ acc, const double* __restrict__ r, int n, int first)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] = first ? r[i] : acc[i] + r[i];
}

// The ONE launch site of k_rlepso_run (the D = 10 / D = 40 instantiations dispatch to one out-of-line body per function kind and run the any-kind body for a
// (kind, noise model) pair without one).
static int launch_rlepso_run(mbx_batch* b, const float* d_table, int rows, int n_gens, const RunOut& out, hipStream_t stream)
{
#define MBX_RUN_LAUNCH(T, ...) hipLaunchKernelGGL((k_rlepso_run<T, __VA_ARGS__>), dim3(b->B), dim3(T), b->lds_bytes, stream, make_params(b), d_table, rows, n_gens, out)
    if (b->fixed_geometry == 1) { if (b->fdr_fast) MBX_RUN_LAUNCH(MBX_RUN10_THREADS, 100, 10, 5, false); else MBX_RUN_LAUNCH(MBX_RUN10_THREADS, 100, 10, 5, true); }
    else if (b->fixed_geometry == 2) { if (b->fdr_fast) MBX_RUN_LAUNCH(1024, 128, 40, 5, false); else MBX_RUN_LAUNCH(1024, 128, 40, 5, true); }
    else if (b->fixed_geometry == 7) MBX_RUN_LAUNCH(512, 100, 30, 5, true);
    else if (b->fixed_geometry == 8) MBX_RUN_LAUNCH(256, 100, 12, 5, true);
    else if (b->fixed_geometry == 10) MBX_RUN_LAUNCH(1024, 100, 40, 5, true);
    else return fail(MBX_E_UNSUPPORTED, "k_rlepso_run: no instantiation for this geometry");
#undef MBX_RUN_LAUNCH
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_rlepso_rollout(mbx_batch* b, const float* d_table, int n_gens, float* d_traj_actions, double* d_traj_state,
                                  double* d_traj_reward, uint8_t* d_traj_done, double* d_state_out, double* d_reward_out,
                                  uint8_t* d_done_out, void* stream)
{
    if (!b || !d_table) return fail(MBX_E_ARG, "mbx_rlepso_rollout: bad arguments");
    if (b->cfg.algo != MBX_ALGO_RLEPSO) return fail(MBX_E_UNSUPPORTED, "mbx_rlepso_rollout: the batch is not an RLEPSO batch");
    if (n_gens < 1) return fail(MBX_E_ARG, "mbx_rlepso_rollout: n_gens must be >= 1");
    if (b->d_tape) return fail(MBX_E_ARG, "mbx_rlepso_rollout: a replay tape holds one generation and no policy draws; use mbx_step with recorded actions");
    const int rows = mbx_rlepso_policy_table_rows(b);
    if (mbx_rlepso_rollout_resident(b) == 1)
        return launch_rlepso_run(b, d_table, rows, n_gens, RunOut{d_traj_actions, d_traj_state, d_traj_reward, d_traj_done, d_state_out, d_reward_out, d_done_out},
                                 (hipStream_t)stream);
    // run-time geometries: one k_rlepso_step launch per generation, same outputs
    const int64_t B = b->B, A = b->action_dim;
    for (int g = 0; g < n_gens; ++g) {
        double* r = d_traj_reward ? d_traj_reward + g * B : (d_reward_out ? b->d_scratch : nullptr);
        double* st = d_traj_state ? d_traj_state + g * B : d_state_out;
        uint8_t* dn = d_traj_done ? d_traj_done + g * B : d_done_out;
        launch_rlepso_step(b, (hipStream_t)stream, nullptr, st, r, dn, d_table, rows, d_traj_actions ? d_traj_actions + g * B * A : nullptr);
        if (d_reward_out)
            hipLaunchKernelGGL(k_sum_rewards, dim3((b->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_reward_out, r, b->B, g == 0);
    }
    if (d_traj_state && d_state_out)
        HIP_TRY(hipMemcpyAsync(d_state_out, d_traj_state + (n_gens - 1) * B, B * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    if (d_traj_done && d_done_out)
        HIP_TRY(hipMemcpyAsync(d_done_out, d_traj_done + (n_gens - 1) * B, B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_lde_rollout_resident(const mbx_batch* b)
{
    if (!b) return fail(MBX_E_ARG, "mbx_lde_rollout_resident: null batch");
    if (b->cfg.algo != MBX_ALGO_LDE) return 0;
    if (b->rollout_per_generation) return 0;
    // NP 50 at D 10 / D 30: an instantiation with the second tile array exists (all 24 kinds); NP 100 / D 30 (config 3 as written): the lean one only
    if (b->fixed_geometry == 9 || b->fixed_geometry == 3) return (b->lde_run_kinds_ok || b->lde_run_kinds_two) ? 1 : 0;
    return b->fixed_geometry == 6 && b->lde_run_kinds_ok ? 1 : 0;
}

extern "C" int mbx_lde_rollout(mbx_batch* b, const mbx_lstm_policy* net, const double* d_state_in, float* d_h, float* d_c, int n_gens,
                               float* d_traj_actions, double* d_traj_state, double* d_traj_reward, uint8_t* d_traj_done,
                               double* d_state_out, double* d_reward_out, uint8_t* d_done_out, void* stream)
{
    if (!b || !net || !net->d_weights || !d_state_in || !d_h || !d_c || !d_state_out) return fail(MBX_E_ARG, "mbx_lde_rollout: bad arguments");
    if (b->cfg.algo != MBX_ALGO_LDE) return fail(MBX_E_UNSUPPORTED, "mbx_lde_rollout: the batch is not an LDE batch");
    if (n_gens < 1) return fail(MBX_E_ARG, "mbx_lde_rollout: n_gens must be >= 1");
    if (b->d_tape) return fail(MBX_E_ARG, "mbx_lde_rollout: a replay tape holds one generation and no policy draws; use mbx_step with recorded actions");
    if (net->in_dim != b->state_dim || net->out_dim != b->action_dim || net->hidden < 1 || net->hidden > 64)
        return fail(MBX_E_ARG, "mbx_lde_rollout: network %d -> %d -> %d does not fit the batch (state %d, action %d; hidden <= 64)",
                    net->in_dim, net->hidden, net->out_dim, b->state_dim, b->action_dim);
    const int64_t B = b->B, A = b->action_dim, NF = b->state_dim;
    if (mbx_lde_rollout_resident(b) == 1 && net->hidden == 50) {
        LdeRunArgs ka{};
        ka.bp = make_params(b);
        {   // the weights may have changed since the last call: rebuild the k-blocked copy the kernel reads (one small launch, ~3 us)
            const LstmPolicy src{net->d_weights, net->in_dim, net->hidden, net->out_dim};
            hipLaunchKernelGGL(k_lde_repack<0>, dim3(64), dim3(256), 0, (hipStream_t)stream, src, b->d_lstm_pack);
        }
        ka.net = LstmPolicy{b->d_lstm_pack, net->in_dim, net->hidden, net->out_dim};
        ka.state_in = d_state_in; ka.hbuf = d_h; ka.cbuf = d_c; ka.n_gens = n_gens;
        ka.out = LdeRunOut{d_traj_actions, d_traj_state, d_traj_reward, d_traj_done, d_state_out, d_reward_out, d_done_out};
        if (b->fixed_geometry == 6)
            hipLaunchKernelGGL((k_lde_run<100, 30>), dim3(b->B), dim3(lde_run_threads(100)), (size_t)lde_run_lds_doubles(100, 30, 50, false) * sizeof(double),
                               (hipStream_t)stream, ka);
        else if (b->fixed_geometry == 9)
            hipLaunchKernelGGL((k_lde_run<50, 10>), dim3(b->B), dim3(lde_run_threads(50)), (size_t)lde_run_lds_doubles(50, 10, 50, true) * sizeof(double),
                               (hipStream_t)stream, ka);
        else if (b->lde_run_kinds_ok)
            hipLaunchKernelGGL((k_lde_run<50, 30>), dim3(b->B), dim3(lde_run_threads(50)), (size_t)lde_run_lds_doubles(50, 30, 50, false) * sizeof(double),
                               (hipStream_t)stream, ka);
        else        // plain bbob --dim 30 at the reference's NP = 50 with F3 / F4 / F5 / F15 / F20 / F24 in the batch: the instantiation that carries the second tile array
            hipLaunchKernelGGL((k_lde_run<50, 30, 50, true>), dim3(b->B), dim3(lde_run_threads(50)), (size_t)lde_run_lds_doubles(50, 30, 50, true) * sizeof(double),
                               (hipStream_t)stream, ka);
        HIP_TRY(hipGetLastError());
        return MBX_OK;
    }
    // any other geometry / objective: mbx_lde_policy + mbx_step per generation, same outputs.  The features travel through d_state_out.
    if (d_state_in != d_state_out)
        HIP_TRY(hipMemcpyAsync(d_state_out, d_state_in, (size_t)B * NF * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    double* r_scratch = b->d_scratch;
    float* a_scratch = (float*)(b->d_scratch + B);
    for (int g = 0; g < n_gens; ++g) {
        float* acts = d_traj_actions ? d_traj_actions + g * B * A : a_scratch;
        // (h, c) and the action rows of instances that are already done stay untouched, as in the resident kernel (include/mbx.h)
        if (const int rc = lde_policy_launch(b, net, d_state_out, d_h, d_c, acts, nullptr, stream, 1)) return rc;
        double* r = d_traj_reward ? d_traj_reward + g * B : (d_reward_out ? r_scratch : nullptr);
        uint8_t* dn = d_traj_done ? d_traj_done + g * B : d_done_out;
        if (const int rc = mbx_step(b, acts, d_state_out, r, dn, stream)) return rc;
        if (d_traj_state)
            HIP_TRY(hipMemcpyAsync(d_traj_state + g * B * NF, d_state_out, (size_t)B * NF * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
        if (d_reward_out)
            hipLaunchKernelGGL(k_sum_rewards, dim3((b->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_reward_out, r, b->B, g == 0);
    }
    if (d_traj_done && d_done_out)
        HIP_TRY(hipMemcpyAsync(d_done_out, d_traj_done + (n_gens - 1) * B, B, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_rlpso_rollout(mbx_batch* b, const mbx_gauss_mlp* net, int n_steps, float* d_actions_out, double* d_state_out,
                                 double* d_reward_out, uint8_t* d_done_out, void* stream)
{
    size_t lds = 0;
    if (const int rc = check_gauss_mlp(b, net, "mbx_rlpso_rollout", &lds)) return rc;
    if (b->cfg.algo != MBX_ALGO_RLPSO) return fail(MBX_E_UNSUPPORTED, "mbx_rlpso_rollout: the batch is not an RL-PSO batch");
    if (n_steps < 1) return fail(MBX_E_ARG, "mbx_rlpso_rollout: n_steps must be >= 1");
    if (b->d_tape) return fail(MBX_E_ARG, "mbx_rlpso_rollout: a replay tape holds one step; use mbx_step with recorded actions");
    if ((size_t)(net->in_dim + 2 * net->h1 + 2 * net->h2 + 2) * sizeof(float) > (size_t)kRpActDoubles * sizeof(double))
        return fail(MBX_E_UNSUPPORTED, "mbx_rlpso_rollout: hidden layers too wide for the in-kernel actor");
    const GaussMlp g{net->d_weights, net->in_dim, net->h1, net->h2, net->out_dim, net->min_sigma, net->max_sigma, net->variant};
    const size_t step_lds = (size_t)rp_lds_doubles(1, b->cfg.dim, 2 * gauss_mlp_net_floats(net->in_dim, net->h1, net->h2, 1)) * sizeof(double);
    if (step_lds > 64 * 1024) return fail(MBX_E_UNSUPPORTED, "mbx_rlpso_rollout: the actor does not fit the LDS budget of the step kernel");
    if (n_steps == 1)
        hipLaunchKernelGGL(k_rlpso_step<false>, dim3(b->B), dim3(kThreads), step_lds, (hipStream_t)stream, make_params(b),
                           (const float*)nullptr, g, 1, d_state_out, d_reward_out, d_done_out, d_actions_out);
    else
        hipLaunchKernelGGL(k_rlpso_step<true>, dim3(b->B), dim3(kThreads), step_lds, (hipStream_t)stream, make_params(b),
                           (const float*)nullptr, g, n_steps, d_state_out, d_reward_out, d_done_out, d_actions_out);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_gleet_policy(mbx_batch* b, const mbx_gleet_actor* net, const double* d_state, float* d_actions, float* d_mu_sigma,
                                void* stream)
{
    if (!b || !net || !net->d_weights || !d_state || !d_actions) return fail(MBX_E_ARG, "mbx_gleet_policy: bad arguments");
    if (b->cfg.algo != MBX_ALGO_GLEET) return fail(MBX_E_UNSUPPORTED, "mbx_gleet_policy: the batch is not a GLEET batch");
    if (net->n_floats != GpOff::total) return fail(MBX_E_ARG, "mbx_gleet_policy: expected %d weights, got %d", GpOff::total, net->n_floats);
    if (b->cfg.np > kGpThreads) return fail(MBX_E_UNSUPPORTED, "mbx_gleet_policy: np %d > %d", b->cfg.np, kGpThreads);
    const GleetActor g{net->d_weights, net->min_sigma, net->max_sigma};
    hipLaunchKernelGGL(k_gleet_policy, dim3(b->B), dim3(kGpThreads), gleet_policy_lds_bytes(b->cfg.np), (hipStream_t)stream, make_params(b),
                       g, d_state, d_actions, d_mu_sigma);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_qlpso_rollout(mbx_batch* b, const double* d_q_table, int n_steps, int32_t* d_actions_out, double* d_state_out,
                                 double* d_reward_out, uint8_t* d_done_out, void* stream)
{
    if (!b || !d_q_table) return fail(MBX_E_ARG, "mbx_qlpso_rollout: bad arguments");
    if (b->cfg.algo != MBX_ALGO_QLPSO) return fail(MBX_E_UNSUPPORTED, "mbx_qlpso_rollout: the batch is not a QLPSO batch");
    if (n_steps < 1) return fail(MBX_E_ARG, "mbx_qlpso_rollout: n_steps must be >= 1");
    if (b->d_tape && n_steps != 1) return fail(MBX_E_ARG, "mbx_qlpso_rollout: a replay tape holds one step");
    const size_t lds = (size_t)ql_lds_doubles(1, b->cfg.np, b->cfg.dim) * sizeof(double);
    if (n_steps == 1)
        hipLaunchKernelGGL(k_qlpso_step<false>, dim3(b->B), dim3(kThreads), lds, (hipStream_t)stream, make_params(b), (const int32_t*)nullptr,
                           d_q_table, 1, d_state_out, d_reward_out, d_done_out, d_actions_out);
    else
        hipLaunchKernelGGL(k_qlpso_step<true>, dim3(b->B), dim3(kThreads), lds, (hipStream_t)stream, make_params(b), (const int32_t*)nullptr,
                           d_q_table, n_steps, d_state_out, d_reward_out, d_done_out, d_actions_out);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_results(mbx_batch* b, double* d_cost_curves, double* d_fes, double* d_return, int32_t* d_steps,
                           int32_t* d_cost_len, void* stream)
{
    if (!b) return fail(MBX_E_ARG, "null batch");
    hipLaunchKernelGGL(k_results, dim3((b->B + 255) / 256), dim3(256), 0, (hipStream_t)stream, b->d_state, b->state_stride,
                       b->sc_off, b->B, b->cfg.n_logpoint, d_cost_curves, d_fes, d_return,
                       d_steps, d_cost_len);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

// Test / diagnostics: the device math routines the objectives are built from, applied element-wise.
__global__ void k_debug_math(int op, const double* __restrict__ x, const double* __restrict__ y, double* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y ? y[i] : 0.;
    double r;
    switch (op) {
    case 0: r = m_log(a); break;
    case 1: r = m_exp(a); break;
    case 2: r = m_sin(a); break;
    case 3: r = m_cos(a); break;
    case 4: r = m_pow(a, b); break;
    case 5: r = osc1(a); break;
    default: r = asy1(a, b); break;
    }
    out[i] = r;
}

// Test / diagnostics: the element-wise draws of RLEPSO's move phase for one (seed, generation, episode), converted exactly as rl_move converts them
// (mbx_rlepso.hpp; include/mbx_layout.h section 3): out[e] = {CLPSO uniform, FDR weight, tournament candidate 1, tournament candidate 2}.
__global__ void k_debug_rlepso_draws(uint64_t seed, uint32_t gen, uint32_t episode, int NP, int D, double* __restrict__ out)
{
    const int it = blockIdx.x * blockDim.x + threadIdx.x;          // element pair (2 it, 2 it + 1)
    if (2 * it >= NP * D) return;
    const Rng rng{(uint32_t)seed, (uint32_t)(seed >> 32), gen, episode};
    const U4 wa = rng.draw((uint32_t)it, MBX_SITE_ELEM_A), wt = rng.draw((uint32_t)it, MBX_SITE_TOURN);
    double* o = out + (int64_t)it * 8;
    o[0] = u32d(wa.x); o[1] = u32d(wa.z); o[2] = (double)__umulhi(wt.x, (uint32_t)NP); o[3] = (double)__umulhi(wt.y, (uint32_t)NP);
    if (2 * it + 1 < NP * D) { o[4] = u32d(wa.y); o[5] = u32d(wa.w); o[6] = (double)__umulhi(wt.z, (uint32_t)NP); o[7] = (double)__umulhi(wt.w, (uint32_t)NP); }
}

extern "C" int mbx_debug_rlepso_draws(uint64_t seed, int gen, int episode, int np, int dim, double* d_out, void* stream)
{
    if (np < 1 || dim < 1 || !d_out) return fail(MBX_E_ARG, "mbx_debug_rlepso_draws: bad arguments");
    const int pairs = (np * dim + 1) / 2;
    hipLaunchKernelGGL(k_debug_rlepso_draws, dim3((pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, seed, (uint32_t)gen, (uint32_t)episode, np, dim, d_out);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

__global__ void k_clock_probe(uint64_t* __restrict__ out, int n, int sleep_units)
{
    if (threadIdx.x != 0) return;
    for (int i = 0; i < n; ++i) {
        const uint64_t t = __builtin_amdgcn_s_memtime(), r = __builtin_amdgcn_s_memrealtime();
        out[2 * i] = t; out[2 * i + 1] = r;
        for (int k = 0; k < sleep_units; k += 127) __builtin_amdgcn_s_sleep(127);      // 127 x 64 clocks per instruction
    }
}

__global__ void k_clock_mark(uint64_t* __restrict__ out)
{
    if (threadIdx.x == 0) { out[0] = __builtin_amdgcn_s_memtime(); out[1] = __builtin_amdgcn_s_memrealtime(); }
}

extern "C" int mbx_debug_clock_probe(uint64_t* d_out, int n_samples, int sleep_units, void* stream)
{
    if (!d_out || n_samples < 2 || sleep_units < 1) return fail(MBX_E_ARG, "mbx_debug_clock_probe: bad arguments");
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out, n_samples, sleep_units);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_debug_clock_mark(uint64_t* d_out2, void* stream)
{
    if (!d_out2) return fail(MBX_E_ARG, "mbx_debug_clock_mark: bad arguments");
    hipLaunchKernelGGL(k_clock_mark, dim3(1), dim3(64), 0, (hipStream_t)stream, d_out2);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int mbx_debug_clock_slots(mbx_batch* b, uint64_t* d_slots2)
{
    if (!b) return fail(MBX_E_ARG, "mbx_debug_clock_slots: null batch");
    b->d_clk = (unsigned long long*)d_slots2;
    return MBX_OK;
}

extern "C" int mbx_batch_launch_info(const mbx_batch* b, int32_t out[4])
{
    if (!b || !out) return fail(MBX_E_ARG, "mbx_batch_launch_info: bad arguments");
    out[0] = b->threads; out[1] = (int32_t)b->lds_bytes; out[2] = b->fixed_geometry; out[3] = (int32_t)b->state_stride;
    if (b->cfg.algo == MBX_ALGO_LDE && b->fixed_geometry == 3) out[0] = MBX_LDE50_STEP_THREADS;       // k_lde_step's own workgroup size (k_lde_reset keeps b->threads)
    if (b->cfg.algo == MBX_ALGO_DEDDQN) {                      // the step kernel (small workgroups, one-row evaluator scratch), not k_dq_reset
        out[0] = kDqStepThreads; out[1] = (int32_t)(dq_lds_doubles(1, b->cfg.np, b->cfg.dim) * sizeof(double));
    }
    return MBX_OK;
}

extern "C" int mbx_debug_math(int op, const double* d_x, const double* d_y, double* d_out, int n, void* stream)
{
    if (op < 0 || op > 6 || !d_x || !d_out || n < 0 || ((op == 4 || op == 6) && !d_y)) return fail(MBX_E_ARG, "mbx_debug_math: bad arguments");
    if (n) hipLaunchKernelGGL(k_debug_math, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, op, d_x, d_y, d_out, n);
    HIP_TRY(hipGetLastError());
    return MBX_OK;
}

extern "C" int64_t mbx_instance_state_doubles(const mbx_batch* b)
{
    if (!b) return fail(MBX_E_ARG, "null batch");
    return geom_of(b->cfg).state_doubles;
}

extern "C" int mbx_debug_read_state(mbx_batch* b, int instance, double* host_out)
{
    if (!b || instance < 0 || instance >= b->B || !host_out) return fail(MBX_E_ARG, "mbx_debug_read_state: bad arguments");
    HIP_TRY(hipDeviceSynchronize());
    const int64_t n = mbx_instance_state_doubles(b);
    HIP_TRY(hipMemcpy(host_out, b->d_state + (int64_t)instance * b->state_stride, n * sizeof(double), hipMemcpyDeviceToHost));
    // cost curve is reported padded with its last value, like mbx_results
    double* sc = host_out + b->sc_off;
    const int len = (int)sc[MBX_SC_COST_LEN];
    for (int k = len; k <= b->cfg.n_logpoint && len > 0; ++k) sc[MBX_NSCALAR + k] = sc[MBX_NSCALAR + len - 1];
    return MBX_OK;
}

extern "C" int mbx_debug_write_state(mbx_batch* b, int instance, const double* host_in)
{
    if (!b || instance < 0 || instance >= b->B || !host_in) return fail(MBX_E_ARG, "mbx_debug_write_state: bad arguments");
    if (b->cfg.algo == MBX_ALGO_RLEPSO) {
        // the near-tie flag of the FDR scan bounds every denominator |p_jd - p_id| + 1e-5 by the box (csrc/mbx_rlepso.hpp: fdr_exact): pbest positions outside it -- which
        // reset / step never produce -- would void the exactness of the exemplar index, so such a block is refused
        const DevProblem& P = b->suite->h_problems[b->h_problem_idx[instance]];
        const double* pb = host_in + MBX_RLEPSO_ST_PBPOS(b->cfg.np, b->cfg.dim);
        for (int e = 0; e < b->cfg.np * b->cfg.dim; ++e)
            if (!(pb[e] >= P.lb && pb[e] <= P.ub))
                return fail(MBX_E_ARG, "mbx_debug_write_state: pbest position %d (%g) lies outside the problem's box [%g, %g]", e, pb[e], P.lb, P.ub);
    }
    HIP_TRY(hipDeviceSynchronize());
    const int64_t n = mbx_instance_state_doubles(b);
    HIP_TRY(hipMemcpy(b->d_state + (int64_t)instance * b->state_stride, host_in, n * sizeof(double), hipMemcpyHostToDevice));
    return MBX_OK;
}

extern "C" const char* mbx_last_error(void) { return g_err.c_str(); }
extern "C" const char* mbx_version(void) { return "metabox_amd libmbx 0.4 (gfx950; Philox stream layout 2: include/mbx_layout.h section 3; linear maps as fma chains)"; }

#ifdef MBX_PHASE_TIMING
// Instrumented builds only (not declared in include/mbx.h): cumulative per-phase cycles of k_rlepso_step (thread 0 of every
// block), summed over blocks.
extern "C" int mbx_debug_phase_cycles(unsigned long long* out, int n, int reset)
{
    std::vector<unsigned long long> h(8192 * 16);
    if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(mbx::g_phase_cycles), h.size() * 8) != hipSuccess) return -1;
    for (int i = 0; i < n && i < 16; ++i) { out[i] = 0; for (int b = 0; b < 8192; ++b) out[i] += h[b * 16 + i]; }
    if (reset) {
        std::fill(h.begin(), h.end(), 0ull);
        if (hipMemcpyToSymbol(HIP_SYMBOL(mbx::g_phase_cycles), h.data(), h.size() * 8) != hipSuccess) return -1;
    }
    return 0;
}
#endif
