/*
 * mbx.h — C-ABI of the MI355X MetaBBO rollout engine (libmbx.so).
 *
 * The reference (GMC-DRL/MetaBox) has no FFI: its hot path sits behind a duck-typed Python plugin
 * protocol.  This header is the drop-in boundary a maintainer binds with ctypes
 * to replace that path; every entry point names the reference interface it stands in for.
 *
 * Conventions
 *   - plain C types only; `stream` arguments are a hipStream_t passed as void* (NULL = default stream).
 *   - pointers named d_* are DEVICE pointers owned by the caller; all others are host pointers.
 *   - every function returns 0 on success or a negative MBX_E_* code; mbx_last_error() returns a
 *     thread-local message for the last failure.  No exceptions cross the ABI.
 *   - a handle is not thread-safe; distinct handles may be used from distinct threads.
 *   - launches are asynchronous on `stream`; the caller synchronises before reading d_* outputs.
 */
#ifndef MBX_H
#define MBX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MBX_OK            0
#define MBX_E_ARG        -1   /* invalid argument (reference raises ValueError / AssertionError)   */
#define MBX_E_HIP        -2   /* a HIP runtime call failed                                          */
#define MBX_E_UNSUPPORTED -3  /* valid request that this build does not implement (NotImplementedError) */
#define MBX_E_NOMEM      -4

/* ------------------------------------------------------------------------------------------------
 * Problem description — one entry per problem instance of a suite.
 * Replaces the per-instance attributes of the reference's problem objects:
 *   BBOB / noisy BBOB: src/problem/bbob.py:31-47 (dim, shift, rotate, bias, lb, ub, optimum) plus the
 *   per-class extras created in the constructors (bbob.py:233,250,294,314,360,423,589,609,634,694,
 *   744,768-794,847,873); noise mix-ins bbob.py:96-146.
 *   Protein docking: src/problem/protein_docking.py:9-26 (coor_init, q, e, r, basis, eigval).
 * `kind` selects the base objective: 1..24 = BBOB F1..F24 (noisy ids map onto their base kind),
 * MBX_KIND_PROTEIN = protein-docking energy.  Unused pointers are NULL.
 * ---------------------------------------------------------------------------------------------- */
#define MBX_KIND_PROTEIN 100

#define MBX_NOISE_NONE    0
#define MBX_NOISE_GAUSS   1   /* f*exp(a*N)                                   bbob.py:108-119 */
#define MBX_NOISE_UNIFORM 2   /* f*U^b*max(1,(1e9/(f+1e-99))^(a*(.49+1/D)*U'))  bbob.py:122-132 */
#define MBX_NOISE_CAUCHY  3   /* f+a*max(0,1e3+[U<b]*N/(|N'|+1e-199))         bbob.py:135-146 */

typedef struct mbx_problem_desc {
    int32_t func_id;      /* 1..24, 101..130, or 0 for protein                                   */
    int32_t kind;         /* base objective, see above                                           */
    int32_t dim;
    int32_t n_peaks;      /* Gallagher: 101 or 21; protein: number of atoms (100)                */
    int32_t noise_kind;   /* MBX_NOISE_*                                                         */
    int32_t reserved;
    double  bias, lb, ub;
    double  pen_coef;     /* coefficient of boundaryHandling(x) = pen_coef * sum(max(0,|x|-ub)^2) */
    double  s[4];         /* kind-specific scalars (metabox_amd/problem/bbob.py: BBOB_Problem.desc) */
    double  noise_a, noise_b;
    const double* dshift; /* [dim]      subtracted from x before m1                              */
    const double* m1;     /* [dim,dim]  first linear map, row-major (z_i = sum_k m1[i,k] y_k)    */
    const double* m2;     /* [dim,dim]  second linear map or NULL                                */
    const double* v0;     /* [dim]      per-dimension constants or NULL                          */
    const double* v1;
    const double* v2;
    const double* py;     /* Gallagher peaks  [n_peaks,dim]; protein: basis/sqrt(eigval) [dim,3*n] */
    const double* pc;     /* Gallagher C      [n_peaks,dim]; protein: coor_init [n,3]            */
    const double* pw;     /* Gallagher w      [n_peaks];     protein: sqrt(e)|q|r  [3,n,n]       */
} mbx_problem_desc;

typedef struct mbx_suite mbx_suite;
typedef struct mbx_batch mbx_batch;

/* Upload a problem set (reference: construct_problem_set, src/utils.py:4-27 -> the list of problem
 * objects every optimizer evaluates against).  Also evaluates optimum_i = f_i(opt_i) on the device
 * like BBOB_Basic_Problem.__init__ (bbob.py:42); opt == NULL entries (protein) get optimum = NaN,
 * the ABI's spelling of the reference's `optimum = None`.  `opt` is [n_problems][dim] or NULL. */
int mbx_suite_create(const mbx_problem_desc* descs, int n_problems, const double* opt, mbx_suite** out);
int mbx_suite_destroy(mbx_suite* s);
int mbx_suite_size(const mbx_suite* s);
/* host copy of optimum_i (problem.optimum; NaN = None) */
int mbx_suite_optimum(const mbx_suite* s, double* optimum_out /* [n_problems] */);
/* Protein docking: the number of atom pairs i < j the energy kernel walks for a candidate inside the box -- the pairs that can reach the 9 A cut-off (the rest of the 4950
 * are provably beyond it: an atom moves by at most ub |sum_k v0_k |basis_k||_2).  What the inter-rank partition weights a protein problem with
 * (metabox_amd/problem/protein_docking.py: close_pairs computes the same number on the host; tests/test_protein.py compares the two).  -1 for a BBOB problem. */
int mbx_suite_close_pairs(const mbx_suite* s, int problem);

/* Stand-alone objective evaluation: Basic_Problem.eval / F*.func (src/problem/basic_problem.py:12-34).
 * d_x is [n, dim] row-major, d_f is [n]; the bias is included, the optimum is NOT subtracted.
 * noisy != 0 applies the noise model like NoisyProblem.eval (bbob.py:100-102) with Philox draws keyed
 * by (seed, row); with d_noise_draws != NULL ([3, n]: the model's draws in the reference's call order)
 * those values are used instead of Philox (replay of recorded numpy draws). */
int mbx_eval(mbx_suite* s, int problem, const double* d_x, int n, double* d_f,
             int noisy, uint64_t seed, const double* d_noise_draws, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Lock-step batch of independent (problem x run) optimizer instances.
 * ---------------------------------------------------------------------------------------------- */
#define MBX_ALGO_RLEPSO 1   /* src/optimizer/rlepso_optimizer.py   one step = one generation       */
#define MBX_ALGO_LDE    2   /* src/optimizer/lde_optimizer.py      one step = one generation       */
#define MBX_ALGO_DEDDQN 3   /* src/optimizer/de_ddqn_optimizer.py  one step = one trial vector     */
#define MBX_ALGO_RANDOM_SEARCH 4 /* src/optimizer/random_search.py  one step = NP uniform samples   */
#define MBX_ALGO_RLPSO  5   /* src/optimizer/rl_pso_optimizer.py   one step = one particle         */
#define MBX_ALGO_GLEET  6   /* src/optimizer/gleet_optimizer.py    one step = one generation       */
#define MBX_ALGO_QLPSO  7   /* src/optimizer/qlpso_optimizer.py    one step = one particle         */
#define MBX_ALGO_DE     8   /* src/optimizer/deap_de.py            one step = one sweep (np trials)   -- classic baseline, no agent */
#define MBX_ALGO_PSO    9   /* src/optimizer/deap_pso.py           one step = one sweep (np moves)    -- classic baseline, no agent */
#define MBX_ALGO_CMAES  10  /* src/optimizer/deap_cmaes.py         one step = one generation          -- classic baseline, no agent */

typedef struct mbx_algo_cfg {
    int32_t algo;          /* MBX_ALGO_*                                                          */
    int32_t np;            /* population size: reference hard-codes 100 (RLEPSO, rlepso_optimizer.py:11),
                              50 (LDE, lde_optimizer.py:10), 100 (DE-DDQN, de_ddqn_optimizer.py:12) */
    int32_t dim;
    int32_t max_fes;       /* config.maxFEs       (src/config.py:74,88)                           */
    int32_t log_interval;  /* config.log_interval (src/config.py:102)                             */
    int32_t n_logpoint;    /* config.n_logpoint   (src/config.py:77,90)                           */
    int32_t early_stop;    /* 1 = reference rule `done = fes>=maxFEs or gbest<=1e-8`; 0 = fixed horizon */
    int32_t n_group;       /* RLEPSO: 5 (rlepso_optimizer.py:26)                                  */
    uint32_t flags;        /* MBX_F_* below, OR-ed; 0 = the defaults.  Per batch: two batches of one process may differ. */
} mbx_algo_cfg;

/* Per-batch behaviour options (mbx_algo_cfg.flags).  They select among kernels that compute the SAME update(); none changes a result except
 * MBX_F_FDR_FAST, and that one only where two FDR candidates have quotients within two ulp of each other.
 *   MBX_F_FDR_FAST               RLEPSO: the FDR exemplar (rlepso_optimizer.py:97-109) by cross-multiplied comparison alone.  Default (flag clear): the scan also
 *                                detects every comparison that comes near a tie and settles those items with the reference's ROUNDED quotients and np.argmin's
 *                                first-index rule -- the exemplar index is the reference's on any input (tests/test_fdr_ties.py) at +3..5 % of a generation.
 *   MBX_F_GENERIC_GEOMETRY       keep the run-time-geometry kernels where a compile-time-geometry instantiation exists (the tests compare the two bit for bit)
 *   MBX_F_ROLLOUT_PER_GENERATION the mbx_*_rollout entry points step one launch per generation instead of the resident kernel (same outputs)
 * Test override: the environment variables MBX_FDR_FAST=1, MBX_GENERIC_GEOMETRY=1, MBX_ROLLOUT_PER_GENERATION=1 are OR-ed into the flags of every batch
 * created while they are set (read once, inside mbx_batch_create; never at call time). */
#define MBX_F_FDR_FAST               1u
#define MBX_F_GENERIC_GEOMETRY       2u
#define MBX_F_ROLLOUT_PER_GENERATION 4u
/* the flags a batch was created with, environment overrides included (negative = MBX_E_*) */
int mbx_batch_flags(const mbx_batch* b);

/* Dimensions of the per-step tensors for a configuration (so callers can size buffers):
 *   RLEPSO        : state [1]      (fes/maxFEs, rlepso_optimizer.py:170-171), action [35] float32
 *   LDE           : state [np+10]  (lde_optimizer.py:145-157),               action [2*np] float32
 *   DEDDQN        : state [99]     (de_ddqn_optimizer.py:76-129),            action [1] int32
 *   RANDOM_SEARCH : state [1]      (fes/maxFEs),                             no action (pass NULL to mbx_step)
 *   RLPSO         : state [2*dim]  (rl_pso_optimizer.py:62-63),             action [1] float32
 *   GLEET         : state [np*27]  (gleet_optimizer.py:111-124),            action [np] float32
 *   QLPSO         : state [1]      (qlpso_optimizer.py:89-90,125),          action [1] int32 in {0..3}
 *   DE, PSO, CMAES: state [1]      (fes/maxFEs),                             no action (pass NULL to mbx_step) */
int mbx_state_dim(const mbx_algo_cfg* cfg);
int mbx_action_dim(const mbx_algo_cfg* cfg);
/* number of doubles of external random numbers one instance consumes per step (see mbx_set_tape) */
int64_t mbx_tape_stride(const mbx_algo_cfg* cfg);

/* Create B instances.  problem_idx[i] indexes the suite, seeds[i] is the Philox key of instance i
 * (results depend only on (problem, seed), never on batch position or GPU count).
 * Replaces the (problem x run) enumeration of Tester.test / rollout (src/tester.py:190-202,317-328)
 * and the `PBO_Env(problem, optimizer)` construction per pair. */
int mbx_batch_create(mbx_suite* s, const mbx_algo_cfg* cfg, const int32_t* problem_idx,
                     const uint64_t* seeds, int n_instances, mbx_batch** out);
int mbx_batch_destroy(mbx_batch* b);

/* Replace the Philox source by caller-supplied random numbers for the NEXT mbx_reset / mbx_step:
 * d_tape is [n_instances, mbx_tape_stride()] doubles laid out as documented in
 * include/mbx_layout.h (one slot per draw site of the reference's update()).  NULL restores
 * Philox.  This is the replay hook that lets recorded numpy draws drive the kernel. */
int mbx_set_tape(mbx_batch* b, const double* d_tape);

/* PBO_Env.reset() for every instance: problem.reset(); optimizer.init_population(problem)
 * (src/environment/basic_environment.py:17-19, rlepso_optimizer.py:39-65).  Writes the initial state
 * [n_instances, state_dim] (float64). */
int mbx_reset(mbx_batch* b, double* d_state_out, void* stream);

/* PBO_Env.step(action) for every instance that is not done: optimizer.update(action, problem)
 * (basic_environment.py:21-22, rlepso_optimizer.py:173-263).  d_actions is
 * [n_instances, action_dim] (float32; int32 for DEDDQN; NULL for RANDOM_SEARCH).  Outputs: next state (float64),
 * reward (float64), done (uint8).  Done instances are left untouched and report reward 0 (their state row keeps
 * its last value).  Workgroups are dispatched most-expensive-objective first.  *
 * RLEPSO, FDR exemplar (src/optimizer/rlepso_optimizer.py:97-109): the reference takes np.argmin of ROUNDED quotients.  Every RLEPSO kernel -- this one, the
 * compile-time-geometry instantiations and the resident rollout kernels -- scans by cross-multiplication, flags each comparison that comes within 2^-49 of a
 * tie and settles the flagged items with the reference's rounded quotients and first-index rule: the exemplar index equals the reference's on every input
 * whose pbest positions lie inside [lb, ub] (which reset / step guarantee; a block injected with mbx_debug_write_state must keep it).  MBX_F_FDR_FAST in
 * mbx_algo_cfg.flags drops the flag and the second pass (differs from the reference only where two quotients are <= 1 ulp apart: tests/test_fdr_ties.py). */
int mbx_step(mbx_batch* b, const void* d_actions, double* d_state_out, double* d_reward_out,
             uint8_t* d_done_out, void* stream);

/* Per-instance results, the fields rollout_episode returns (src/agent/rlepso_agent.py:294-303):
 * cost curve `optimizer.cost` padded to n_logpoint+1 entries with its last value
 * (src/tester.py:204-205), fes, return (sum of rewards), executed env-steps, and the live length of the
 * cost list.  Any pointer may be NULL. */
int mbx_results(mbx_batch* b, double* d_cost_curves /* [B, n_logpoint+1] */, double* d_fes /* [B] */,
                double* d_return /* [B] */, int32_t* d_steps /* [B] */, int32_t* d_cost_len /* [B] */,
                void* stream);

/* Give an existing batch new (problem, seed) pairs without re-allocating anything: the next mbx_reset starts episode 0 of the new
 * instances.  This is what the reference does when PBO_Env.reset() is called for another problem / run with the same optimizer object
 * (src/environment/basic_environment.py:17-19 -> init_population).  Synchronises the device. */
int mbx_batch_rebind(mbx_batch* b, const int32_t* problem_idx /* [B] */, const uint64_t* seeds /* [B] */);

/* The public attributes the reference's optimizers expose after every update() -- optimizer.fes, .cost, .log_index
 * (src/optimizer/rlepso_optimizer.py:27-30, 241-261) -- for ONE instance: copies the instance's scalar block and cost list,
 * host_out[0 .. MBX_NSCALAR + n_logpoint] (layout: MBX_SC_* in include/mbx_layout.h), with one small device-to-host copy on `stream`
 * and waits for it.  The single-instance compatibility view (B = 1) calls it once per step instead of reading the whole state back. */
int mbx_read_public(mbx_batch* b, int instance, double* host_out /* [MBX_NSCALAR + n_logpoint + 1] */, void* stream);

/* Test / diagnostics: copy the internal per-instance optimizer state to the host.
 * RLEPSO layout: see MBX_RLEPSO_* offsets in include/mbx_layout.h. */
int64_t mbx_instance_state_doubles(const mbx_batch* b);
int mbx_debug_read_state(mbx_batch* b, int instance, double* host_out);
/* The inverse: overwrite one instance's state block from the host (host_in: mbx_instance_state_doubles doubles, same layout).  What a caller of the
 * reference does with `copy.deepcopy(env)` / a pickled optimizer (src/tester.py, src/agent/utils.py:44-48 save_class): snapshot an instance and
 * resume it later -- and what the tests use to hand the generation kernels a crafted swarm (tests/test_fdr_ties.py).  Synchronises the device.
 * RLEPSO: a block whose pbest positions leave the problem's box [lb, ub] is refused (MBX_E_ARG): reset / step never produce one, and the FDR scan's near-tie flag relies on it. */
int mbx_debug_write_state(mbx_batch* b, int instance, const double* host_in);

/* The RLEPSO / RL-PSO actor as ONE kernel launch per step (src/agent/rlepso_agent.py:9-47 Actor.forward without
 * fixed_action; src/agent/rl_pso_agent.py:9-47 PolicyNetwork.forward): two MLPs in_dim -> h1 -> h2 -> out_dim (ReLU, ReLU, none) sharing their input,
 *   mu = (tanh(mu_net(x)) + 1)/2,  sigma = (tanh(sigma_net(x)) + 1)/2 * (max_sigma - min_sigma) + min_sigma,
 *   action = clamp(Normal(mu, sigma).sample(), 0, 1).
 * float32 arithmetic like the reference's torch modules.  d_weights holds, for the mu net and then for the sigma net:
 *   W1^T [in_dim][h1] | b1 [h1] | W2^T [h1][h2] | b2 [h2] | W3^T [h2][out_dim] | b3 [out_dim]
 * (W^T = torch's nn.Linear.weight transposed, row-major).  The Normal draws replace torch's global generator by the
 * instance's Philox stream: counter (j, MBX_SITE_POLICY, next generation, episode). */
#define MBX_POLICY_RLEPSO 0   /* sigma affine in tanh, action clamped to [0,1]        (rlepso_agent.py:24-32) */
#define MBX_POLICY_RLPSO  1   /* sigma clamped to [min,max], out-of-range samples re-folded (rl_pso_agent.py:24-35) */
typedef struct mbx_gauss_mlp {
    const float* d_weights;
    int32_t in_dim, h1, h2, out_dim;
    float min_sigma, max_sigma;
    int32_t variant;              /* MBX_POLICY_* */
} mbx_gauss_mlp;

/* d_state [n_instances, in_dim] float64 (what mbx_reset / mbx_step wrote) -> d_actions [n_instances, out_dim] float32,
 * ready for mbx_step.  d_mu_sigma, if not NULL, receives [n_instances, 2, out_dim] float32 (mu row, sigma row). */
int mbx_gauss_policy(mbx_batch* b, const mbx_gauss_mlp* net, const double* d_state, float* d_actions, float* d_mu_sigma,
                      void* stream);

/* LDE's PolicyNet as ONE kernel launch per generation (src/agent/lde_agent.py:8-29 PolicyNet.forward / sampler, :147-163 the rollout loop
 * body): one LSTM cell in_dim -> hidden (torch.nn.LSTM gate order i, f, g, o), mu = Linear(h'), sigma = sigmoid(Linear(h')),
 * action = clip(Normal(mu, sigma).sample(), 0, 1).  d_weights (float32), every matrix TRANSPOSED so that consecutive output units are
 * consecutive words:  W_ih^T [in_dim][4 hidden] | W_hh^T [hidden][4 hidden] | b_ih + b_hh [4 hidden] | W_mu^T [hidden][out_dim] |
 * W_sigma^T [hidden][out_dim] | b_mu [out_dim] | b_sigma [out_dim].
 * d_state [n_instances, in_dim] float64 (what mbx_reset / mbx_step wrote); d_h, d_c [n_instances, hidden] float32 are read and
 * overwritten with (h', c'); d_actions [n_instances, out_dim] float32 (may be NULL: no sampling), ready for mbx_step; d_mu_sigma, if not
 * NULL, receives [n_instances, 2, out_dim].  Philox draws (j, MBX_SITE_POLICY, gen + 1, episode) of the instance's own stream. */
typedef struct mbx_lstm_policy {
    const float* d_weights;
    int32_t in_dim, hidden, out_dim;
} mbx_lstm_policy;
int mbx_lde_policy(mbx_batch* b, const mbx_lstm_policy* net, const double* d_state, float* d_h, float* d_c, float* d_actions,
                   float* d_mu_sigma, void* stream);

/* DE-DDQN's greedy action for the whole batch in ONE launch (src/agent/de_ddqn_agent.py:59-68, 108-117: `action = argmax Q(state)`;
 * Q = the ReLU MLP of :26-36 / src/agent/networks.py:4-26, in_dim -> width x depth -> n_act) on the float32 matrix cores
 * (v_mfma_f32_16x16x4_f32: one float32 fma chain per unit, k ascending, starting at the bias).  Only the reference's architecture is
 * built: in_dim 99, width 100, depth 4, n_act 4 (MBX_E_UNSUPPORTED otherwise; callers keep their PyTorch route for other shapes).
 * d_weights (float32): for every layer of the torch module in order, the weight TRANSPOSED, Wt [in][out] row-major, then the bias [out].
 * d_state [n_instances, in_dim] float64 (what mbx_reset / mbx_step wrote) -> d_actions [n_instances] int32, ready for mbx_step (first
 * maximum, like torch.argmax); d_q, if not NULL, receives the Q values [n_instances, n_act] float32. */
typedef struct mbx_qnet {
    const float* d_weights;
    int32_t in_dim, width, depth, n_act;
} mbx_qnet;
int mbx_ddqn_qnet(mbx_batch* b, const mbx_qnet* net, const double* d_state, int32_t* d_actions, float* d_q, void* stream);

/* The agent's act() and the environment's step() in ONE launch (the loop body of RLEPSO_Agent.rollout_episode,
 * src/agent/rlepso_agent.py:294-303: `action = actor(state); state, reward, done = env.step(action)`).
 * RLEPSO's state is the scalar fes/maxFEs (rlepso_optimizer.py:170-171) and fes is an integer, so the actor's (mu, sigma)
 * take at most mbx_rlepso_policy_table_rows() distinct values.  mbx_rlepso_policy_table evaluates the actor once at all
 * of them (row k <- state k/maxFEs; same kernel and float32 arithmetic as mbx_gauss_policy) into
 * d_table [rows, 2, out_dim] float32; it has to be rebuilt whenever the weights change.  mbx_rlepso_act_step then draws
 * each instance's action from row `fes` with exactly the Philox draws of mbx_gauss_policy and performs mbx_step with
 * it: mbx_gauss_policy + mbx_step and mbx_rlepso_act_step give bit-identical trajectories.  d_actions_out, if not
 * NULL, receives the sampled actions [n_instances, out_dim] (e.g. for log-probabilities). */
int mbx_rlepso_policy_table_rows(const mbx_batch* b);
int mbx_rlepso_policy_table(mbx_batch* b, const mbx_gauss_mlp* net, float* d_table, void* stream);
int mbx_rlepso_act_step(mbx_batch* b, const float* d_table, float* d_actions_out, double* d_state_out,
                        double* d_reward_out, uint8_t* d_done_out, void* stream);

/* The whole loop of RLEPSO_Agent.rollout_episode (src/agent/rlepso_agent.py:294-303: `while not is_done: action = actor(state);
 * state, reward, is_done = env.step(action)`), up to n_gens generations of every instance in ONE launch, with the instance's state
 * ON CHIP between generations: with the actor inside the kernel (d_table from mbx_rlepso_policy_table) nothing has to leave the
 * workgroup between two generations, so the state block (include/mbx_layout.h section 2) is read once, kept in LDS / registers and
 * written once per launch instead of once per generation.  Every generation does exactly what mbx_rlepso_act_step does, with the
 * same Philox counters: n_gens calls of mbx_rlepso_act_step and one call of mbx_rlepso_rollout leave bit-identical states, cost
 * curves and returns.  A workgroup leaves the launch when its instance terminates (rlepso_optimizer.py:246-249) and its slot goes
 * to the next workgroup of the grid, so instances that finish early cost nothing afterwards.
 * Per-generation records for training (rlepso_agent.py:143-190 collects exactly these), each may be NULL:
 *   d_traj_actions [n_gens, n_instances, out_dim] float32  sampled actions (rows of generations after termination are not written)
 *   d_traj_state   [n_gens, n_instances] float64           state AFTER the generation (fes / maxFEs)
 *   d_traj_reward  [n_gens, n_instances] float64           reward (0 after termination)
 *   d_traj_done    [n_gens, n_instances] uint8             is_done (1 after termination)
 * d_state_out / d_done_out [n_instances]: state / is_done after the last executed generation; d_reward_out [n_instances]: SUM of
 * the rewards of the executed generations (like mbx_rlpso_rollout).
 * The compile-time geometries (NP 100 at D 10 / D 12 = protein docking / D 30 / D 40, and NP 128 / D 40, 5 groups) run the resident kernel; any other geometry is stepped
 * with one mbx_rlepso_act_step launch per generation behind the same interface (MBX_F_ROLLOUT_PER_GENERATION in the batch's flags
 * forces that route).  mbx_rlepso_rollout_resident tells which route a batch takes: 1 = one resident launch per call,
 * 0 = one launch per generation (2 n_gens launches with d_reward_out).  Neither route allocates or reads the environment at call time. */
int mbx_rlepso_rollout_resident(const mbx_batch* b);
int mbx_rlepso_rollout(mbx_batch* b, const float* d_table, int n_gens, float* d_traj_actions, double* d_traj_state,
                       double* d_traj_reward, uint8_t* d_traj_done, double* d_state_out, double* d_reward_out,
                       uint8_t* d_done_out, void* stream);

/* The whole loop of LDE_Agent.rollout_episode (src/agent/lde_agent.py:147-163: `while not is_done: action, h, c = net.sampler(state, h, c);
 * state, reward, is_done = env.step(action)`) with update() of src/optimizer/lde_optimizer.py:159-198, up to n_gens generations of every
 * instance in ONE launch: population, fitness order, features and the LSTM's (h, c) stay in LDS between generations, the PolicyNet
 * (lde_agent.py:8-29) is evaluated inside the workgroup from the packed weights of `net` (mbx_lstm_policy), and the state block is read once
 * and written once per launch.  Every generation does exactly what mbx_lde_policy followed by mbx_step does, with the same Philox counters:
 * n_gens such pairs and one call of mbx_lde_rollout leave bit-identical state blocks, (h, c), features, rewards and cost curves.
 *   d_state_in  [n_instances, NP + 10] float64   the features the last reset / step / rollout produced (may be d_state_out itself)
 *   d_h, d_c    [n_instances, hidden] float32    LSTM state, updated in place (not touched for instances that were done before the call)
 * Per-generation records, each may be NULL:
 *   d_traj_actions [n_gens, n_instances, 2 NP] float32   sampled actions (rows of generations after termination are not written)
 *   d_traj_state   [n_gens, n_instances, NP + 10] float64   features AFTER the generation (not written after termination)
 *   d_traj_reward  [n_gens, n_instances] float64   reward (0 after termination);   d_traj_done [n_gens, n_instances] uint8 (1 after termination)
 * d_state_out [n_instances, NP + 10] / d_done_out: features / is_done after the last executed generation (rows of instances that were done
 * before the call are not written); d_reward_out: SUM of the rewards of the executed generations.
 * The resident kernel is built for NP 50 at D 10 and D 30 (all 24 BBOB kinds and the noisy suite; at D 30 the batches that hold F3 / F4 / F5 / F15 / F20 / F24 run an
 * instantiation with a second tile array, the others the lean one config 3 is timed on) and for NP 100 at D 30 (config 3 as written: all of bbob-noisy; bbob without those six
 * kinds), hidden 50; any other batch is stepped with mbx_lde_policy + mbx_step per generation behind the same interface (MBX_F_ROLLOUT_PER_GENERATION in the batch's flags forces
 * that route).  mbx_lde_rollout_resident: 1 / 0. */
int mbx_lde_rollout_resident(const mbx_batch* b);
int mbx_lde_rollout(mbx_batch* b, const mbx_lstm_policy* net, const double* d_state_in, float* d_h, float* d_c, int n_gens,
                    float* d_traj_actions, double* d_traj_state, double* d_traj_reward, uint8_t* d_traj_done, double* d_state_out,
                    double* d_reward_out, uint8_t* d_done_out, void* stream);

/* RL-PSO moves ONE particle per env step, so a rollout is maxFEs - NP steps of D-element arithmetic plus one evaluation:
 * launch latency, not work.  mbx_rlpso_rollout runs `n_steps` consecutive steps of every instance in ONE launch with the
 * actor evaluated inside the kernel (the loop of RL_PSO_Agent.rollout_episode, src/agent/rl_pso_agent.py:112-124).  It is
 * bit-identical to n_steps x (mbx_gauss_policy + mbx_step).  d_reward_out receives the SUM of the rewards of the steps
 * executed by this call, d_state_out / d_done_out the state and flag after the last one; d_actions_out, if not NULL,
 * the last sampled action [n_instances]. */
int mbx_rlpso_rollout(mbx_batch* b, const mbx_gauss_mlp* net, int n_steps, float* d_actions_out, double* d_state_out,
                      double* d_reward_out, uint8_t* d_done_out, void* stream);

/* GLEET's attention actor (src/agent/gleet_agent.py:314-444, Actor.forward without fixed_action) for every swarm of the batch in
 * ONE launch: embedding, encoder layer, memory-query decoder layer (4 heads of 4, swarm-wide normalisation, FF 16), the two
 * 16-32-8-1 heads, squashing, Normal sample and clamp.  The architecture is the reference's fixed configuration (:31-45);
 * d_weights is the actor's state_dict flattened in its own order (5426 float32: embedder, encoder.0 {W_query, W_key, W_val,
 * W_out, FF.0.weight, FF.0.bias, FF.2.weight, FF.2.bias}, embedder_for_decoder, decoder.0 {...}, mu_net, sigma_net).
 * d_state [n_instances, np, 27] float64 (what mbx_reset / mbx_step wrote) -> d_actions [n_instances, np] float32;
 * d_mu_sigma, if not NULL, receives [n_instances, 2, np].  float32 arithmetic; Philox draws (i, MBX_SITE_POLICY, gen + 1, episode). */
typedef struct mbx_gleet_actor {
    const float* d_weights;
    int32_t n_floats;             /* must be 5426 */
    float min_sigma, max_sigma;
} mbx_gleet_actor;
int mbx_gleet_policy(mbx_batch* b, const mbx_gleet_actor* net, const double* d_state, float* d_actions, float* d_mu_sigma, void* stream);

/* QLPSO with its tabular policy inside the step kernel, `n_steps` env steps per launch: the loop of QLPSO_Agent.rollout_episode
 * (src/agent/qlpso_agent.py:66-75) with __get_action (:35-38: softmax over the Q-row of the state, np.random.choice) evaluated on
 * the device.  d_q_table is [4, 4] float64 (states x actions).  With a replay tape (n_steps = 1) the choice uniform comes from
 * the tape, so the reference's own decisions are reproduced.  Outputs as mbx_rlpso_rollout; d_actions_out [n_instances] int32. */
int mbx_qlpso_rollout(mbx_batch* b, const double* d_q_table, int n_steps, int32_t* d_actions_out, double* d_state_out,
                      double* d_reward_out, uint8_t* d_done_out, void* stream);

/* Test / diagnostics: apply one of the device math routines the objectives are built from to n device values.
 * op: 0 log, 1 exp, 2 sin, 3 cos, 4 pow(x, y), 5 T_osz(x) (bbob.py:51-67), 6 T_asy(x; beta_lin = y) (bbob.py:70-82). */
int mbx_debug_math(int op, const double* d_x, const double* d_y, double* d_out, int n, void* stream);

/* Test / diagnostics: the element-wise Philox draws of RLEPSO's move phase (include/mbx_layout.h section 3: sites ELEM_A and TOURN) of the instance with
 * key `seed` at generation `gen` of episode `episode`, converted exactly as the generation kernel converts them: d_out [np * dim, 4] float64 =
 * {CLPSO uniform in [0, 1), FDR weight in [0, 1), tournament candidate 1, tournament candidate 2 (integers in [0, np))} per element
 * (replaces np.random.rand(NP, D) x 2 and np.random.randint(0, NP, (NP, D, 2)), src/optimizer/rlepso_optimizer.py:76-109).  The statistical tests of
 * the stream (tests/test_gpu_rlepso.py::test_move_phase_draws_are_uniform) read it; they do not depend on the oracle. */
int mbx_debug_rlepso_draws(uint64_t seed, int gen, int episode, int np, int dim, double* d_out, void* stream);

/* Diagnostics: how the generation kernel of this batch is launched.  out[0] = threads per workgroup, out[1] = dynamic LDS bytes per
 * workgroup, out[2] = compile-time-geometry instantiation in use (0 = run-time geometry), out[3] = stride, in doubles,
 * between the state blocks of consecutive instances (>= mbx_instance_state_doubles).  Host-only, no device work. */
int mbx_batch_launch_info(const mbx_batch* b, int32_t out[4]);

/* Measurement: the shader clock DURING somebody else's kernel.  One wave takes `n_samples` samples, `sleep_units` x 64 clocks apart (s_sleep), of the
 * shader's cycle counter (s_memtime) and of the constant 100 MHz real-time counter (s_memrealtime): d_out [n_samples, 2] uint64.  Launched on a side stream
 * next to a timed window, the ratio of the two differences is the clock the chip ran at in that window WITHOUT a profiler attached (bench.py:
 * roofline.valu.clock_ghz; rocprofv3's counter passes lower the clock).  mbx_debug_clock_mark writes one (s_memtime, s_memrealtime) pair: two marks around a
 * launch on its own stream bracket it in the same time base. */
int mbx_debug_clock_probe(uint64_t* d_out, int n_samples, int sleep_units, void* stream);
int mbx_debug_clock_mark(uint64_t* d_out2, void* stream);
/* Measurement: the shader clock a resident RLEPSO launch runs at, measured by the launch ITSELF.  While a slot pair is attached, thread 0 of every workgroup of
 * k_rlepso_run reads s_memtime (shader cycles) and s_memrealtime (100 MHz) at its start and end and adds the two differences to d_slots2[0] / d_slots2[1] (the caller
 * zeroes them): clock [GHz] = d_slots2[0] / d_slots2[1] / 10, averaged over the workgroups' lifetimes -- no cross-CU counter arithmetic (the s_memtime counters of
 * different XCDs are offset against each other), no probe wave beside the launch, no profiler.  One scalar load per workgroup when nothing is attached.  NULL detaches.
 * Host-only call, takes effect at the next mbx_rlepso_rollout.  (bench.py: roofline.valu.clock_ghz / shader_cycles_per_generation of the timed windows themselves.) */
int mbx_debug_clock_slots(mbx_batch* b, uint64_t* d_slots2);

const char* mbx_last_error(void);
/* "metabox_amd libmbx <major.minor> (gfx950; Philox stream layout <n>: ...)".  The stream layout number changes whenever the assignment of Philox
 * counters to draws changes (include/mbx_layout.h section 3): trajectories, tapes and golden files made under another layout number do not
 * reproduce and have to be regenerated.  Layout 1 = round 1 (53-bit element-wise uniforms, one call per element); layout 2 = since round 2
 * (32-bit element-wise uniforms, one ELEM_A call per element pair, tournament draws on their own site TOURN). */
const char* mbx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MBX_H */
