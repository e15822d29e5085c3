/*
 * mbx_layout.h — data layouts that are part of the C-ABI contract (shared by libmbx.so, its host
 * mirror and the test oracle; it defines formats only, no algorithm).
 *
 *  1. the per-step "tape" of external random numbers (mbx_set_tape),
 *  2. the per-instance optimizer state exposed by mbx_debug_read_state,
 *  3. the Philox4x32-10 draw-site map (which counter produces which random number).
 *
 * All sizes are in doubles; NP = population size, D = dimension.
 */
#ifndef MBX_LAYOUT_H
#define MBX_LAYOUT_H

#include <stdint.h>

/* ---------------------------------------------------------------- 1. RLEPSO tape (per instance, per step)
 * Slots follow the draw order of RLEPSO_Optimizer.update (reference: rlepso_optimizer.py:179-180,
 * 77,88,108, eval noise, 238, 137-138, eval noise):
 *   rand1[NP] rand2[NP] clpso_u[NP*D] tourn_idx[NP*D*2] fdr_u[NP*D] noise_main[3*NP]
 *   reinit_u[NP] reinit_pos_u[NP*D] reinit_vel_u[NP*D] noise_reinit[3*NP]
 * Every value is the raw draw (U[0,1), integer index, or N(0,1)); the kernel applies low+(high-low)*u.
 * mbx_reset (init_population, rlepso_optimizer.py:40-42) uses the reinit_pos_u / reinit_vel_u /
 * noise_reinit slots.  The three noise rows hold the noise model's draws in the reference's call
 * order (gauss: N; uniform: U,U'; cauchy: U,N,N').                                                    */
#define MBX_RLEPSO_TAPE_RAND1(NP, D)      ((int64_t)0)
#define MBX_RLEPSO_TAPE_RAND2(NP, D)      ((int64_t)(NP))
#define MBX_RLEPSO_TAPE_CLPSO(NP, D)      ((int64_t)2 * (NP))
#define MBX_RLEPSO_TAPE_TOURN(NP, D)      ((int64_t)2 * (NP) + (int64_t)(NP) * (D))
#define MBX_RLEPSO_TAPE_FDR(NP, D)        ((int64_t)2 * (NP) + (int64_t)3 * (NP) * (D))
#define MBX_RLEPSO_TAPE_NOISE0(NP, D)     ((int64_t)2 * (NP) + (int64_t)4 * (NP) * (D))
#define MBX_RLEPSO_TAPE_REINIT(NP, D)     ((int64_t)5 * (NP) + (int64_t)4 * (NP) * (D))
#define MBX_RLEPSO_TAPE_REPOS(NP, D)      ((int64_t)6 * (NP) + (int64_t)4 * (NP) * (D))
#define MBX_RLEPSO_TAPE_REVEL(NP, D)      ((int64_t)6 * (NP) + (int64_t)5 * (NP) * (D))
#define MBX_RLEPSO_TAPE_NOISE1(NP, D)     ((int64_t)6 * (NP) + (int64_t)6 * (NP) * (D))
#define MBX_RLEPSO_TAPE_STRIDE(NP, D)     ((int64_t)9 * (NP) + (int64_t)6 * (NP) * (D))

/* ---------------------------------------------------------------- 2. RLEPSO instance state (HBM, doubles)
 * One contiguous block per instance (one workgroup streams it in and out per generation):
 *   cur_pos[NP*D] vel[NP*D] pbest_pos[NP*D] c_cost[NP] pbest[NP] per_no_improve[NP] gbest_pos[D]
 *   scalars[MBX_NSCALAR] cost_curve[n_logpoint+1]
 * (the fields of RLEPSO_Optimizer.__particles, rlepso_optimizer.py:53-61, plus fes/cost/log_index).   */
#define MBX_RLEPSO_ST_POS(NP, D)      ((int64_t)0)
#define MBX_RLEPSO_ST_VEL(NP, D)      ((int64_t)(NP) * (D))
#define MBX_RLEPSO_ST_PBPOS(NP, D)    ((int64_t)2 * (NP) * (D))
#define MBX_RLEPSO_ST_CCOST(NP, D)    ((int64_t)3 * (NP) * (D))
#define MBX_RLEPSO_ST_PBEST(NP, D)    ((int64_t)3 * (NP) * (D) + (NP))
#define MBX_RLEPSO_ST_PNI(NP, D)      ((int64_t)3 * (NP) * (D) + 2 * (NP))
#define MBX_RLEPSO_ST_GBPOS(NP, D)    ((int64_t)3 * (NP) * (D) + 3 * (NP))
#define MBX_RLEPSO_ST_SCALARS(NP, D)  ((int64_t)3 * (NP) * (D) + 3 * (NP) + (D))

/* scalar slots (shared by every algorithm's state block) */
#define MBX_SC_GBEST      0   /* gbest_val                                         */
#define MBX_SC_FES        1   /* optimizer.fes                                     */
#define MBX_SC_LOG_INDEX  2   /* optimizer.log_index                               */
#define MBX_SC_COST_LEN   3   /* len(optimizer.cost)                               */
#define MBX_SC_DONE       4   /* 1.0 once update() has returned is_done            */
#define MBX_SC_RETURN     5   /* sum of rewards (rollout_episode's R)              */
#define MBX_SC_GEN        6   /* number of update() calls executed this episode    */
#define MBX_SC_EPISODE    7   /* number of resets so far (Philox counter word 3)   */
#define MBX_SC_GBEST_IDX  8   /* gbest_index                                       */
#define MBX_SC_REINIT     9   /* 1.0 if __reinit fired in the last step (diagnostic) */
#define MBX_NSCALAR       16

#define MBX_RLEPSO_STATE_DOUBLES(NP, D, NLOG) \
    (MBX_RLEPSO_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)

/* ---------------------------------------------------------------- 3. Philox4x32-10 draw sites
 * key     = (seed_lo, seed_hi)             the instance's 64-bit seed
 * counter = (index, site, generation, episode)
 * A call yields four 32-bit words w0..w3.  u53(a,b) = ((a>>5)*2^26 + (b>>6)) / 2^53  in [0,1).
 * Integer draws in [0,n): mulhi32(w, n).  Normal pairs: Box-Muller on (1-u53(w0,w1), u53(w2,w3)):
 * r = sqrt(-2 ln(1-ua)), n0 = r cos(2 pi ub), n1 = r sin(2 pi ub).
 *
 *   site                 index        words
 *   MBX_SITE_ELEM_A      e >> 1       RLEPSO: ONE call per element pair carries its four element-wise uniforms, 32 bits each, u32(w) = w / 2^32:
 *                                     u32(w0) = clpso_u[e] and u32(w2) = fdr_u[e] for the even element, u32(w1) / u32(w3) for the odd one
 *                                     (the CLPSO mask compares against pci in [0.05, 0.5] and both act as step weights: 2^-32 resolution)
 *   MBX_SITE_TOURN       e >> 1       mulhi(w0,NP), mulhi(w1,NP) = tournament pair of the even element, mulhi(w2,NP), mulhi(w3,NP) = of the odd one;
 *                                     only consumed where !(clpso_u[e] > pci_i)
 *   MBX_SITE_ELEM_B      -            (RLEPSO: not drawn any more; the FDR weights ride in the ELEM_A call)
 *   MBX_SITE_PART        i            u53(w0,w1) = rand1[i];    u53(w2,w3) = rand2[i]
 *   MBX_SITE_REINIT      i            u53(w0,w1) = reinit_u[i]
 *   MBX_SITE_ELEM_R      e            u53(w0,w1) = pos_u[e];    u53(w2,w3) = vel_u[e]   (init + reinit)
 *   MBX_SITE_NOISE0_A/B  i            main evaluation:   A -> (ua,ub), B -> (uc,ud)
 *   MBX_SITE_NOISE1_A/B  i            init / reinit evaluation
 *       gauss  : N = n0(ua,ub)            uniform: U = ua, U' = ub
 *       cauchy : U = ua, (N, N') = (n0, n1)(uc,ud)
 */
#define MBX_SITE_ELEM_A    0u
#define MBX_SITE_ELEM_B    1u
#define MBX_SITE_PART      2u
#define MBX_SITE_REINIT    3u
#define MBX_SITE_ELEM_R    4u
#define MBX_SITE_NOISE0_A  5u
#define MBX_SITE_NOISE0_B  6u
#define MBX_SITE_NOISE1_A  7u
#define MBX_SITE_NOISE1_B  8u
/* stand-alone mbx_eval noise: counter = (row, MBX_SITE_EVAL_A/B, 0, 0) */
#define MBX_SITE_EVAL_A    9u
#define MBX_SITE_EVAL_B    10u

/* ---------------------------------------------------------------- 4. LDE (lde_optimizer.py) layouts
 * tape per step, draw order of LDE_Optimizer.update (:88-90 pbest index, :109-121 r1/r2 after rejection,
 * :44-47 crossover uniforms + jrand, eval noise):
 *   pbest_idx[NP] r0[NP] r1[NP] jrand[NP] noise[3*NP] cross_u[NP*D]
 * mbx_reset (init_population :133-134) uses cross_u as the position uniforms and noise[] for the evaluation.
 * state block: pop[NP*D] (kept sorted by fitness, :74-79) fit[NP] hist_sum[8] scalars[16] cost_curve[nlog+1];
 * hist_sum accumulates past_histo (:139,186), scalars[MBX_SC_HCOUNT] its length.
 * Philox: MBX_SITE_LDE_PART(i): mulhi(w0,bound)=pbest_idx, w1 -> r0 over the NP-1 indices != i, w2 -> r1 over the
 * NP-2 indices not in {i,r0} (same distribution as the reference's rejection loop), mulhi(w3,D)=jrand;
 * MBX_SITE_LDE_ELEM: reset (generation 0), index e: u53(w0,w1) = initial-position uniform;  step (generation >= 1), index e >> 2: ONE call per
 * four consecutive elements, u32(w[e & 3]) = w / 2^32 = crossover uniform of element e (compared against the float32 rate CR_i).               */
#define MBX_LDE_TAPE_PIDX(NP, D)   ((int64_t)0)
#define MBX_LDE_TAPE_R0(NP, D)     ((int64_t)(NP))
#define MBX_LDE_TAPE_R1(NP, D)     ((int64_t)2 * (NP))
#define MBX_LDE_TAPE_JRAND(NP, D)  ((int64_t)3 * (NP))
#define MBX_LDE_TAPE_NOISE(NP, D)  ((int64_t)4 * (NP))
#define MBX_LDE_TAPE_CROSS(NP, D)  ((int64_t)7 * (NP))
#define MBX_LDE_TAPE_STRIDE(NP, D) ((int64_t)7 * (NP) + (int64_t)(NP) * (D))
#define MBX_LDE_BINS 5
#define MBX_LDE_ST_POP(NP, D)      ((int64_t)0)
#define MBX_LDE_ST_FIT(NP, D)      ((int64_t)(NP) * (D))
#define MBX_LDE_ST_HSUM(NP, D)     ((int64_t)(NP) * (D) + (NP))      /* 8 doubles: [0..5) running sum of past_histo, [5] the last histogram, packed 10 bits per bin */
#define MBX_LDE_ST_SCALARS(NP, D)  ((int64_t)(NP) * (D) + (NP) + 8)
#define MBX_LDE_STATE_DOUBLES(NP, D, NLOG) (MBX_LDE_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_SC_HCOUNT 10
#define MBX_SITE_LDE_PART  11u
#define MBX_SITE_LDE_ELEM  12u

/* ---------------------------------------------------------------- 5. DE-DDQN (de_ddqn_optimizer.py) layouts
 * tape per step (draw order of update(): binomial's randint(D,1) and rand(1,D) (operators/crossover.py:11,14), eval
 * noise of the single trial, then __get_state's randint(0,NP,5) (:85)):
 *   r[5] @0 | jrand @5 | noise[3] @8 | cross_u[D] @16
 * mbx_reset: r[5] @0 | pos_u[NP*D] @16 | noise_init[3*NP] @16+NP*D.
 * state block: X[NP*D] cost[NP] gbest_pos[D] prebest_pos[D] r[8] N_tot[4*10] N_succ[4*4*10] OM_sum[4*4*10]
 *   OM_max[4*4*10] OM_W[50*6] extra[16] scalars[16] cost_curve[nlog+1].
 * The generation deques (maxlen gen_max = 10, appendleft) are rings: generation-back index g lives in slot
 * (g - gen) mod 10.  extra[]: see MBX_DQ_X_*.  Philox: MBX_SITE_DQ_R(idx 0: w0..w3 -> r0..r3, idx 1: w0 -> r4),
 * MBX_SITE_DQ_JRAND(idx 0: w0), cross_u -> MBX_SITE_LDE_ELEM(d), init positions -> MBX_SITE_LDE_ELEM(e).        */
#define MBX_DQ_GENMAX 10
#define MBX_DQ_W      50
#define MBX_DQ_NFEAT  99
#define MBX_DQ_TAPE_R(NP, D)       ((int64_t)0)
#define MBX_DQ_TAPE_JRAND(NP, D)   ((int64_t)5)
#define MBX_DQ_TAPE_NOISE(NP, D)   ((int64_t)8)
#define MBX_DQ_TAPE_CROSS(NP, D)   ((int64_t)16)
#define MBX_DQ_TAPE_POS(NP, D)     ((int64_t)16)
#define MBX_DQ_TAPE_NOISE_INIT(NP, D) ((int64_t)16 + (int64_t)(NP) * (D))
#define MBX_DQ_TAPE_STRIDE(NP, D)  ((int64_t)16 + (int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_DQ_ST_X(NP, D)        ((int64_t)0)
#define MBX_DQ_ST_COST(NP, D)     ((int64_t)(NP) * (D))
#define MBX_DQ_ST_GBPOS(NP, D)    ((int64_t)(NP) * (D) + (NP))
#define MBX_DQ_ST_PREPOS(NP, D)   (MBX_DQ_ST_GBPOS(NP, D) + (D))
#define MBX_DQ_ST_R(NP, D)        (MBX_DQ_ST_PREPOS(NP, D) + (D))
#define MBX_DQ_ST_NTOT(NP, D)     (MBX_DQ_ST_R(NP, D) + 8)
#define MBX_DQ_ST_NSUCC(NP, D)    (MBX_DQ_ST_NTOT(NP, D) + 40)
#define MBX_DQ_ST_OMSUM(NP, D)    (MBX_DQ_ST_NSUCC(NP, D) + 160)
#define MBX_DQ_ST_OMMAX(NP, D)    (MBX_DQ_ST_OMSUM(NP, D) + 160)
#define MBX_DQ_ST_OMW(NP, D)      (MBX_DQ_ST_OMMAX(NP, D) + 160)
#define MBX_DQ_ST_EXTRA(NP, D)    (MBX_DQ_ST_OMW(NP, D) + 300)
#define MBX_DQ_ST_SCALARS(NP, D)  (MBX_DQ_ST_EXTRA(NP, D) + 16)
#define MBX_DQ_STATE_DOUBLES(NP, D, NLOG) (MBX_DQ_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_DQ_X_GWORST   0   /* c_gworst                                            */
#define MBX_DQ_X_CPRE     1   /* c_prebest (never updated after init: reference :135) */
#define MBX_DQ_X_POINTER  2
#define MBX_DQ_X_GEN      3   /* the reference's __gen (population sweeps started)    */
#define MBX_DQ_X_STAG     4
#define MBX_DQ_X_OMWLEN   5
#define MBX_DQ_X_G0       6   /* row that X_gbest / X_prebest alias while they are numpy views (:55,58) */
#define MBX_DQ_X_GBVIEW   7
#define MBX_DQ_X_PREVIEW  8
#define MBX_DQ_X_MEDLO    9   /* cache: the cost order statistics NP/2 - 1 and NP/2 the last update() found (NaN after reset); */
#define MBX_DQ_X_MEDHI    10  /* the next update() re-validates them with one counting pass before it trusts them          */
#define MBX_SITE_DQ_R      13u
#define MBX_SITE_DQ_JRAND  14u
/* mbx_gauss_policy: index j = action component, u53(w0,w1), u53(w2,w3) -> Box-Muller, first normal used */
#define MBX_SITE_POLICY    15u

/* ---------------------------------------------------------------- 6. Random_search (random_search.py) layouts
 * tape per step / reset: pos_u[NP*D] | noise[3*NP];  state block: scalars[16] cost_curve[nlog+1].
 * Philox: positions MBX_SITE_LDE_ELEM(e), noise MBX_SITE_NOISE0_A/B(i), generation counter = number of populations
 * drawn so far in the episode (0 for the initial one).                                                              */
#define MBX_RS_TAPE_POS(NP, D)     ((int64_t)0)
#define MBX_RS_TAPE_NOISE(NP, D)   ((int64_t)(NP) * (D))
#define MBX_RS_TAPE_STRIDE(NP, D)  ((int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_RS_ST_SCALARS(NP, D)   ((int64_t)0)
#define MBX_RS_STATE_DOUBLES(NP, D, NLOG) ((int64_t)MBX_NSCALAR + (int64_t)(NLOG) + 1)

/* ---------------------------------------------------------------- 7. RL-PSO (rl_pso_optimizer.py) layouts
 * One env step = ONE particle (index `cur`, round robin): velocity / position update with the action as the gbest
 * attraction weight, one evaluation, pbest / gbest, reward (pre_cost - new_cost) / (max_cost - gbest) (:76-148).
 * state vector [2 D] = gbest_position | current_position[cur]  (:62-63);  action [1] float32.
 * state block: pos[NP*D] vel[NP*D] pbest_pos[NP*D] c_cost[NP] pbest[NP] gbest_pos[D] scalars[16] cost_curve[nlog+1];
 * scalars beyond the common ones: inertia w (decremented EVERY step, :85-86), max_cost of the initial population (:41),
 * cur.  tape per reset: pos_u[NP*D] | vel_u[NP*D] | noise[3*NP] (draw order of init_population :30-36);
 * tape per step: rand1 | noise[3] (:89, then the evaluation's own draws).
 * Philox: reset (gen 0): MBX_SITE_ELEM_R(e): u53(w0,w1) = pos_u, u53(w2,w3) = vel_u; noise MBX_SITE_NOISE1_A/B(i).
 *         step (gen = number of the step): MBX_SITE_PART(0): u53(w0,w1) = rand1; noise MBX_SITE_NOISE0_A/B(0).       */
#define MBX_RLPSO_TAPE_POS(NP, D)    ((int64_t)0)
#define MBX_RLPSO_TAPE_VEL(NP, D)    ((int64_t)(NP) * (D))
#define MBX_RLPSO_TAPE_NOISE_INIT(NP, D) (2 * (int64_t)(NP) * (D))
#define MBX_RLPSO_TAPE_RAND1(NP, D)  ((int64_t)0)
#define MBX_RLPSO_TAPE_NOISE(NP, D)  ((int64_t)1)
#define MBX_RLPSO_TAPE_STRIDE(NP, D) (2 * (int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_RLPSO_ST_POS(NP, D)      ((int64_t)0)
#define MBX_RLPSO_ST_VEL(NP, D)      ((int64_t)(NP) * (D))
#define MBX_RLPSO_ST_PBPOS(NP, D)    (2 * (int64_t)(NP) * (D))
#define MBX_RLPSO_ST_CCOST(NP, D)    (3 * (int64_t)(NP) * (D))
#define MBX_RLPSO_ST_PBEST(NP, D)    (3 * (int64_t)(NP) * (D) + (NP))
#define MBX_RLPSO_ST_GBPOS(NP, D)    (3 * (int64_t)(NP) * (D) + 2 * (int64_t)(NP))
#define MBX_RLPSO_ST_SCALARS(NP, D)  (3 * (int64_t)(NP) * (D) + 2 * (int64_t)(NP) + (D))
#define MBX_RLPSO_STATE_DOUBLES(NP, D, NLOG) (MBX_RLPSO_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_SC_RLPSO_W       10
#define MBX_SC_RLPSO_MAXCOST 11
#define MBX_SC_RLPSO_CUR     12

/* ---------------------------------------------------------------- 8. GLEET (gleet_optimizer.py) layouts
 * One env step = one PSO generation; action [NP] float32 = each particle's share of the pbest attraction (:206-210);
 * state [NP, 27] = 9 features of the particle (observe() :127-152) | the features it had when it last improved
 * (pbest_feature) | the features of the gbest particle when gbest last improved (gbest_feature)  (:111-124, 290-296).
 * state block: pos[NP*D] vel[NP*D] pbest_pos[NP*D] c_cost[NP] pbest[NP] per_no_improve[NP] gbest_pos[D]
 *              pbest_feature[NP*9] gbest_feature[9 (+1 pad)] scalars[16] cost_curve[nlog+1];
 * scalars beyond the common ones: inertia w (-= 0.5 / (maxFEs / NP) per generation), max_cost (= the MINIMUM of the initial
 * costs, :51), no_improve.
 * tape per reset: pos_u[NP*D] | vel_u[NP*D] | noise[3*NP];  per step: rand1[NP] | rand2[NP] | noise[3*NP].
 * Philox: reset (gen 0): MBX_SITE_ELEM_R(e): u53(w0,w1) = pos_u, u53(w2,w3) = vel_u; noise MBX_SITE_NOISE1_A/B(i).
 *         step (gen = generation): MBX_SITE_PART(i): u53(w0,w1) = rand1, u53(w2,w3) = rand2; noise MBX_SITE_NOISE0_A/B(i). */
#define MBX_GLEET_NFEAT 9
#define MBX_GLEET_TAPE_POS(NP, D)    ((int64_t)0)
#define MBX_GLEET_TAPE_VEL(NP, D)    ((int64_t)(NP) * (D))
#define MBX_GLEET_TAPE_NOISE_INIT(NP, D) (2 * (int64_t)(NP) * (D))
#define MBX_GLEET_TAPE_RAND1(NP, D)  ((int64_t)0)
#define MBX_GLEET_TAPE_RAND2(NP, D)  ((int64_t)(NP))
#define MBX_GLEET_TAPE_NOISE(NP, D)  (2 * (int64_t)(NP))
#define MBX_GLEET_TAPE_STRIDE(NP, D) (2 * (int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_GLEET_ST_POS(NP, D)      ((int64_t)0)
#define MBX_GLEET_ST_VEL(NP, D)      ((int64_t)(NP) * (D))
#define MBX_GLEET_ST_PBPOS(NP, D)    (2 * (int64_t)(NP) * (D))
#define MBX_GLEET_ST_CCOST(NP, D)    (3 * (int64_t)(NP) * (D))
#define MBX_GLEET_ST_PBEST(NP, D)    (3 * (int64_t)(NP) * (D) + (NP))
#define MBX_GLEET_ST_PNI(NP, D)      (3 * (int64_t)(NP) * (D) + 2 * (int64_t)(NP))
#define MBX_GLEET_ST_GBPOS(NP, D)    (3 * (int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_GLEET_ST_PFEAT(NP, D)    (3 * (int64_t)(NP) * (D) + 3 * (int64_t)(NP) + (D))
#define MBX_GLEET_ST_GFEAT(NP, D)    (MBX_GLEET_ST_PFEAT(NP, D) + 9 * (int64_t)(NP))
#define MBX_GLEET_ST_SCALARS(NP, D)  (MBX_GLEET_ST_GFEAT(NP, D) + 10)
#define MBX_GLEET_STATE_DOUBLES(NP, D, NLOG) (MBX_GLEET_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_SC_GLEET_W         10
#define MBX_SC_GLEET_MAXCOST   11
#define MBX_SC_GLEET_NOIMPROVE 12

/* ---------------------------------------------------------------- 9. QLPSO (qlpso_optimizer.py) layouts
 * One env step = ONE particle (round robin over NP = 30): ring-neighbourhood best of size 4 / 8 / 16 / 30 chosen by the action,
 * velocity (W = 0.729844, C = 1.49618, no velocity clamp), position clipping, one evaluation, swarm diversity, reward in
 * {2, 1, 0, -2} from (cost improved?, diversity grew?) (:7-16, 45-125).  state [1] = the action the NEXT particle took last time
 * (initially a random integer in 0..3, :89); action [1] int32 in 0..3.  The pointer is NOT reset by init_population (:33).
 * state block: pop[NP*D] vel[NP*D] pbest_pos[NP*D] cost[NP] sstate[NP] scalars[16] cost_curve[nlog+1];
 * scalars beyond the common ones: diversity, pointer.
 * tape per reset: pos_u[NP*D] | noise[3*NP] | sstate[NP];  per step: rand_a | rand_b | noise[3] | choice_u (the uniform that
 * QLPSO_Agent's np.random.choice consumes before the step; used by the fused policy only).
 * Philox: reset (gen 0): MBX_SITE_LDE_ELEM(e): u53(w0,w1) = pos_u; noise MBX_SITE_NOISE1_A/B(i); MBX_SITE_PART(i): mulhi(w0, 4) = sstate.
 *         step (gen = number of the step): MBX_SITE_PART(0): u53(w0,w1) = rand_a, u53(w2,w3) = rand_b; noise MBX_SITE_NOISE0_A/B(0);
 *         MBX_SITE_POLICY(0): u53(w0,w1) = choice_u.                                                                              */
#define MBX_QLPSO_TAPE_POS(NP, D)        ((int64_t)0)
#define MBX_QLPSO_TAPE_NOISE_INIT(NP, D) ((int64_t)(NP) * (D))
#define MBX_QLPSO_TAPE_SSTATE(NP, D)     ((int64_t)(NP) * (D) + 3 * (int64_t)(NP))
#define MBX_QLPSO_TAPE_RAND(NP, D)       ((int64_t)0)
#define MBX_QLPSO_TAPE_NOISE(NP, D)      ((int64_t)2)
#define MBX_QLPSO_TAPE_CHOICE(NP, D)     ((int64_t)5)
#define MBX_QLPSO_TAPE_STRIDE(NP, D)     ((int64_t)(NP) * (D) + 4 * (int64_t)(NP) + 8)
#define MBX_QLPSO_ST_POP(NP, D)          ((int64_t)0)
#define MBX_QLPSO_ST_VEL(NP, D)          ((int64_t)(NP) * (D))
#define MBX_QLPSO_ST_PBPOS(NP, D)        (2 * (int64_t)(NP) * (D))
#define MBX_QLPSO_ST_COST(NP, D)         (3 * (int64_t)(NP) * (D))
#define MBX_QLPSO_ST_SSTATE(NP, D)       (3 * (int64_t)(NP) * (D) + (NP))
#define MBX_QLPSO_ST_SCALARS(NP, D)      (3 * (int64_t)(NP) * (D) + 2 * (int64_t)(NP))
#define MBX_QLPSO_STATE_DOUBLES(NP, D, NLOG) (MBX_QLPSO_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_SC_QLPSO_DIVERSITY 10
#define MBX_SC_QLPSO_POINTER   11

/* ---------------------------------------------------------------- 10. classic baselines (deap_de.py, deap_pso.py, deap_cmaes.py)
 * No agent: mbx_reset builds and evaluates the initial population, every mbx_step (actions = NULL) is one sweep over the
 * population (DE, PSO: NP sequential single-individual updates, each billed 1 FE and checked for logging / termination like the
 * reference's inner loop) or one generation (CMA-ES).  state [1] = fes / maxFEs.  DEAP itself (deap==1.3.3, requirements.txt:7) is
 * not part of the reference tree: DE and PSO are fully written out in the reference's wrappers, CMA-ES follows deap.cma.Strategy's
 * published algorithm.  There are no reference traces for these (parity unpinned); no replay tape either.
 * DE   (deap_de.py:8-82, NP 50, F 0.5, Cr 0.5, selTournament(k=3, tournsize=3) donors):
 *      state block: X[NP*D] cost[NP] scalars[16] cost_curve[nlog+1].
 *      Philox, reset (gen 0): MBX_SITE_LDE_ELEM(e) -> position; noise MBX_SITE_NOISE1_A/B(i).
 *      sweep (gen = sweep number): MBX_SITE_CLASSIC(3k+j), j = 0..2: mulhi(w0..w2, NP) = the three aspirants of donor j,
 *      w3 of j = 0: mulhi(w3, D) = forced crossover index; MBX_SITE_LDE_ELEM(k*D+i) -> crossover uniform; noise MBX_SITE_NOISE0_A/B(k).
 * PSO  (deap_pso.py:8-122, 50 particles, phi1 = phi2 = 2, speed in +-ub/2, gbest updated inside the sweep):
 *      state block: X[NP*D] speed[NP*D] pbest_pos[NP*D] pbest[NP] gbest_pos[D] scalars[16] cost_curve[nlog+1].
 *      Philox, reset: MBX_SITE_ELEM_R(e): u53(w0,w1) -> position, u53(w2,w3) -> speed; noise MBX_SITE_NOISE1_A/B(i).
 *      sweep: MBX_SITE_ELEM_A(k*D+i): u53(w0,w1) -> u1, u53(w2,w3) -> u2; noise MBX_SITE_NOISE0_A/B(k).
 * CMAES (deap_cmaes.py:12-66 + deap.cma.Strategy, lambda 50, centroid = ub, sigma 0.5):
 *      state block: centroid[D] C[D*D] B[D*D] diagD[D] ps[D] pc[D] scalars[16] cost_curve[nlog+1]; scalars: sigma, update_count.
 *      Philox, generation g >= 1: MBX_SITE_ELEM_A(i*D+d): Box-Muller(u53(w0,w1), u53(w2,w3)) first normal -> arz[i][d];
 *      noise MBX_SITE_NOISE0_A/B(i).                                                                                            */
#define MBX_SITE_CLASSIC     16u
#define MBX_SITE_TOURN       17u   /* RLEPSO, see the table above */
#define MBX_DE_ST_X(NP, D)          ((int64_t)0)
#define MBX_DE_ST_COST(NP, D)       ((int64_t)(NP) * (D))
#define MBX_DE_ST_SCALARS(NP, D)    ((int64_t)(NP) * (D) + (NP))
#define MBX_DE_STATE_DOUBLES(NP, D, NLOG) (MBX_DE_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_PSO_ST_X(NP, D)         ((int64_t)0)
#define MBX_PSO_ST_SPEED(NP, D)     ((int64_t)(NP) * (D))
#define MBX_PSO_ST_PBPOS(NP, D)     (2 * (int64_t)(NP) * (D))
#define MBX_PSO_ST_PBEST(NP, D)     (3 * (int64_t)(NP) * (D))
#define MBX_PSO_ST_GBPOS(NP, D)     (3 * (int64_t)(NP) * (D) + (NP))
#define MBX_PSO_ST_SCALARS(NP, D)   (3 * (int64_t)(NP) * (D) + (NP) + (D))
#define MBX_PSO_STATE_DOUBLES(NP, D, NLOG) (MBX_PSO_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_CMA_ST_CENTROID(NP, D)  ((int64_t)0)
#define MBX_CMA_ST_C(NP, D)         ((int64_t)(D))
#define MBX_CMA_ST_B(NP, D)         ((int64_t)(D) + (int64_t)(D) * (D))
#define MBX_CMA_ST_DIAGD(NP, D)     ((int64_t)(D) + 2 * (int64_t)(D) * (D))
#define MBX_CMA_ST_PS(NP, D)        (2 * (int64_t)(D) + 2 * (int64_t)(D) * (D))
#define MBX_CMA_ST_PC(NP, D)        (3 * (int64_t)(D) + 2 * (int64_t)(D) * (D))
#define MBX_CMA_ST_SCALARS(NP, D)   (4 * (int64_t)(D) + 2 * (int64_t)(D) * (D))
#define MBX_CMA_STATE_DOUBLES(NP, D, NLOG) (MBX_CMA_ST_SCALARS(NP, D) + MBX_NSCALAR + (int64_t)(NLOG) + 1)
#define MBX_SC_CMA_SIGMA   10
#define MBX_SC_CMA_UPDATES 11

#define MBX_PHILOX_M0 0xD2511F53u
#define MBX_PHILOX_M1 0xCD9E8D57u
#define MBX_PHILOX_W0 0x9E3779B9u
#define MBX_PHILOX_W1 0xBB67AE85u

#endif /* MBX_LAYOUT_H */
