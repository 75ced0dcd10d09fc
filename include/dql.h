/*
 * dql.h — C ABI of libdql_hip.so: the MI355X (gfx950) vectorised UAV-landing environment +
 * tabular Double-Q trainer hot path.
 *
 * This is the drop-in boundary.  The reference has no FFI: the path sits behind
 * Python classes.  Each entry point below names the reference interface it replaces; the Python
 * host package `dql_multirotor_landing_amd` binds these with ctypes and re-exposes the reference's
 * class/method names (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 (DQL_OK) or a negative dql_status; dql_last_error() gives
 * the thread-local message of the last failure.  All pointers are plain host pointers unless the
 * name says "dev".  A dql_ctx owns its device memory and stream; it is not thread-safe; calls on
 * one ctx are stream-ordered and asynchronous until a dql_get_* / dql_stats / dql_sync call.
 * No torch types, no C++ types.
 */
#ifndef DQL_H
#define DQL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DQL_ABI_VERSION 6 /* 6: dql_agent_mirror_update_deferred / _complete; 5: tick replay operators (dql_*_run), measurement symbols moved to dql_diag.h as dql_diag_* */

typedef enum dql_status {
  DQL_OK = 0,
  DQL_EINVAL = -1, /* bad argument / unsupported configuration (Python: ValueError) */
  DQL_EHIP = -2,   /* HIP runtime error (Python: RuntimeError) */
  DQL_ESTATE = -3, /* call order violation, e.g. step before reset (Python: ValueError) */
  DQL_ENOMEM = -4,
  DQL_ERCCL = -5,  /* RCCL error, or librccl could not be loaded (Python: RuntimeError) */
  DQL_EPEER = -6   /* a peer-to-peer table exchange gave up on a missing peer: the replicas may differ from here on (Python: RuntimeError) */
} dql_status;

/* CheckResult codes, declaration order of pkg/mdp.py:68-77 */
enum {
  DQL_TERMINAL_CONTACT = 0,
  DQL_TERMINAL_SUCCESS = 1,
  DQL_TERMINAL_FLYZONE_X = 2,
  DQL_TERMINAL_FLYZONE_Y = 3,
  DQL_TERMINAL_FLYZONE_Z = 4,
  DQL_TERMINAL_MINIMUM_ALTITUDE = 5,
  DQL_TERMINAL_TIMEOUT = 6,
  DQL_NON_TERMINAL_SUCCESS = 7,
  DQL_NON_TERMINAL = 8,
  DQL_N_CHECK_CODES = 9
};

/* Quirk switches.  mode="reference" = all set; mode="paper" = DQL_Q_GOAL_COUNT_KEPT only. */
enum {
  DQL_Q_FAIL_TERM_EVERY_STEP = 1 << 0,  /* B7  pkg/mdp.py:528-536 */
  DQL_Q_STICKY_CHECK = 1 << 1,          /* B8  pkg/mdp.py:363-425: _check_result is only ever set inside an episode, so after the first
                                         * NON_TERMINAL_SUCCESS every later step keeps the success code (and its reward term) */
  DQL_Q_SHAPING_SURVIVES_RESET = 1 << 2,/* B9  pkg/mdp.py:196-197,469-474 */
  DQL_Q_FROZEN_ACC_REFERENCE = 1 << 3,  /* B19 pkg/observation_utils.py:137-150: last_velocity never updated */
  DQL_Q_BOOTSTRAP_ON_POS_CHANGE = 1 << 4,/* B3  pkg/double_q_learning.py:139-145 */
  DQL_Q_UPDATE_TABLE_A_ONLY = 1 << 5,   /* B1/B2 pkg/double_q_learning.py:101-108,136-146: both arms of the coin pick Q_table_a and
                                         * it values its own greedy action.  Cleared (mode="paper"): Double Q-learning as the paper
                                         * has it: a fair coin picks the table to update, the OTHER table values the picked
                                         * table's greedy action at s'; Q_table_b learns too. */
  DQL_Q_GOAL_COUNT_KEPT = 1 << 6,       /* second half of B8, pkg/mdp.py:402-425: _curriculum_check counts the steps spent in the goal bins at
                                         * the working level and is reset only by a goal-bin step at another level — leaving the bins keeps
                                         * it, so "Goal state reached" needs f_ag such steps in TOTAL, not in a row.  This is the success
                                         * criterion the reference's trainer promotes on (B17), kept in paper mode.  Cleared: the counter
                                         * restarts whenever the goal bins are left (one second WITHOUT interruption inside the innermost
                                         * bins: stricter than the code and than the paper's "in that curriculum step's discrete states") */
  DQL_Q_REFERENCE = 0x7f
};

enum { DQL_F32 = 0, DQL_F64 = 1 };
enum { DQL_TRAJ_RPM = 0, DQL_TRAJ_EIGHT = 1 };

#define DQL_MAX_LEVELS 5
#define DQL_N_ACTIONS 3
#define DQL_N_ANGLES 7
#define DQL_STATES_PER_LEVEL (3 * 3 * 3 * DQL_N_ANGLES)            /* 189 */
#define DQL_CELLS_PER_LEVEL (DQL_STATES_PER_LEVEL * DQL_N_ACTIONS) /* 567 */
#define DQL_N_STATES (DQL_MAX_LEVELS * DQL_STATES_PER_LEVEL)       /* 945 */
#define DQL_N_CELLS (DQL_MAX_LEVELS * DQL_CELLS_PER_LEVEL)         /* 2835 = (5,3,3,3,7,3) C-order */
#define DQL_TARGET_FRAC_BITS 26 /* fixed-point scale of the int64 TD-target accumulators */

/* One POD configuration block; the Python dataclass `DqlConfig` mirrors it field for field with the
 * reference's defaults (constants harvested in SURVEY.md appendix A, cited per field). */
typedef struct dql_config {
  /* ---- MDP: pkg/mdp.py:87-147, 214-255 ---- */
  int32_t working_curriculum_step; /* 0..4 */
  int32_t two_axis;                /* 0: TrainingMdp (x / pitch), 1: x+y (SimulationMdp layout) */
  uint32_t quirks;                 /* DQL_Q_* */
  int32_t dtype;                   /* DQL_F32 | DQL_F64: arithmetic of the fused simulator kernel */
  double f_ag, t_max, p_max, v_max, a_max;
  double theta_max, delta_theta, beta, sigma_a, minimum_altitude;
  double w_p, w_v, w_theta, w_dur, w_fail, w_succ;
  double lim_p[DQL_MAX_LEVELS], lim_v[DQL_MAX_LEVELS], lim_a[DQL_MAX_LEVELS]; /* pkg/mdp.py:45-53 */
  double vz_setpoint, yaw_setpoint; /* pkg/mdp.py:212 (-0.1 training), :580 (-0.4 simulation) */
  /* ---- agent / trainer: pkg/trainer.py:31-33 ---- */
  double gamma, alpha_min, alpha_omega;
  /* ---- simulator (replaces Gazebo + RotorS + 5 ROS nodes) ---- */
  double dt;           /* worlds/basic.world:64-70  0.002 */
  int32_t manager_div; /* physics ticks per manager tick: 100 Hz -> 5 (launch/environment.launch:55) */
  int32_t trajectory;  /* DQL_TRAJ_* (pkg/moving_platform.py:87-127) */
  double gravity;      /* 9.8 (worlds/basic.world:36) */
  double mass;         /* 0.68 + 4*0.009 + 1e-5 (hummingbird.xacro:29,32) */
  double inertia[3];   /* composite diagonal about the base origin */
  double arm_length, rotor_z;   /* 0.17, 0.01 */
  double k_f, k_m;              /* 8.54858e-06, 0.016 */
  double rotor_alpha_up, rotor_alpha_down; /* exp(-dt/0.0125), exp(-dt/0.025) (common.h:147-183) */
  double rotor_max;             /* 838 */
  double c_drag, c_roll;        /* 8.06428e-05, 1e-06 */
  double k_R[3], k_W[3];        /* pkg/attitude_controller.py:86-87 */
  double pid_vz[6];             /* Kp Ki Kd lower upper windup: launch/drone.launch:35-40 */
  double pid_yaw[6];            /* launch/drone.launch:49-54 */
  double bw_c;                  /* Butterworth c = 1 (pkg/filters.py:93) */
  double mp_r_x, mp_t_x;        /* platform amplitude and speed, omega = t_x / r_x (environment.launch:60-72) */
  double mp_dt;                 /* 1 / frequency = 0.01 */
  double mp_top_z, mp_half_x, mp_half_y; /* landing surface 0.455, half extents 0.5 (+ drone half width) */
  double drone_bottom;          /* base box half height 0.06 */
  double z_init, init_sigma;    /* pkg/trainer.py:41 (4.0), p_max/3 (landing_simulation_env.py:189) */
  int32_t init_uniform;         /* start offset x0 and placement: 0: x0 ~ N(0,sigma) at level 0 else U(-p_max, p_max), drone at clip(x0 + mp, mp +- p_max)
                                   (TrainingLandingEnv.reset, pkg/landing_simulation_env.py:181-209); 1: always U, same placement; 2: always U,
                                   drone at clip(mp - x0, +-p_max) as SimulationLandingEnv.reset has it (:331-343: offset subtracted, absolute clip) */
  int32_t per_env_platform;     /* 1: r_x, t_x drawn per env from the ranges below (BASELINE config 5) */
  int32_t goal_logic;           /* 1: TrainingMdp.check goal / success branch (pkg/mdp.py:402-425); 0: SimulationMdp.check (:784-845) */
  int32_t fold_per_step;        /* 0: per-visit fold (prod of (1-alpha) over the m visits of a launch); 1: ONE alpha step per launch towards the
                                   launch's mean target (count still advances by m): smoother at large N */
  double mp_r_lo, mp_r_hi, mp_t_lo, mp_t_hi;
  double noise_pos_sd, noise_vel_sd, kalman_q; /* scripts/manager_node.py:83-98 */
} dql_config;

/* Aggregated counters since the last dql_stats_reset (device-side reductions). */
typedef struct dql_stats {
  int64_t agent_steps;   /* launches of the fused step kernel */
  int64_t decisions;     /* env-steps: (env, step) pairs in which an action was taken (reset periods excluded) */
  int64_t episodes;      /* finished episodes */
  int64_t by_code[DQL_N_CHECK_CODES]; /* terminal histogram by CheckResult code */
  double reward_sum;     /* sum of step rewards over all decisions */
  int64_t physics_ticks; /* global physics tick index */
} dql_stats;

typedef struct dql_ctx dql_ctx;

/* ---- library ---- */
int dql_abi_version(void);
const char* dql_last_error(void);
int dql_device_count(int* count);
/* fill cfg with the reference defaults; training flavour */
int dql_config_default(dql_config* cfg);

/* ---- context: replaces gym.make("Landing-Training-v0") + DoubleQLearningAgent() (pkg/trainer.py:176-183,46-48) ---- */
int dql_create(const dql_config* cfg, int device, int64_t n_envs, uint64_t seed, int64_t env_id_offset, dql_ctx** out);
int dql_destroy(dql_ctx* ctx);
int dql_sync(dql_ctx* ctx);
int dql_n_envs(dql_ctx* ctx, int64_t* n);
int dql_state_bytes_per_env(dql_ctx* ctx, int64_t* bytes);

/* alpha(count) table: alpha[c] for c < n, alpha_min beyond (pkg/trainer.py:88-110, computed by the host exactly
 * as the reference does so that device and oracle share bit-identical values) */
int dql_set_alpha_table(dql_ctx* ctx, const double* alpha, int32_t n);
/* Trainer creates a new env per curriculum level (pkg/trainer.py:172-183): new limits, every env re-enters via reset */
int dql_set_curriculum(dql_ctx* ctx, int32_t working_curriculum_step);

/* ---- env: TrainingLandingEnv.reset / .step (pkg/landing_simulation_env.py:167-282) ---- */
/* mark envs for reset (mask NULL = all); the reset (placement + one agent period + first discrete state)
 * is executed by the next step/train call, as the reference's reset() runs one agent period of simulation */
int dql_reset(dql_ctx* ctx, const uint8_t* mask_or_null);
/* one agent step with caller-supplied actions (uint8 per env: ax | ay << 2, ax, ay in 0 / 1 / 2, ay only in two_axis configs); no table
 * update.  The actions are staged through pinned host memory.  Up to 16 384 envs (the single-env drop-in path among them) the codes are
 * checked while they are staged and a bad one is REFUSED before anything is flown: DQL_EINVAL, no env has moved (the reference raises before
 * publishing the action, pkg/mdp.py:544-545).  Beyond that the step kernel checks them per env (no host loop over n_envs, no wait): an
 * out-of-range action is flown as "hold" and makes the next dql_step_outputs / dql_stats_get return DQL_EINVAL once — a caller that reads
 * results with dql_get_states / dql_get_sim_state only must ask dql_stats_get for the verdict. */
int dql_step(dql_ctx* ctx, const uint8_t* actions);
/* what `TrainingLandingEnv.step` returns (pkg/landing_simulation_env.py:245-282), for every env, in ONE device round trip: packed
 * states, reward, done ("Termination condition" in info), CheckResult code, "Number of steps", cumulative reward (after this step's
 * reward), and whether the env spent the period in reset().  Any output may be NULL.  The step kernel's results are gathered into
 * pinned host memory by a small kernel on the context's stream; the call returns when it has run. */
int dql_step_outputs(dql_ctx* ctx, int32_t* idx_x, int32_t* idx_y, double* reward, uint8_t* done, int8_t* code, int32_t* step_count,
                     double* cumulative_reward, uint8_t* was_reset);
/* same with the actions already in device memory (n_envs bytes, readable on the context's device): no copy, no host-side
 * validation, fully asynchronous (the buffer must stay valid until the step has run); values other than 0 / 1 act as "hold" */
int dql_step_dev(dql_ctx* ctx, const uint8_t* dev_actions);
/* n fused agent steps: on-device eps-greedy guess + step + TD-target accumulation + table apply
 * (pkg/trainer.py:191-212 loop body for all envs at once) */
int dql_train_steps(dql_ctx* ctx, int32_t n_steps, double eps);
/* n greedy steps without learning (scripts/simulation.py:49-56) */
int dql_eval_steps(dql_ctx* ctx, int32_t n_steps);

/* per-env results of the last step (host out buffers of n_envs elements) */
int dql_get_states(dql_ctx* ctx, int32_t* idx_x, int32_t* idx_y_or_null); /* packed ((((k*3+p)*3+v)*3+a)*7+theta) */
int dql_get_rewards(dql_ctx* ctx, double* rewards);
int dql_get_dones(dql_ctx* ctx, uint8_t* dones, int8_t* codes_or_null);
int dql_get_actions(dql_ctx* ctx, uint8_t* actions);
/* raw continuous state, field-major [n_fields][n_envs] as doubles (see dql_field_names).
 * All 64 real fields mean the same in float32 and float64 contexts EXCEPT the two PIDs' Butterworth filter fields vz_x1 vz_x2 vz_y1 vz_y2 vz_y3 and
 * yw_*: a float64 context keeps the reference's histories there (x1, x2 = the last two inputs, y1..y3 = the last three outputs, pkg/filters.py:98-109),
 * a float32 context the three states of the same recurrence in transposed form in (x1, x2, y1) with y2 = y3 = 0:
 *   t1 = 2b x1 + b x2 - a2 y2 - a3 y3,  t2 = b x1 - a2 y1 - a3 y2,  t3 = -a3 y1   (b, a2, a3: csrc/dql_device.hpp butterworth).
 * State written by one dtype must be mapped before it is handed to the other (host: dql_multirotor_landing_amd/state_layout.py; float64 -> float32
 * only: three numbers do not determine five).  The library cannot tell the two apart: dql_set_sim_state trusts the caller.  The Trainer's
 * checkpoints carry dtype + layout version and map or refuse a shard by themselves.
 * A float32 x-axis context (two_axis = 0) flies the attitude law's closed form for a roll set-point of exactly 0: dql_set_sim_state refuses
 * (DQL_EINVAL) a non-zero roll_sp field there instead of ignoring it. */
int dql_get_sim_state(dql_ctx* ctx, double* out, int32_t n_fields_capacity);
int dql_set_sim_state(dql_ctx* ctx, const double* in, int32_t n_fields);
int dql_get_sim_ints(dql_ctx* ctx, int32_t* out, int32_t n_fields_capacity);
int dql_set_sim_ints(dql_ctx* ctx, const int32_t* in, int32_t n_fields);
int dql_n_fields(int32_t* n_real, int32_t* n_int);
const char* dql_field_name(int32_t i, int32_t is_int);
/* last latched observation per env: [6][n] = rel_p_x, rel_p_y, rel_v_x, rel_v_y, rel_a_x, rel_a_y */
int dql_get_obs(dql_ctx* ctx, double* out);

/* ---- tables: DoubleQLearningAgent.Q_table_a/_b/state_action_counter (pkg/double_q_learning.py:35-40) ---- */
int dql_get_tables(dql_ctx* ctx, double* qa, double* qb, double* count); /* 2835 doubles each, C order (5,3,3,3,7,3) */
int dql_set_tables(dql_ctx* ctx, const double* qa, const double* qb, const double* count);
int dql_transfer(dql_ctx* ctx, int32_t k, double ratio); /* transfer_learning (pkg/double_q_learning.py:77-89) */

/* ---- multi-GPU exchange: int64 accumulators [4][2835] = Q_table_a's {sum of targets (fixed point), visits},
 * then Q_table_b's (all zero under DQL_Q_UPDATE_TABLE_A_ONLY) ---- */
int dql_set_sync_period(dql_ctx* ctx, int32_t k_steps); /* 1 = apply every step (single-GPU semantics) */
/* windowed accumulation: every step also adds its accumulators into the window buffer and updates only the local
 * work tables; dql_apply_accum folds the (all-reduced) window into the base tables and re-bases the work tables */
int dql_set_windowed(dql_ctx* ctx, int32_t on);
/* use a caller-owned device buffer of 4*2835 int64 as the window (e.g. a torch tensor handed to torch.distributed);
 * NULL restores the context's own buffer.  The buffer is zeroed; the context never frees it. */
int dql_set_window_buffer(dql_ctx* ctx, void* dev_ptr);
int dql_stream_handle(dql_ctx* ctx, void** hip_stream);
/* The table update of launch j rides in the writer workgroups of launch j+1 (tables act with one period of delay);
 * dql_flush folds what is still pending (master tables and window) now.  Table getters / setters flush by themselves;
 * call it before all-reducing the window buffer directly. */
int dql_flush(dql_ctx* ctx);
/* fold the (all-reduced) window into the base tables; master and acting tables restart from them.  With fold_per_step a
 * cell visited m times (all ranks together) in a window of L launches takes min(m, L) learning-rate steps towards the
 * window's mean target (L = 1: the single-launch rule), so the step size per env-step does not depend on the sync period */
int dql_apply_accum(dql_ctx* ctx);
int dql_get_accum(dql_ctx* ctx, int64_t* out); /* host copy of the 4*2835 window words, for tests */
int dql_set_accum(dql_ctx* ctx, const int64_t* in);
/* agent periods launched so far: the physics tick schedule and the per-period RNG counters are functions of it, so a
 * checkpointed run resumes by restoring the env fields (dql_set_sim_state / _ints), the tables and this index */
int dql_get_step_index(dql_ctx* ctx, int64_t* step_index);
int dql_set_step_index(dql_ctx* ctx, int64_t step_index);
/* checkpoint barrier: fold what is pending and make the acting tables equal to the master tables (what a resumed run starts from) */
int dql_publish_tables(dql_ctx* ctx);

/* ---- RCCL communicator (ncclAllReduce(ncclInt64, ncclSum) of the window over xGMI; no PyTorch) ----
 * librccl.so is loaded on first use (dlopen), so single-GPU users never pay for it.  One process per GPU: rank 0 calls
 * dql_comm_unique_id and hands the 128 bytes to the other ranks out of band (the Python host uses a file, comm.py), then
 * every rank calls dql_comm_create.  The host-buffer collectives below are the Trainer's control plane (chunk counters,
 * judged envs' episode logs, the bench's max-over-ranks timing); they stage through a device buffer on the communicator's
 * own stream and return when the result is in `inout` / `out`. */
#define DQL_COMM_ID_BYTES 128
typedef struct dql_comm dql_comm;
enum { DQL_OP_SUM = 0, DQL_OP_MAX = 1 };
int dql_comm_unique_id(uint8_t* id_out /* [DQL_COMM_ID_BYTES] */);
int dql_comm_create(int device, int32_t rank, int32_t world, const uint8_t* id /* [DQL_COMM_ID_BYTES] */, dql_comm** out);
int dql_comm_destroy(dql_comm* comm);
int dql_comm_info(dql_comm* comm, int32_t* rank, int32_t* world, int32_t* device);
int dql_comm_allreduce_f64(dql_comm* comm, double* inout, int64_t n, int32_t op);
int dql_comm_allreduce_i64(dql_comm* comm, int64_t* inout, int64_t n, int32_t op);
int dql_comm_allgather_u64(dql_comm* comm, const uint64_t* in, int64_t n, uint64_t* out /* [world][n], rank order */);
int dql_comm_barrier(dql_comm* comm);
/* data path: the context's window is summed over the ranks in place, on the context's stream (asynchronous; the fold
 * that follows, dql_apply_accum, is stream-ordered behind it).  Flushes first.  NULL detaches. */
int dql_attach_comm(dql_ctx* ctx, dql_comm* comm_or_null);
int dql_allreduce_window(dql_ctx* ctx);

/* ---- one-shot peer-to-peer exchange (second step; no counterpart in the reference) ----
 * The same sum of the window accumulators without a collective: every rank pushes its 90 KB window into a slot of EVERY rank's
 * exchange buffer (HIP IPC mappings of uncached device memory: world concurrent one-hop writes over the direct xGMI links), raises
 * a flag there, waits for its peers' flags and adds the slots up in rank order.  Call order on every rank:
 *   dql_p2p_create(ctx, rank, world, handle)  -> 64-byte IPC handle of this rank's buffer; gather all ranks' handles (any way:
 *   dql_comm_allgather_u64, files, ...), dql_p2p_connect(ctx, handles in rank order), then per exchange
 *   dql_p2p_exchange_window(ctx) (asynchronous, on the context's stream; includes the flush) followed by dql_apply_accum(ctx) —
 *   the drop-in for dql_allreduce_window.  A peer that never shows up makes the wait give up after option "p2p_spin_limit" polls
 *   (default 60 M, a minute or two) instead of hanging the GPU.  ONE waiter decides per exchange: the window is then either the full
 *   sum or untouched (this rank's own accumulators), never half-summed; the sequence number of the first failed exchange is kept
 *   (dql_p2p_status: failed_seq, 0 = none) and from then on dql_stats_get — the per-chunk synchronisation point of the training
 *   loop — returns DQL_EPEER, so a training run cannot carry on with diverging replicas.  world <= DQL_P2P_MAX_RANKS.
 *   Ranks that share ONE GPU (tests, rehearsals) need HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment before the first HIP call:
 *   the host driver only supports dmabuf IPC, and hipIpcGetMemHandle fails with "invalid argument" without it. */
#define DQL_P2P_HANDLE_BYTES 64
#define DQL_P2P_MAX_RANKS 8
int dql_p2p_create(dql_ctx* ctx, int32_t rank, int32_t world, uint8_t* handle_out);
int dql_p2p_connect(dql_ctx* ctx, const uint8_t* all_handles);
/* peers that live in THIS process (one host thread driving several contexts, SURVEY.md section 8b "threading"): their exchange buffers
 * are connected by pointer instead of through an IPC handle (which cannot be opened by the process that exported it).  peers[world] in
 * rank order, NULL for ranks of other processes (those come through dql_p2p_connect).  Peer access between devices is enabled here. */
int dql_p2p_connect_local(dql_ctx* ctx, dql_ctx* const* peers);
int dql_p2p_exchange_window(dql_ctx* ctx);
/* the two halves of dql_p2p_exchange_window, for a host thread that drives several ranks: enqueue EVERY local rank's push (flush +
 * push + signal) before any local rank's wait (wait + sum) — streams of one process may share a hardware queue, and a wait kernel
 * queued ahead of the push it waits for would only end at its poll bound */
int dql_p2p_push_window(dql_ctx* ctx);
int dql_p2p_wait_window(dql_ctx* ctx);
int dql_p2p_status(dql_ctx* ctx, int32_t* failed_seq);

/* ---- stats ----  (timers, the kernel's self-test and other measurement equipment: include/dql_diag.h) */
int dql_stats_get(dql_ctx* ctx, dql_stats* out);
int dql_stats_reset(dql_ctx* ctx);
/* knobs: "block" (0 = auto, 64, 128, 256 threads per workgroup; 512 = float32 at 4 waves per SIMD); "tick" (0 = auto; layout of the
 * 500 Hz loop: 1 plain loop on scalar-register constants, 2 constants in vector registers + loop laid out per manager period, 3 the
 * packed float32 tick, 4 constants as instruction literals — float32 contexts whose vehicle / controller constants are the
 * reference's, DQL_EINVAL otherwise: same arithmetic, bit for bit, in all);
 * "periods_per_launch" P in 1..32 (default 1): dql_train_steps / dql_eval_steps run P agent periods per kernel launch — every env
 * stays in registers between them, so the state round trip through HBM and the launch boundary are paid once per P periods.
 * Table timing in units of launches is unchanged (a launch acts on every accumulator up to the launch before the previous one,
 * all P periods of a launch act on the same tables and add to the same accumulators; a per-step fold takes min(visits, P)
 * learning-rate steps per launch); P = 1 is the period-by-period schedule.  n_steps need not be a multiple of P.
 * "fair_prio" (-1 automatic, 0, 1): the waves that share a SIMD take turns at the issue priority (s_setprio by period + wave-slot parity) so that they finish a
 * launch together instead of the older one first; automatic = when the context has more env waves than the device has SIMDs; 1 for contexts that share one GPU. */
int dql_set_option(dql_ctx* ctx, const char* name, int32_t value);

/* ---- episode log (Trainer promotion rule, pkg/trainer.py:218-232) ----
 * The reference appends one 0/1 ("Goal state reached" in info["Termination condition"]) per finished episode to a
 * deque(maxlen=100) and promotes when its sum / 100 exceeds 0.96: the rule needs the ORDER in which episodes finish.  With
 * the log enabled every launch records, per wave of 64 envs, a 64-bit mask of the envs whose episode ended in that agent
 * period and a mask of those that ended in TERMINAL_SUCCESS; periods in launch order x envs in index order is the global
 * completion order.  n_waves = (n_envs + 63) / 64.  A launch with a full log fails with DQL_ESTATE. */
int dql_episode_log_enable(dql_ctx* ctx, int32_t capacity_periods); /* 0 disables and frees */
/* copies the periods logged since the last read into done_masks / goal_masks [n_periods][n_waves] and empties the log */
int dql_episode_log_read(dql_ctx* ctx, uint64_t* done_masks, uint64_t* goal_masks, int32_t max_periods, int32_t* n_periods);
/* the same, restricted to the first n_words words (64 envs each) of every period: done_masks / goal_masks [n_periods][n_words].  The
 * promotion rule (pkg/trainer.py:218-232) judges a fixed few envs — the first global env ids —, the population's totals come from dql_stats_get. */
int dql_episode_log_read_words(dql_ctx* ctx, uint64_t* done_masks, uint64_t* goal_masks, int32_t max_periods, int32_t n_words, int32_t* n_periods);

/* ---- stateless batch operators (host arrays in/out, computed on the device; drop-in class methods) ---- */
/* TrainingMdp.discrete_state (pkg/mdp.py:257-333): 4 x double[n] -> packed idx int32[n]; -1 where the reference raises */
int dql_discretise(const dql_config* cfg, int device, const double* rel_p, const double* rel_v, const double* rel_a,
                   const double* angle, int64_t n, int32_t* idx_out);
/* TrainingMdp / SimulationMdp methods for n independent MDPs (pkg/mdp.py:257-560, 625-877).  `stages` selects which
 * of the reference's methods run in this call, in the reference's call order:
 *   DQL_MDP_ACTION (continuous_action) | DQL_MDP_DISCRETISE (discrete_state) | DQL_MDP_CHECK (check) | DQL_MDP_REWARD (reward)
 *   DQL_MDP_SIMULATION: SimulationMdp.check flavour (no goal / success logic, pkg/mdp.py:784-845)
 * mdp_state: double[8][n] in/out = angle set-point, shaping_p, shaping_v, shaping_a, cumulative, step_count,
 * curriculum_check, check_code; prev_idx int32[n] in (-1 = none); idx_io int32[n]: current packed state, written by
 * DISCRETISE and read by CHECK / REWARD; obs double[7][n] = rel_p (axis), rel_p (other axis), rel_v, rel_a, angle,
 * abs_p_z, contact */
enum { DQL_MDP_ACTION = 1, DQL_MDP_DISCRETISE = 2, DQL_MDP_CHECK = 4, DQL_MDP_REWARD = 8, DQL_MDP_SIMULATION = 16, DQL_MDP_ALL = 15 };
int dql_mdp_transition(const dql_config* cfg, int device, int64_t n, uint32_t stages, const uint8_t* action, const double* obs,
                       double* mdp_state, const int32_t* prev_idx, int32_t* idx_io, double* reward_out, uint8_t* done_out);
/* ManagerNode.publish_obs (scripts/manager_node.py:192-214, 292-310) + ObservationUtils.get_relative_state / get_observation
 * (pkg/observation_utils.py:77-158) replayed for n_series independent scripted series of n_ticks 100 Hz ticks, with the
 * device functions the fused step kernel runs.  in double[n_series][n_ticks][14] = drone p(3), v(3), quaternion w x y z (world
 * frame), platform x y u v as reported at that tick; contact uint8[n_series][n_ticks]; out double[n_series][n_ticks][12] =
 * observation p_x p_y v_x v_y a_x a_y, the v_z and yaw PID plant states, the platform set-point x y u v published by the tick.
 * Noise (cfg noise_*_sd > 0) comes from the Philox stream of (seed, series). */
int dql_manager_run(const dql_config* cfg, int device, int64_t n_series, int64_t n_ticks, const double* in, const uint8_t* contact,
                    uint64_t seed, double* out);
/* The plant of the fused step — what replaces Gazebo's motor-model plugin + ODE — replayed OPEN LOOP for n_series independent
 * series of n_ticks 500 Hz physics ticks, with the device functions the step kernel runs, in the kernel's order: rotation matrix of
 * the attitude quaternion; forces and moments of the CURRENT rotor speeds (thrust, rotor drag, rolling moment, drag torque:
 * rotors_gazebo_plugins/src/gazebo_motor_model.cpp:434-482) and one semi-implicit Euler step of the rigid body (dt, g:
 * worlds/basic.world:36-73); first-order rotor filter towards min(command, max_rot_velocity) (common.h:147-183,
 * gazebo_motor_model.cpp:358-364); platform extrapolation and the bumper-contact latch.  No controller, no MDP.
 * init double[n_series][21] = drone p(3), v(3), quaternion w x y z, body rates(3), rotor speeds(4), platform x y u v;
 * rotor_cmd double[n_series][n_ticks][4] commanded rotor speeds (rad/s, >= 0); out double[n_series][n_ticks][20] = p(3), v(3),
 * quaternion(4), body rates(3), rotor speeds(4), platform x y, contact latch (0 / 1) AFTER each tick.  Exists so that tests can hold
 * the plant against closed forms (tests/test_plant_closed_forms.py); Gazebo itself cannot run here (parity vs Gazebo: unpinned). */
int dql_plant_run(const dql_config* cfg, int device, int64_t n_series, int64_t n_ticks, const double* init, const double* rotor_cmd, double* out);
/* ---- the control-side functions of the 500 Hz / 100 Hz ticks replayed ALONE, one call per reference class method, with the device functions
 * the fused step kernel runs (float64: the reference's expressions operation by operation; float32: the forms every throughput figure runs on).
 * They exist so that each function can be held against the reference's own outputs (tests/golden G8, G9, G11) in BOTH dtypes ---- */
/* ButterworthFilter.update (pkg/filters.py:98-109) over x[n] from zero histories, c = cfg->bw_c */
int dql_butterworth_run(const dql_config* cfg, int device, const double* x, int64_t n, double* y_out);
/* KalmanFilter3D.filter (pkg/filters.py:53-80) over a velocity series vel[n][3] sampled at t = 0.01 i: z = dv / dt, dt <= 0 -> 0.01 (dt_le0[i] != 0
 * forces that branch), Q = cfg->kalman_q, R = cfg->noise_vel_sd^2 -> acc_out[n - 1][3] */
int dql_kalman_run(const dql_config* cfg, int device, const double* vel, const uint8_t* dt_le0, int64_t n, double* acc_out);
/* PID.output (pkg/pid.py:62-104) replayed over n 500 Hz ticks at t = 0.002 (i + 1); params[7] = Kp Ki Kd lower upper windup setpoint (Kd must be 0:
 * launch/drone.launch:37,51), state[n] sampled every 5th tick as the 100 Hz manager publishes it -> control effort and integral per tick */
int dql_pid_run(const dql_config* cfg, int device, const double* params, const double* state, int64_t n, double* effort_out, double* integral_out);
/* AttitudeController.compute_rotor_velocities (pkg/attitude_controller.py:107-156) for n samples: quat_xyzw[n][4] (ROS order), omega[n][3] body rates,
 * cmd[n][4] = roll, pitch, yaw rate, thrust -> rotor_out[n][4] commanded rotor speeds (rad/s).  xonly = 1 (float32 only, every roll command
 * exactly 0): the closed form the x-axis kernels compile in */
int dql_attitude_run(const dql_config* cfg, int device, const double* quat_xyzw, const double* omega, const double* cmd, int64_t n, int32_t xonly,
                     double* rotor_out);
/* MovingPlatform.compute_trajectory (pkg/moving_platform.py:87-127) from phase 0 -> out[n][4] = x, y, u, v at successive 100 Hz ticks.  carry > 0
 * (float32 only): sine / cosine evaluated at every carry-th tick and rotated through the constant phase step in between, as the fused float32
 * step carries them through the four or five manager ticks of an agent period */
int dql_platform_run(const dql_config* cfg, int device, int64_t n, int32_t carry, double* out);
/* start coordinate of the drone along one axis for n (random offset x0, platform coordinate) pairs: the placement arithmetic of
 * TrainingLandingEnv.reset / SimulationLandingEnv.reset selected by cfg->init_uniform (see dql_config) */
int dql_place(const dql_config* cfg, int device, const double* x0, const double* mp, int64_t n, double* out);
/* ---- a DoubleQLearningAgent's tables RESIDENT on the device (pkg/double_q_learning.py:32-146) ----
 * The stateless dql_agent_predict / dql_agent_update below ship all three tables (3 x 22 680 B) both ways per call — fine for a batch,
 * 6x slower than the reference's own Python for a caller that steps ONE env (BASELINE configs[0]).  A dql_agent keeps them in device
 * memory between calls; arguments and results travel through pinned host memory the kernels read and write directly (one launch
 * and one wait per call, no allocation, no copy engine).  dql_agent_update_resident returns, per transition, the new value of the
 * updated cell and of its visit counter, so the caller can patch its host copy of the tables instead of fetching them. */
typedef struct dql_agent dql_agent;
int dql_agent_create(int device, dql_agent** out);
int dql_agent_destroy(dql_agent* agent);
int dql_agent_set_tables(dql_agent* agent, const double* qa_or_null, const double* qb_or_null, const double* count_or_null);
int dql_agent_get_tables(dql_agent* agent, double* qa_or_null, double* qb_or_null, double* count_or_null);
int dql_agent_predict_resident(dql_agent* agent, const int32_t* idx, int64_t n, uint8_t* action_out);
int dql_agent_update_resident(dql_agent* agent, const int32_t* sa, const int32_t* ns, const double* alpha, double gamma, const double* reward,
                              int64_t n, uint32_t quirks, const uint8_t* coin_or_null, const uint8_t* done_or_null, double* q_new_out,
                              double* count_new_out, uint8_t* next_action_out_or_null /* predict(ns[n-1]) on the updated tables: the
                              reference's loop (pkg/trainer.py:191-212) asks for exactly that next, and gets it without a second round trip */);
/* The resident agent driven the way the reference's loop drives it (pkg/trainer.py:191-212) — ONE transition per call — against tables
 * that live in the CALLER's arrays (DoubleQLearningAgent.Q_table_a / Q_table_b / state_action_counter: float64, C order,
 * [n_levels][3][3][3][7][3], public and writable at any time).  Every call first compares the caller's three arrays with what the device
 * holds (memcmp against a pinned host shadow, ~1 us per table) and uploads what differs: writes between calls are honoured without a
 * dirty flag.  dql_agent_mirror_update applies pkg/double_q_learning.py:91-146 to the one transition (sa = cell index idx * 3 + action,
 * ns = packed next state; quirks / coin / done as in dql_agent_update) and patches the ONE changed cell and its visit counter into the
 * caller's arrays itself; it also keeps predict(ns) on the updated tables, which dql_agent_mirror_predict hands out without a device round
 * trip when asked for exactly that state next (as the reference's loop does) and the tables were not written in between.
 * Levels n_levels..4 of the device tables are zero.  DQL_EINVAL: null array, n_levels outside 1..5, index outside the n_levels levels. */
int dql_agent_mirror_predict(dql_agent* agent, const double* qa, const double* qb, const double* count, int32_t n_levels, int32_t idx, uint8_t* action_out);
int dql_agent_mirror_update(dql_agent* agent, double* qa, double* qb, double* count, int32_t n_levels, int32_t sa, int32_t ns, double alpha,
                            double gamma, double reward, uint32_t quirks, int32_t coin, int32_t done);
/* The same update in two halves (ABI v6): _deferred launches the kernel and returns — the caller's arrays still hold the OLD cell; dql_agent_mirror_complete, or the next
 * call of any kind on this agent, waits for the kernel and patches the cell and its visit counter in (into the arrays that were passed to _deferred: they must stay alive
 * until then).  For a host loop that has other work between update() and the next guess() (the reference's has: pkg/trainer.py:204-212) the kernel's round trip hides
 * behind that work.  dql_agent_mirror_update == _deferred + _complete. */
int dql_agent_mirror_update_deferred(dql_agent* agent, double* qa, double* qb, double* count, int32_t n_levels, int32_t sa, int32_t ns, double alpha,
                                     double gamma, double reward, uint32_t quirks, int32_t coin, int32_t done);
int dql_agent_mirror_complete(dql_agent* agent);
/* DoubleQLearningAgent.predict (pkg/double_q_learning.py:119-124) for n packed states */
int dql_agent_predict(int device, const double* qa, const double* qb, const int32_t* idx, int64_t n, uint8_t* action_out);
/* DoubleQLearningAgent.update (pkg/double_q_learning.py:91-146) replayed strictly in order for n transitions:
 * sa int32[n] = cell index (idx*3+action), ns int32[n] = packed next state; tables updated in place.
 * quirks: DQL_Q_UPDATE_TABLE_A_ONLY (B1/B2) = the reference: always Q_table_a, valued by itself; cleared = Double Q-learning:
 * coin[i] (required then: 0 -> Q_table_a, 1 -> Q_table_b; the caller draws it where the reference draws its unused uniform)
 * picks the table to update and the OTHER table values the picked table's greedy action at s'.
 * DQL_Q_BOOTSTRAP_ON_POS_CHANGE (B3) = bootstrap only when the position bin changed; cleared = bootstrap unless done[i]
 * (required then).  Other quirk bits do not concern the agent and are ignored. */
int dql_agent_update(int device, double* qa, double* qb, double* count, const int32_t* sa, const int32_t* ns,
                     const double* alpha, double gamma, const double* reward, int64_t n, uint32_t quirks,
                     const uint8_t* coin_or_null, const uint8_t* done_or_null);

/* DoubleQLearningAgent.transfer_learning (pkg/double_q_learning.py:77-89) on host tables: Q[k] = Q[k-1] * ratio (k = 0 wraps, B6) */
int dql_agent_transfer(int device, double* qa, double* qb, int32_t k, double ratio);

#ifdef __cplusplus
}
#endif
#endif /* DQL_H */
