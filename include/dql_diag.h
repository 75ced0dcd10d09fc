/*
 * dql_diag.h — measurement and self-test entry points of libdql_hip.so.  NOT part of the drop-in boundary: nothing in the reference maps onto
 * these, the host classes (TrainingMdp, DoubleQLearningAgent, TrainingLandingEnv, Trainer) never call them, and a maintainer who swaps the
 * library in behind the reference's classes needs none of them (INTEGRATION.md section 3 lists what IS needed).  They exist for bench.py, the
 * profiling tools under tools/ and the self-tests in tests/; same conventions as dql.h (0 / negative dql_status, dql_last_error()).
 */
#ifndef DQL_DIAG_H
#define DQL_DIAG_H

#include "dql.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- stream timers (HIP events on the context's stream; torch.cuda.Event would only see torch's stream) ---- */
int dql_diag_timer_start(dql_ctx* ctx);                    /* hipEventRecord on the ctx stream */
int dql_diag_timer_stop(dql_ctx* ctx, double* elapsed_ms); /* records, synchronises, returns elapsed */
/* arm / disarm per-launch event pairs around the fused step kernel and around every table exchange */
int dql_diag_kernel_timer(dql_ctx* ctx, int32_t on);
/* average device duration of the fused step kernel over the launches made while the kernel timer was armed */
int dql_diag_kernel_time_ms(dql_ctx* ctx, double* avg_ms, int64_t* launches);
/* average device time of the exchanges (all-reduce or peer-to-peer push / wait / sum, + fold) made while the kernel timer was armed */
int dql_diag_sync_time_ms(dql_ctx* ctx, double* avg_ms, int64_t* syncs);
/* holds the context's stream for this long (a one-wave timer kernel): phase offset between contexts that share a GPU (tools/exp_cohorts.py) */
int dql_diag_delay(dql_ctx* ctx, double microseconds);
/* the window accumulators' device buffer (4 * 2835 int64), for a caller that wants to reduce it with a collective of its own; the product path
 * never needs the pointer (dql_allreduce_window / dql_p2p_exchange_window reduce it in place) */
int dql_diag_accum_dev_ptr(dql_ctx* ctx, void** dev_ptr, int64_t* n_int64);

/* ---- self-test ---- */
/* The float32 tick's square root (csrc/dql_device.hpp sqrt_pos: v_rsq_f32 + one residual correction; until the end of round 5 with a Goldschmidt step in between): counts the inputs with bit
 * patterns lo_bits .. hi_bits whose result is NOT the correctly rounded sqrt.  The CPU oracle computes sqrtf(); parity is bit for bit only while
 * this count is 0 on the tick's domain [1e-30, FLT_MAX] — all 2.1e9 inputs take under a second. */
int dql_diag_selftest_sqrt(int device, uint32_t lo_bits, uint32_t hi_bits, int64_t* not_correctly_rounded);

#ifdef __cplusplus
}
#endif
#endif /* DQL_DIAG_H */
