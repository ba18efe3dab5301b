/*
 * trx_debug.h -- switches and probes of the TESTING build of libtrx (libtrx_testing.so: the sources of libtrx.so
 * compiled with -DTRX_TESTING by __graft_entry__.build()).
 *
 * NOT part of the production ABI: libtrx.so exports none of these symbols, reads nothing from the environment and
 * has no process-wide mutable state (include/trx.h, Conventions).  Everything here is process-wide: a switch flipped
 * by one host thread changes what another thread's in-flight call enqueues.  That is acceptable for tests/ and the
 * A/B scripts under profiles/, which use them to force a code path -- a number of rows per wave, the one-row kernel on
 * a short light curve, a re-enabled bug for the "never written" guard -- and for nothing else.  With every switch at
 * its default the testing library enqueues exactly what the production library does.
 *
 * Environment variables (testing build only; read once): TRX_PROBE_ROWS, TRX_BOUNDED, TRX_STAR_CHAIN give three of
 * the switches below their initial value; TRX_PROBE_CELLS, TRX_THIRD_STRIDE, TRX_GRID_CAP, TRX_GRID_CAP_PLAIN,
 * TRX_GRID_CAP_LONG, TRX_ROWC_CAP, TRX_SCAN_CAP, TRX_WAVE_FLOOR, TRX_WAVE_FLOOR3, TRX_CHAIN_DRAWS override launch
 * geometry constants (experiments; the values the production library has compiled in are the measured best).
 */
#ifndef TRX_DEBUG_H
#define TRX_DEBUG_H

#include "trx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rows staged per wavefront of the batched kernel (light curves below trx_set_cell_packing_below's threshold),
 * 1..22; 0 = automatic (default) */
int trx_set_rows_per_wave(int rows);
/* light curves with fewer than n points (default 320) are processed in batches of rows per wave, longer ones one row
 * per wave; 0 = never.  Model values agree to rounding between the two; chi^2 differs by summation order. */
int trx_set_cell_packing_below(int n_time);
/* (host only, touches no device) checks the plan by which the batched likelihood kernel deals `rows` rows to its
 * waves at `rows_per_wave` rows each -- tapered towards the end of the launch when `taper` is set --: every row in
 * exactly one batch.  *positions: wave positions per XCD; *rows_min: the smallest batch. */
int trx_debug_batch_plan(long rows, int rows_per_wave, int taper, long* positions, int* rows_min);

/* process-wide forms of TRX_FLAG_ALL_SUBEXPOSURES (0), TRX_FLAG_NO_STENCIL (0), TRX_FLAG_EVALUATE_EXCLUDED (0),
 * TRX_FLAG_COUNT_EVALUATIONS (1); defaults 1 / 1 / 1 / 0 */
int trx_set_supersample_tiers(int on);
int trx_set_stencil(int on);
int trx_set_skip_excluded(int on);
int trx_set_debug_node_counts(int on);
/* 0 = a full Kepler solve at every node instead of Newton steps from the exposure centre's solution (default 1) */
int trx_set_kepler_stepping(int on);

/* bounded evaluation of the scenario entry points: 0 = never (TRX_FLAG_FULL_EVALUATION for every call), 1 = light
 * curves of one row per wave only, 2 = always (default) */
int trx_set_bounded_evaluation(int mode);
/* 1: trx_lnl_batch / trx_lnz_scenario treat their rows the bounded way too, as for an evidence without prior: a row
 * then holds its chi^2/2 or, if abandoned, a lower bound of it that exceeds the call's smallest by more than 90 */
int trx_set_debug_bounded_lnl(int on);
/* 1: trx_scenario_enqueue / trx_star_enqueue zero the chi^2 arrays of a call before its likelihood kernels run, so
 * that a row no kernel writes shows as a perfect fit instead of as whatever the stream's previous call left there */
int trx_set_debug_poison(int on);
/* 1: re-enables a bug of round 4 (the third pass of the bounded evaluation skipped its last batches when nothing was
 * probed), so that a test can show the "never written" status of the record catching it */
int trx_set_debug_bug(int on);
/* rows per wave of the probe pass of the bounded evaluation of batched light curves: 0 = as many as its LDS layout
 * holds (default), 1 = as many as the other passes take, n > 1 = n (at most 22).  Results do not depend on it. */
int trx_set_probe_rows(int rows);
/* 0: trx_star_enqueue enqueues its calls one by one instead of in launch chains (same records, bit for bit) */
int trx_set_star_chain(int on);

/* scratch buffers of calls captured into hipGraphs: how many a live graph still owns, how many wait in the library's
 * pool for reuse (trx_release_scratch frees those) */
int trx_debug_capture_buffers(long* live, long* idle);

#ifdef __cplusplus
}
#endif
#endif /* TRX_DEBUG_H */
