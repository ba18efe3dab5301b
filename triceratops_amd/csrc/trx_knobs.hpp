// Tuning and diagnostics switches of libtrx (not part of the C ABI of include/trx.h).
//
// The PRODUCTION library (libtrx.so) has none: every switch below is a compile-time constant there, no setter is
// exported, nothing is read from the environment -- the library's behaviour is a function of a call's arguments alone
// ("no global state, safe to call concurrently").  What a caller may legitimately choose per call is a
// TRX_FLAG_* bit of that call (include/trx.h: TRX_FLAG_ALL_SUBEXPOSURES, TRX_FLAG_NO_STENCIL, TRX_FLAG_COUNT_EVALUATIONS,
// TRX_FLAG_FULL_EVALUATION, TRX_FLAG_EVALUATE_EXCLUDED).
//
// The TESTING library (libtrx_testing.so: the same sources with -DTRX_TESTING, built next to it by
// __graft_entry__.build()) turns them into process-wide atomics with the setters of include/trx_debug.h and the
// experiments' environment variables: what tests/ and profiles/ use to force a code path (rows per wave, the one-row
// kernel on short light curves, a re-enabled bug for the "never written" guard ...).  Flipping one while another
// thread enqueues is a test's own business; it cannot happen to the production library.
#pragma once
#include <stdlib.h>

#ifdef TRX_TESTING
#include <atomic>
#endif

namespace trx {

#ifdef TRX_TESTING
inline int knob_env_int(const char* name, int dflt)
{
    const char* e = name ? getenv(name) : nullptr;
    return (e && *e) ? atoi(e) : dflt;
}
inline long env_long(const char* name, long dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atol(e) : dflt;
}
inline double env_double(const char* name, double dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atof(e) : dflt;
}
#define TRX_KNOB(name, dflt, env)                                              \
    inline std::atomic<int> g_knob_##name{knob_env_int(env, dflt)};           \
    inline int knob_##name() { return g_knob_##name.load(std::memory_order_relaxed); }
#else
constexpr long env_long(const char*, long dflt) { return dflt; }
constexpr double env_double(const char*, double dflt) { return dflt; }
#define TRX_KNOB(name, dflt, env) \
    constexpr int knob_##name() { return dflt; }
#endif

// name, default, environment variable that gives the testing library its initial value (or nullptr)
TRX_KNOB(rows_per_wave, 0, nullptr)        // rows per wave of the batched kernel, 0 = automatic (batch_rows)
TRX_KNOB(probe_rows, 0, "TRX_PROBE_ROWS")  // rows per wave of the probe pass: 0 = what its LDS holds, 1 = as the other passes
TRX_KNOB(kepler_stepping, 1, nullptr)      // Newton steps from the exposure centre (0: a full solve per sub-exposure)
TRX_KNOB(tiers, 1, nullptr)                // Gauss-node tiers for the exposure average (0: every sub-exposure)
TRX_KNOB(bounded, 2, "TRX_BOUNDED")        // bounded evaluation of scenario calls: 0 never, 1 one row per wave only, 2 always
TRX_KNOB(bounded_lnl, 0, nullptr)          // trx_lnl_batch / trx_lnz_scenario apply it too (tests)
TRX_KNOB(skip_excluded, 1, nullptr)        // rows the EB secondary rule excludes are not evaluated
TRX_KNOB(stencil, 1, nullptr)              // centre-value stencil on dense uniform time grids
TRX_KNOB(debug_nodes, 0, nullptr)          // trx_flux_grid writes evaluation counts
TRX_KNOB(cells_below, 320, nullptr)        // light curves shorter than this: batches of rows per wave
TRX_KNOB(debug_bug, 0, nullptr)            // re-enables round 4's exit-rule bug (the "never written" guard's test)
TRX_KNOB(poison, 0, nullptr)               // chi^2 arrays zeroed before the likelihood kernels of a scenario call
TRX_KNOB(star_chain, 1, "TRX_STAR_CHAIN")  // launch chains in trx_star_enqueue

#undef TRX_KNOB

}  // namespace trx
