/*
 * oracle/bench_window_skip.h -- NOT part of the checker.
 *
 * Included by trx_oracle.c only under -DTRXO_WINDOW_SKIP, which only bench.py's native_oracle() passes
 * (the cpu_baseline leg "window_early_out").  It is the one shortcut of the GPU kernels that a CPU
 * implementation would take as well -- the builder's algorithm, not the reference's -- so it lives outside
 * the file whose job is to restate the reference: the default build of the oracle (oracle/Makefile, what
 * tests/ and smoke() load) does not contain it.
 */
/* cpu_baseline leg of bench.py only: 1 = an exposure whose centre lies outside the transit window returns 1
 * without touching the orbit (the early-out the GPU kernels have), so that one CPU figure
 * is like for like with the GPU's algorithm.  The plain restatement (0, default) evaluates every point, as the
 * reference does. */
static int trxo_window_skip = 0;
static double trxo_exptime_hint = 0.0;      /* exposure the window's margin is taken for (set per call) */
void trxo_set_window_skip(int on) { trxo_window_skip = on ? 1 : 0; }

/* Window: X(E) = ax (cosE - e) + bx sinE = A cos(E - phi) - ax e; an occultation needs |X| < 1 + k.  The arc of
 * E around conjunction on which that holds maps to a mean-anomaly interval; when the mirror arc can lie on the
 * near side too (very eccentric orbits seen along the major axis) the window is the hull of both. */
static void trxo_orbit_window(trxo_orbit* o, double Etr, double exptime)
{
    const double e = o->e, R = (1.0 + o->k) * (1.0 + 1e-9) + 1e-12;
    const double A = sqrt(o->ax * o->ax + o->bx * o->bx), phi = atan2(o->bx, o->ax);
    const double clo = (o->ax * e - R) / A, chi = (o->ax * e + R) / A;
    const double psi = remainder(Etr - phi, TRXO_TWOPI);
    const int open_hi = !(chi < 1.0), open_lo = !(clo > -1.0);
    const double xlo = acos(fmin(fmax(clo, -1.0), 1.0)), xhi = acos(fmin(fmax(chi, -1.0), 1.0));
    double plo, phi2;
    if (open_hi && open_lo) { plo = psi - TRXO_TWOPI; phi2 = psi + TRXO_TWOPI; }
    else if (open_hi) { plo = -xlo; phi2 = xlo; }
    else if (open_lo) { if (psi >= 0.0) { plo = xhi; phi2 = TRXO_TWOPI - xhi; } else { plo = xhi - TRXO_TWOPI; phi2 = -xhi; } }
    else {
        if (psi >= 0.0) { plo = xhi; phi2 = xlo; } else { plo = -xlo; phi2 = -xhi; }
        const double o1 = (psi >= 0.0) ? -xlo : xhi, o2 = (psi >= 0.0) ? -xhi : xlo;
        const double Ay = sqrt(o->ay * o->ay + o->by * o->by), phy = atan2(o->by, o->ay);
        const double d1 = remainder(Etr + (o1 - psi) - phy, TRXO_TWOPI), span = o2 - o1;
        const int peak = (d1 <= 0.0 && d1 + span >= 0.0) || (d1 + span >= TRXO_TWOPI);
        const double cmax = peak ? 1.0 : fmax(cos(d1), cos(d1 + span));
        if (!(Ay * cmax - o->ay * e < -1e-9 * (Ay + fabs(o->ay * e)))) { plo = fmin(plo, o1); phi2 = fmax(phi2, o2); }
    }
    const double Elo = Etr + (plo - psi), Ehi = Etr + (phi2 - psi);
    const double mg = 0.5 * fabs(o->n * exptime) * (1.0 + 1e-9) + 1e-11;
    o->wlo = (Elo - Etr) - e * (sin(Elo) - sin(Etr)) - mg;
    o->whi = (Ehi - Etr) - e * (sin(Ehi) - sin(Etr)) + mg;
}


static int trxo_outside_window(const trxo_orbit* o, double t)
{
    const double ph = o->n * (t - o->t0);
    const double dM = remainder(ph, TRXO_TWOPI), slack = 1e-15 * fabs(ph);
    return ((dM < o->wlo - slack) && !(dM + TRXO_TWOPI <= o->whi + slack)) ||
           ((dM > o->whi + slack) && !(dM - TRXO_TWOPI >= o->wlo - slack));
}
