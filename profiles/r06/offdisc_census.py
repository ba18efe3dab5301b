"""Host-side census (numpy, no GPU) of what the contact cells of BASELINE config 1's rows consist of: for a sample of
synthetic rows, which cells find no admissible Gauss tier (cells_kernel's second sweep: all S sub-exposures), how many
of their sub-exposures are truly off the disc, and how many offdisc_nodes() (csrc/trx_device.hpp) classifies as such.
Restates the plan's criteria (plan_cell) in numpy; pytransit-shaped parameters from synth.tp_rows / eb_rows."""
import sys
import numpy as np
sys.path.insert(0, ".")
from triceratops_amd import synth
from triceratops_amd.constants import Rsun, Rearth

S, EXPT = 20, synth.EXPTIME
t = synth.time_grid(2000)
rng = np.random.default_rng(1)


def kepler(M, e):
    E = M + e * np.sin(M)
    for _ in range(60):
        E = E - (E - e * np.sin(E) - M) / (1 - e * np.cos(E))
    return E


def census(k, P, a, inc, e, w, name):
    tot = dict(cells=0, inwin=0, heavy=0, nodes=0, off_true=0, off_mask=0, nostmt=0, heavy_alloff=0, heavy_contact14=0, heavy_contact23=0, heavy_none=0)
    for r in range(len(k)):
        n = 2 * np.pi / P[r]
        ww = np.deg2rad(w[r]); ii = np.deg2rad(inc[r])
        rt = np.sqrt(1 - e[r] ** 2)
        ftr = np.pi / 2 - ww
        Etr = np.arctan2(rt * np.sin(ftr), e[r] + np.cos(ftr))
        Mtr = Etr - e[r] * np.sin(Etr)
        ax, bx, ay, by = a[r] * np.cos(ww), -a[r] * rt * np.sin(ww), a[r] * np.sin(ww), a[r] * rt * np.cos(ww)
        cosi = np.cos(ii)

        def orbit(tt):
            E = kepler(n * tt + Mtr, e[r])
            sE, cE = np.sin(E), np.cos(E)
            rho = 1 / (1 - e[r] * cE)
            X = ax * (cE - e[r]) + bx * sE; Y = ay * (cE - e[r]) + by * sE
            nr = n * rho
            Xp = (bx * cE - ax * sE) * nr; Yp = (by * cE - ay * sE) * nr
            return X, Y, Xp, Yp, rho, nr
        X, Y, Xp, Yp, rho, nr = orbit(t)
        yc, ycp = Y * cosi, Yp * cosi
        z2 = X * X + yc * yc
        opp2, omk2 = (1 + k[r]) ** 2, (1 - k[r]) ** 2
        g1 = np.abs(2 * (X * Xp + yc * ycp)); g2 = np.abs(Xp * Xp + ycp * ycp - nr * nr * rho * z2)
        G = np.minimum(np.abs(z2 - opp2), np.abs(z2 - omk2))
        kd = min(k[r], 1.0); hx = 0.5 * EXPT * (1 + 0.5 * kd * kd); om = np.abs(nr) * rho

        def adm(radius):
            tau = radius * hx
            return (G >= 1.25 * (g2 * tau * tau + g1 * tau)) & (om * tau <= 0.15) & (Y > np.abs(Yp) * tau)
        off = (z2 > opp2) & (G >= 1.25 * (g2 * (1.5 * hx) ** 2 + g1 * 1.5 * hx)) & (om * 1.5 * hx <= 0.15)
        front = Y > 0
        inwin = front & (np.abs(X) < 1 + k[r] + 0.2)
        heavy = inwin & ~adm(1.8) & ~off
        tot["cells"] += len(t); tot["inwin"] += int((inwin & ~off).sum()); tot["heavy"] += int(heavy.sum())
        idx = np.nonzero(heavy)[0]
        if len(idx) == 0:
            continue
        fr = (np.arange(S) + 0.5) / S - 0.5
        tt = t[idx, None] + EXPT * fr[None, :]
        Xs, Ys, _, _, _, _ = orbit(tt)
        z2s = Xs * Xs + (Ys * cosi) ** 2
        offt = ~((Ys >= 0) & (z2s < opp2))
        tot["nodes"] += offt.size; tot["off_true"] += int(offt.sum())
        tot["heavy_alloff"] += int(offt.all(axis=1).sum())
        zmin, zmax = z2s.min(axis=1), z2s.max(axis=1)
        c14 = (zmin < opp2) & (zmax > opp2); c23 = (zmin < omk2) & (zmax > omk2)
        tot["heavy_contact14"] += int(c14.sum()); tot["heavy_contact23"] += int((c23 & ~c14).sum())
        tot["heavy_none"] += int((~c14 & ~c23 & ~offt.all(axis=1)).sum())
        # offdisc_nodes
        kap = nr[idx] ** 2 * rho[idx]
        f1 = 2 * (X[idx] * Xp[idx] + yc[idx] * ycp[idx]); f2h = Xp[idx] ** 2 + ycp[idx] ** 2 - kap * z2[idx]
        tau = 0.5 * EXPT
        stmt = (om[idx] * tau <= 0.15) & (Y[idx] > np.abs(Yp[idx]) * tau)
        A1 = np.abs(f1) + 2 * np.abs(f2h) * tau; fm = z2[idx] + tau * (np.abs(f1) + np.abs(f2h) * tau)
        F3 = kap * (3 * A1 + 6 * e[r] * om[idx] * fm)
        d = EXPT * fr[None, :]
        q = z2[idx, None] - opp2 - 1e-9 + d * (f1[:, None] + d * f2h[:, None]) - np.abs(d) ** 3 * F3[:, None]
        m = (q > 0) & stmt[:, None]
        assert not (m & ~offt).any(), "a sub-exposure inside the disc classified as off it"
        tot["off_mask"] += int(m.sum()); tot["nostmt"] += int((~stmt).sum())
    print("%-6s rows %d: in-window cells %.1f per row, contact cells %.1f per row (no statement possible: %.2f); of their sub-exposures "
          "%.1f %% truly off the disc, %.1f %% classified; cells: contact 1/4 %.1f, contact 2/3 only %.1f, all off %.1f, none of these %.1f per row"
          % (name, len(k), tot["inwin"] / len(k), tot["heavy"] / len(k), tot["nostmt"] / len(k), 100 * tot["off_true"] / max(tot["nodes"], 1),
             100 * tot["off_mask"] / max(tot["nodes"], 1), tot["heavy_contact14"] / len(k), tot["heavy_contact23"] / len(k),
             tot["heavy_alloff"] / len(k), tot["heavy_none"] / len(k)))


N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = synth.tp_rows(rng, N)
census(r[0] * Rearth / (r[4] * Rsun), r[1], r[3] / (r[4] * Rsun), r[2], r[7], r[8], "TP")
r = synth.eb_rows(rng, N)
census(r[0] / r[5], r[2], r[4] / (r[5] * Rsun), r[3], r[8], r[9], "EB")
r = synth.eb_rows(rng, N, twin=True)
census(r[0] / r[5], r[2], r[4] / (r[5] * Rsun), r[3], r[8], r[9], "EBx2P")
