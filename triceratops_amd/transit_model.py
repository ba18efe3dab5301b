"""Drop-in for the three pytransit.QuadraticModel call shapes the reference uses
(likelihoods.py:24-25 ctor; set_data 61/120/135/348/414/421; evaluate_ps 62-71/124-144;
evaluate_pv 349/415/422), backed by trx_flux_grid(TRX_MODEL_RAW).

Installing this class as `pytransit.QuadraticModel` lets the unmodified reference run on the
MI355X kernels."""
import numpy as np

from . import _lib


class QuadraticModel:
    def __init__(self, interpolate: bool = False, **_ignored):
        if interpolate:
            raise NotImplementedError("only the direct (interpolate=False) model the reference uses")
        self.time = None
        self._time_d = None
        self.nsamples = 1
        self.exptime = 0.0

    def set_data(self, time, lcids=None, pbids=None, nsamples=None, exptimes=None, epids=None):
        self.time = np.ascontiguousarray(time, dtype=np.float64)
        self._time_d = _lib.dev(self.time)
        self.nsamples = int(np.ravel(nsamples)[0]) if nsamples is not None else 1
        self.exptime = float(np.ravel(exptimes)[0]) if exptimes is not None else 0.0

    def evaluate_pv(self, pvp, ldc):
        """pvp (n,7) = [k, t0, p, a, i, e, w], ldc (n,2) -> flux (n, n_time)."""
        pvp = np.atleast_2d(np.asarray(pvp, dtype=np.float64))
        ldc = np.atleast_2d(np.asarray(ldc, dtype=np.float64))
        rows = np.ascontiguousarray(np.concatenate([pvp[:, :7].T, ldc[:, :2].T], axis=0))
        grid, _ = _lib.flux_grid(_lib.MODEL_RAW, 0, self._time_d, _lib.dev(rows), self.exptime,
                                 self.nsamples, want_secdepth=False)
        return grid.cpu().numpy()

    def evaluate_ps(self, k, ldc, t0, p, a, i, e=0.0, w=0.0):
        """scalar parameters -> flux (n_time,)"""
        pvp = np.array([[k, t0, p, a, i, e, w]], dtype=np.float64)
        return self.evaluate_pv(pvp, np.asarray(ldc, dtype=np.float64).reshape(1, 2))[0]
