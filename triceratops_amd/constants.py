"""cgs constants with the values astropy.constants gives the reference
(likelihoods.py:17-22, marginal_likelihoods.py:13-19; CODATA 2018 / IAU 2015)."""
import numpy as np

G = 6.6743e-08
Msun = 1.988409870698051e+33
Rsun = 69570000000.0
Rearth = 637810000.0
au = 14959787070000.0
pi = np.pi
ln2pi = np.log(2 * np.pi)
