"""Kinds and tolerances of the reference (src/Constants.f90:16-48)."""
import numpy as np

dp = np.float64
cdp = np.complex128
atol_dp = 10.0 ** (-15)            # 10**(-precision(1.0_dp)),  Constants.f90:35
rtol_dp = float(np.sqrt(atol_dp))  # Constants.f90:37
