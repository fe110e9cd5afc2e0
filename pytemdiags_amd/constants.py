"""Physical constants of the TEM formulation (DynVarMIP, Gerber & Manzini 2016, section A2).

Values match PyTEMDiags/constants.py:6-14 digit for digit because they are part of parity --
including ``pi = 3.14159``, the truncated value the reference uses in psitem only
(tem_diagnostics.py:674).  The device epilogue hard-codes the same
numbers (csrc/kernels.hpp, tem_epilogue_kernel).
"""
P0 = 101325      # reference surface pressure [Pa]
R = 287.058      # gas constant of dry air [J/K/kg]
Cp = 1004.64     # specific heat of dry air at constant pressure [J/K/kg]
g0 = 9.80665     # gravity at mean sea level [m/s^2]
a = 6.37123e6    # Earth radius [m]
Om = 7.29212e-5  # Earth rotation rate [1/s]
k = R / Cp       # kappa
H = 7 * 1e3      # scale height [m]
pi = 3.14159     # (sic) used by psitem only
