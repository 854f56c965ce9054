import numpy as np
# The reference stores this float64 array as an unused buffer (dipole=False everywhere).
# Values are irrelevant to the hot path; only dtype/shape (119,) matter for state_dict shape.
atomic_masses = np.arange(119, dtype=np.float64)
