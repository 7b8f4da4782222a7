import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import storm_amd
n_int = 100
sys_ = dict(u=np.full(n_int, 3.0), v_gamma=np.zeros(n_int), a_star=np.zeros(n_int), c1=np.ones(n_int),
            rot=np.zeros(n_int), delta_over_x=np.full(n_int, 0.01), inner_c1=1.0, inner_rot=0.0, outer_u=3.0)
print("creating", flush=True)
h = storm_amd.HipSolver("colloc2", 0, 0, sys_, n_int, 1)
print("created", flush=True)
print(h.det(np.array([1.0, 2.0])))
