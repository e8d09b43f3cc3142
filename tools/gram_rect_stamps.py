#!/usr/bin/env python
"""In-kernel cycle stamps of k_gram_rect (the 128 x 256 rectangle of the Gram sweep above 256 states): where a tile period goes.

Needs a library whose mbar_k_quad.hip was compiled with -DMBAR_RECT_STAMPS (lane 0 of every wave of workgroups 0 and 133 then
prints its per-tile averages at the end of each launch):

    cd pymbar_amd/csrc
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMBAR_RECT_STAMPS -x hip -c mbar_k_quad.hip -o /tmp/quad_stamps.o
    hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libmbar_hip_stamps.so /tmp/quad_stamps.o mbar_k_eval.o mbar_k_gram.o \
          mbar_k_pmode.o mbar_k_fused.o mbar_k_solver.o mbar_capi.o mbar_loops.o mbar_comm.o mbar_host.o -ldl
    MBAR_HIP_LIBRARY=$PWD/../../scratch/libmbar_hip_stamps.so python tools/gram_rect_stamps.py 512

Output committed as profiles/r5_gram_rect_tile_anatomy.txt."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pymbar_amd import testsystems as ts  # noqa: E402
from pymbar_amd.device import DeviceMatrix  # noqa: E402

K, N = int(sys.argv[1]) if len(sys.argv) > 1 else 512, 1_000_000
O_k, K_k, N_k = ts.config3_params(K=K, N=N)
N_k[-1] += N - N_k.sum()
with DeviceMatrix.harmonic(O_k, K_k, N_k, seed=0) as dm:
    dm.set_Nk(N_k)
    dm.solve_adaptive(np.zeros(K), maxiter=2, min_sc_iter=0, check_convergence=False)
    dm.synchronize()
