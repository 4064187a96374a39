"""Developer probe: Jacobi residual traces for the SVDs of the replica set-up (run with DQMC_DEBUG_SVD=1)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from detqmc_amd import DetSDW, SDWParams
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
beta = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rep = DetSDW(SDWParams(opdim=2, L=L, beta=beta, s=10, delaySteps=16))
rep.sweepThermalization()
