"""Provenance of tests/golden/wo_contour_reference.npz: a verbatim copy of the reference project's own data file
data/data_contour_WilliamOttoReactor.npz (five plain fp64 arrays X_0, X_1, Y_objective, Y_constraint1, Y_constraint2, each
100 x 100 over [4, 7] x [70, 100]; written by the reference's utils/utils_WilliamOttoReactor.py:24-55 with its plant
problems/WilliamOttoReactor_Problem.py and SciPy fsolve).  It is data -- inputs and expected outputs -- and pins the
oracle's plant restatement (oracle/plants.py); run this script in a container that has /root/reference to refresh it."""
import shutil
import os

SRC = "/root/reference/data/data_contour_WilliamOttoReactor.npz"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "wo_contour_reference.npz")

if __name__ == "__main__":
    shutil.copyfile(SRC, DST)
    print("copied", SRC, "->", DST)
