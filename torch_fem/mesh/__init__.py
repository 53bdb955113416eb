from pytorch_fem_solver_amd.mesh import *  # noqa: F401,F403
