from pytorch_fem_solver_amd.basis import *  # noqa: F401,F403
