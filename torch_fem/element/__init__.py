from pytorch_fem_solver_amd.element import *  # noqa: F401,F403
