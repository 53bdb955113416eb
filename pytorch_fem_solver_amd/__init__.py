"""MI355X-native element-wise FEM assembly behind the torch_fem API (work in progress)."""
