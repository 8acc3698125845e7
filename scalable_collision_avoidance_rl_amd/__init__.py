"""MI355X-native batched drone_env hot path (step / reset / get_local_states).

Drop-in for `/root/reference/drone_env.py` class `drones`; see drone_env.py in this
package, include/dronesim.h (C ABI), csrc/drone_kernel.hpp (the gfx950 step / observe / rollout kernel),
csrc/dronesim.hip (the ABI's entry points + the small kernels) and csrc/policy.hip (batched policies).
Importing the package needs neither a GPU nor the built library; constructing an
environment needs both (no CPU fallback)."""
from .drone_env import (DroneState, StepResult, clip_deltas, dim, drones, dt, formation_O, gradient_control,
                        lattice_divisions, max_time_steps, proportional_control, shard_range)

__all__ = ["drones", "DroneState", "StepResult", "dim", "dt", "max_time_steps", "formation_O",
           "clip_deltas", "lattice_divisions", "shard_range", "gradient_control", "proportional_control"]
