"""Convenience shims for the reference's artefacts -- no throughput value, CPU-side only.

* `load_reference_modules(path)`: the reference saves whole lists of per-agent module objects with
  ``torch.save(self.criticsNN, ...)`` / ``torch.save(self.actors, ...)`` (SAC_agents.py:404-406, 580-582), so
  the pickles name its classes (``utils.CriticNN`` ...).  They are opened here WITHOUT the reference's code:
  unknown classes from its modules resolve to empty stand-ins that only carry the pickled state (for the
  networks: the ``torch.nn.Linear`` layers under the reference's attribute names), which is all
  `policies.BatchedMLP.from_*` reads.
* `plot_trajectory` / `animate_trajectory` / `show_state`: the figures of `drones.plot` / `animate` / `show`
  (drone_env.py:404-607) from state trajectories -- lists of ``[N,5]`` arrays as the reference's loop collects
  them (train_problem.py:70,104), or slices of the batched tensors (`trajectory_from_rollout`).
matplotlib is imported lazily; nothing here touches the GPU."""
from __future__ import annotations

import os
import pickle

import numpy as np

_REFERENCE_MODULES = ("utils", "SAC_agents", "drone_env", "__main__")
_standins = {}


def _standin(module, name):
    key = (module, name)
    if key not in _standins:
        import torch
        base = torch.nn.Module if name.endswith("NN") else object     # utils.py: CriticNN, NormalActorNN, DiscreteSoftmaxNN
        _standins[key] = type(name, (base,), {"__module__": f"{__name__}.standin.{module}",
                                              "__doc__": f"state-only stand-in for the reference's {module}.{name}"})
    return _standins[key]


class _ReferencePickle:
    """pickle-module lookalike for ``torch.load(pickle_module=...)``."""
    __name__ = "reference_pickle"

    class Unpickler(pickle.Unpickler):
        """Resolves ONLY what such a file legitimately names, as EXACT (module, name) pairs: torch's tensor /
        parameter rebuild helpers, its storage classes, the `torch.nn` layer classes the reference's networks are made
        of (utils.py:22-117, 271-302), `collections` containers, NumPy's array reconstruction, builtins' plain
        containers, and the reference's own classes (as stand-ins).  No module prefix is trusted: `torch.storage`,
        `torch.serialization`, `torch._utils` and `torch.nn` all hold callables that run arbitrary code when a pickle
        names them (`torch.storage._load_from_bytes` -> `torch.load(..., weights_only=False)`, `torch.serialization.load`,
        `torch.nn.utils...`).  Anything else raises instead of being imported -- a pickle is code, and a file from an
        untrusted source must still not be opened with this loader."""
        _STORAGES = tuple(f"{t}Storage" for t in ("Float", "Double", "Half", "BFloat16", "Long", "Int", "Short", "Char",
                                                  "Byte", "Bool", "ComplexFloat", "ComplexDouble", "Untyped", "Typed"))
        _ALLOWED = {("collections", "OrderedDict"), ("collections", "defaultdict"), ("collections", "deque"),
                    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "list"),
                    ("builtins", "dict"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"),
                    ("builtins", "complex"), ("builtins", "bool"), ("builtins", "slice"), ("builtins", "range"),
                    ("_codecs", "encode"),
                    ("numpy", "ndarray"), ("numpy", "dtype"),
                    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
                    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
                    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch", "Tensor"),
                    ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_tensor_v2"),
                    ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
                    ("torch._tensor", "_rebuild_from_type_v2"),
                    ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
                    ("torch.nn.parameter", "Parameter"),
                    ("torch.nn.modules.linear", "Linear"), ("torch.nn.modules.linear", "Identity"),
                    ("torch.nn.modules.activation", "ReLU"), ("torch.nn.modules.activation", "Softmax"),
                    ("torch.nn.modules.activation", "Tanh"), ("torch.nn.modules.activation", "Sigmoid"),
                    ("torch.nn.modules.container", "Sequential"), ("torch.nn.modules.container", "ModuleList"),
                    ("torch.optim.adam", "Adam")} | {("torch", s) for s in _STORAGES}

        def find_class(self, module, name):
            if module in _REFERENCE_MODULES:
                try:
                    return super().find_class(module, name)          # the real class, if the caller imported it
                except (ImportError, AttributeError):
                    return _standin(module, name)
            if module == "__builtin__":                              # protocol-2 spelling of builtins
                module = "builtins"
            if (module, name) in self._ALLOWED:
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"{module}.{name} is not something a saved list of the reference's networks "
                                         "contains; refusing to import it")

    @staticmethod
    def load(f, **kw):
        return _ReferencePickle.Unpickler(f, **kw).load()

    Pickler = pickle.Pickler
    dump, dumps, loads = pickle.dump, pickle.dumps, pickle.loads


def load_reference_modules(path):
    """List of per-agent objects from one of the reference's ``*-A2Cactors.pth`` / ``*-A2Ccritics.pth`` /
    ``*-actors.pth`` / ``*-critics.pth`` files (CPU tensors).  The unpickler imports nothing outside torch / NumPy /
    containers (see `_ReferencePickle.Unpickler`); still, only open files you trust."""
    import torch
    return torch.load(path, map_location="cpu", pickle_module=_ReferencePickle, weights_only=False)


def network_kind(module):
    """'normal_actor' | 'discrete_softmax' | 'critic' from the reference's attribute names (utils.py:22-117, 271-302)."""
    if hasattr(module, "out_2") and hasattr(module, "hidden_layer2"):
        return "normal_actor"
    if hasattr(module, "out_1"):
        return "discrete_softmax"
    if hasattr(module, "output_layer"):
        return "critic"
    raise TypeError(f"{type(module).__name__} is not one of the reference's three network classes")


# ------------------------------------------------------------------------------------------------- figures
def agent_colors(n):
    """One colour per agent along a blue -> green -> red ramp (role of drone_env.py:41-51 `num_to_rgb`)."""
    t = np.linspace(0.0, 1.0, max(n, 1))
    return np.stack([np.clip(2 * t - 1, 0, 1), 1 - np.abs(2 * t - 1), np.clip(1 - 2 * t, 0, 1)], -1)


def trajectory_from_rollout(env, out, e=0, state0=None):
    """(trajectory, z_trajectory) of env ``e`` from `drones.rollout` output: positions are recovered from z row 0
    (x_i - xF_i, drone_env.py:357), velocities are not part of the rollout record (left 0)."""
    z = out["z"][:, e].detach().cpu().numpy().astype(np.float64)                  # [T, N, (k+1) c]
    T, N = z.shape[:2]
    c = env.local_state_space // (env.k_closest + 1)
    rows = z.reshape(T, N, env.k_closest + 1, c)
    xF = np.asarray(env.end_points, np.float64).reshape(N, 2)
    traj = []
    if state0 is not None:
        traj.append(np.asarray(state0, np.float64))
    for t in range(T):
        s = np.zeros((N, 5))
        s[:, 0:2] = rows[t, :, 0, 0:2] + xF
        s[:, 4] = env.drone_radius
        traj.append(s)
    return traj, [list(rows[t]) for t in range(T)]


def collision_table(trajectory):
    """[N, T] bool: agent i touches some other agent at step t (gap <= 0, as drone_env.py:462-473 counts)."""
    st = np.stack([np.asarray(s, np.float64) for s in trajectory])                # [T, N, 5]
    d = np.linalg.norm(st[:, :, None, 0:2] - st[:, None, :, 0:2], axis=-1) - st[:, :, None, 4] - st[:, None, :, 4]
    T, N = st.shape[:2]
    d[:, np.arange(N), np.arange(N)] = np.inf
    return (d <= 0).any(-1).T


def _pyplot():
    import matplotlib
    if not os.environ.get("DISPLAY") and matplotlib.get_backend().lower() not in ("agg", "pdf", "svg"):
        matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    return plt


def show_state(env, state=None, show=False):
    """Agents (circles of radius l_i), goals (stars) and obstacles on the grid (drone_env.py:404-434)."""
    plt = _pyplot()
    state = np.asarray(env.state if state is None else state)
    fig, ax = plt.subplots()
    ax.set_xlim(0, env.grid[0]); ax.set_ylim(0, env.grid[1]); ax.grid(True)
    _draw_static(env, ax, state, plt)
    ax.legend()
    if show:
        plt.show()
    return fig


def _draw_static(env, ax, state, plt):
    col = agent_colors(env.n_agents)
    xF = np.asarray(env.end_points).reshape(env.n_agents, 2)
    for ob in np.asarray(getattr(env, "obstacles", np.zeros((0, 3)))).reshape(-1, 3):
        ax.add_patch(plt.Circle((ob[0], ob[1]), ob[2], color="black"))
    circles = []
    for i in range(env.n_agents):
        circles.append(ax.add_patch(plt.Circle((state[i, 0], state[i, 1]), state[i, 4], color=col[i], fill=False, label=f"{i + 1}")))
        ax.plot(xF[i, 0], xF[i, 1], color=col[i], marker="*")
    return circles


def plot_trajectory(env, trajectory, episode=None, show=False):
    """Paths, final positions, goals and collision markers of one episode (drone_env.py:450-514).
    Returns (figure, number of (agent, step) collision marks)."""
    plt = _pyplot()
    st = np.stack([np.asarray(s, np.float64) for s in trajectory])
    hit = collision_table(trajectory)
    col = agent_colors(env.n_agents)
    fig, ax = plt.subplots()
    fig.set_size_inches(4.5, 3.5); fig.tight_layout(); ax.grid(True)
    _draw_static(env, ax, st[-1], plt)
    for i in range(env.n_agents):
        ax.plot(st[:, i, 0], st[:, i, 1], color=col[i])
        ax.plot(st[hit[i], i, 0], st[hit[i], i, 1], color=col[i], marker="v", fillstyle="none", markevery=2, ls="")
    n = int(hit.sum())
    head = f"{env.n_agents} agents, collisions = {n}"
    ax.set_title(head if episode is None else f"Episode {episode + 1} , " + head)
    ax.legend(title="Agents")
    if show:
        plt.show()
    return fig, n


def animate_trajectory(env, trajectory, z_trajectory, deltas, episode=0, name="test", format="gif", folder="videos", fps=30):
    """Agents, their Delta disks and the rays of their localized state over time (drone_env.py:516-607).
    Writes ``folder/name.gif`` (Pillow) or ``.mp4`` (needs an ffmpeg on PATH); returns the file name."""
    plt = _pyplot()
    from matplotlib import animation
    from .drone_env import dt
    N, col = env.n_agents, agent_colors(env.n_agents)
    xF = np.asarray(env.end_points, np.float64).reshape(N, 2)
    deltas = np.broadcast_to(np.asarray(deltas, np.float64), (N,))
    fig, ax = plt.subplots()
    ax.set_xlim(-1, env.grid[0] + 1); ax.set_ylim(-1, env.grid[1] + 1)
    s0 = np.asarray(trajectory[0])
    circles = _draw_static(env, ax, s0, plt)
    disks = [ax.add_patch(plt.Circle((s0[i, 0], s0[i, 1]), s0[i, 4] + deltas[i], color="red", fill=False, ls="--", alpha=0.5))
             for i in range(N)]

    def rays(t):                                    # row 0 hangs off the goal, rows >= 1 off the agent
        s, zs = np.asarray(trajectory[t]), z_trajectory[t]
        out = []
        for i in range(N):
            zi = np.asarray(zs[i]).reshape(-1, np.asarray(zs[i]).shape[-1])
            for k in range(zi.shape[0]):
                a = xF[i] if k == 0 else s[i, 0:2]
                out.append((i, k, a, a + zi[k, 0:2]))
        return out

    lines = {(i, k): ax.plot([a[0], b[0]], [a[1], b[1]], color=col[i], lw=0.5, alpha=0.3 if k == 0 else 0.6)[0]
             for i, k, a, b in rays(0)}
    ax.legend(loc="upper right")

    def update(t):
        s = np.asarray(trajectory[t])
        ax.set_title(f"Episode {episode + 1} .Deltas = {deltas[0]}. Time = {t * dt:.1f}s")
        for i, k, a, b in rays(t):
            lines[(i, k)].set_data([a[0], b[0]], [a[1], b[1]])
        for i in range(N):
            circles[i].center = disks[i].center = (s[i, 0], s[i, 1])
        return circles + disks + list(lines.values())

    anim = animation.FuncAnimation(fig, update, min(len(trajectory), len(z_trajectory)), interval=dt * 1e3)
    os.makedirs(folder, exist_ok=True)
    if format == "gif":
        full, writer = os.path.join(folder, name + ".gif"), animation.PillowWriter(fps=fps)
    elif format == "mp4":
        full, writer = os.path.join(folder, name + ".mp4"), animation.FFMpegWriter(fps=fps)
    else:
        raise ValueError(f"format {format!r} not valid (gif | mp4)")
    anim.save(full, writer=writer)
    plt.close(fig)
    return full
