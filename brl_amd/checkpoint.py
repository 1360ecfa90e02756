"""Model files either side of the path: the trainer's own checkpoints and the reference's pickles.

* ``save_params / load_params`` — ``ppo.py:351-362,550-570`` saves ``params-{i:08}.pkl`` every ``save_model_interval``
  iterations and params + opt_state at the end; those files double as the FSP opponent pool (``ppo.py:383-459``).  Here a
  checkpoint is a torch ``state_dict`` (``params-{i:08}.pt``).
* ``load_haiku_pickle / haiku_to_torch`` — the reference's published models (``bridge_models/*.pkl``,
  ``ppo.py:246-248,343``) are pickled Haiku parameter trees ``{'actor_critic/linear': {'w': [in,out], 'b': [out]}, ...}``
   whose leaves are ``jax.Array`` objects.  Unpickling them normally needs jax; ``load_haiku_pickle`` reads
  them WITHOUT jax by resolving jax's array-reconstruction hook to a numpy constructor.  [RECALL — jax 0.4.23's
  ``ArrayImpl.__reduce__`` returns ``(jax._src.array._reconstruct_array, (fun, args, arr_state, aval_state))`` with
  ``(fun, args, arr_state)`` the wrapped numpy array's own reduce triple; no published pickle is available offline, the
  tests exercise a pickle written with that protocol.  Checked against: that layout only (jax 0.4.23, float32 leaves);
  pickles of newer jax versions or with bf16 / ml_dtypes leaves raise ``UnpicklingError`` rather than load.]  Globals are
  resolved from an allow-list (numpy array reconstruction, plain containers, the jax / haiku stand-ins); anything else —
  ``os.system``, ``builtins.eval`` ... — is refused.
"""
from __future__ import annotations

import io
import os
import pickle
import re

import numpy as np
import torch

from .models import ActorCritic, make_forward_pass


# ---------------------------------------------------------------------------------------------------------------
# torch checkpoints
# ---------------------------------------------------------------------------------------------------------------
def save_params(params: torch.nn.Module, path: str) -> None:
    tmp = path + ".tmp"
    torch.save({k: v.detach().cpu() for k, v in params.state_dict().items()}, tmp)
    os.replace(tmp, path)  # a reader (another rank sampling the pool) never sees a half-written file


def save_opt_state(opt_state: dict, path: str) -> None:
    """Under ``grad_allreduce="sharded"`` Adam's moments are current on each rank's own slices only: every rank calls
    ``opt_state["graphed"].gather_optimizer_state()`` first (a collective — this function, usually called by rank 0 alone, cannot);
    a state with partial moments is refused rather than written."""
    fused = opt_state.get("graphed")
    if getattr(fused, "moments_partial", False):
        raise RuntimeError("save_opt_state: Adam's moments are partial (grad_allreduce='sharded'): call "
                           "opt_state['graphed'].gather_optimizer_state() on every rank first")
    tmp = path + ".tmp"
    torch.save(opt_state["opt"].state_dict(), tmp)
    os.replace(tmp, path)


def load_params(path: str, activation: str, model_type: str, device=None) -> ActorCritic:
    """A ``params-*.pt`` state_dict or a reference ``*.pkl`` Haiku tree -> an ``ActorCritic`` on ``device``."""
    if path.endswith(".pkl"):
        net = haiku_to_torch(load_haiku_pickle(path), activation, model_type)
    else:
        net = make_forward_pass(activation, model_type).init(0)
        net.load_state_dict(torch.load(path, map_location="cpu"))
    return net.to(device) if device is not None else net


def list_checkpoints(directory: str):
    """sorted ``params-*`` files of a run (ppo.py:383-395)."""
    if not os.path.isdir(directory):
        return []
    return sorted(p for p in os.listdir(directory) if p.startswith("params") and not p.endswith(".tmp"))


# ---------------------------------------------------------------------------------------------------------------
# Haiku pickles without jax
# ---------------------------------------------------------------------------------------------------------------
def _reconstruct_array(fun, args, arr_state, aval_state=None):
    """stand-in for ``jax._src.array._reconstruct_array``: rebuild the wrapped numpy array, skip ``device_put``"""
    arr = fun(*args)
    arr.__setstate__(arr_state)
    return arr


class _FlatMapping(dict):
    """stand-in for ``haiku._src.data_structures.FlatMapping`` (older Haiku versions pickle parameter trees as one)"""

    def __init__(self, *a, **kw):
        if len(a) == 1 and not kw and not isinstance(a[0], (dict, list, tuple)):
            a = ()
        super().__init__(*a, **kw)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.update(state.get("_mapping", state))


def _encode_latin1(text, encoding="latin1"):
    """stand-in for ``_codecs.encode`` in a protocol <= 2 numpy reduce (`_codecs.encode(<array bytes as str>, 'latin1')`): that one
    form only — the real function would import whatever codec module an untrusted file names (bz2, zlib, ...)"""
    if encoding not in ("latin1", "latin-1") or not isinstance(text, str):
        raise pickle.UnpicklingError(f"refusing _codecs.encode(..., {encoding!r}): only numpy's latin1 byte strings belong in a parameter tree")
    return text.encode("latin1")


class _HaikuUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) == ("_codecs", "encode"):
            return _encode_latin1
        if module.startswith("jax") and name == "_reconstruct_array":
            return _reconstruct_array
        if module.startswith("haiku") and name in ("FlatMapping", "FlatMap"):
            return _FlatMapping
        if module.startswith("jax") or module.startswith("haiku") or module.startswith("jaxlib"):
            raise pickle.UnpicklingError(f"cannot resolve {module}.{name} without jax / haiku")
        # allow-list: a parameter tree needs numpy's array reconstruction and plain containers, nothing else — a pickle
        # is a program, and model files (initial_model_path, eval_opp_model_path, pool files) may come from third parties
        if (module, name) in _SAFE_GLOBALS:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to resolve {module}.{name}: not part of a Haiku parameter tree")


_SAFE_GLOBALS = frozenset(
    [(m, n) for m in ("numpy.core.multiarray", "numpy._core.multiarray") for n in ("_reconstruct", "scalar")]
    + [("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")]   # ndarray.__reduce_ex__ under protocol 5
    # (array bytes under protocol <= 2 come through `_codecs.encode`: resolved to `_encode_latin1` above, never the real function)
    + [("numpy", "ndarray"), ("numpy", "dtype"), ("collections", "OrderedDict"), ("builtins", "dict"), ("builtins", "list"),
       ("builtins", "tuple"), ("builtins", "set"), ("builtins", "frozenset")])


def load_haiku_pickle(path_or_bytes) -> dict:
    """-> ``{module_name: {'w': np.ndarray [in,out], 'b': np.ndarray [out]}}`` (numpy leaves)."""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        tree = _HaikuUnpickler(io.BytesIO(path_or_bytes)).load()
    else:
        with open(path_or_bytes, "rb") as f:
            tree = _HaikuUnpickler(f).load()
    return {str(k): {str(kk): np.asarray(vv) for kk, vv in dict(v).items()} for k, v in dict(tree).items()}


def _creation_order(tree: dict):
    """Haiku names the Linear modules of one ``ActorCritic.__call__`` ``linear, linear_1, linear_2, ...`` in creation
    order (src/models.py:18-69)."""
    def idx(name):
        m = re.search(r"linear(?:_(\d+))?$", name)
        if not m:
            raise ValueError(f"unexpected module name {name!r} in the parameter tree")
        return int(m.group(1) or 0)
    return [tree[k] for k in sorted(tree, key=idx)]


def _torch_linears(net: ActorCritic):
    body = list(net.body) if net.model.startswith("DeepMind") else list(net.l)
    return body + [net.actor, net.critic]


def haiku_to_torch(tree: dict, activation: str, model_type: str) -> ActorCritic:
    """Haiku ``w`` is [in, out] = the transpose of ``torch.nn.Linear.weight``."""
    net = ActorCritic(38, activation, model_type)
    mods, lins = _creation_order(tree), _torch_linears(net)
    if len(mods) != len(lins):
        raise ValueError(f"{model_type}: expected {len(lins)} Linear modules, the tree has {len(mods)}")
    with torch.no_grad():
        for lin, m in zip(lins, mods):
            w, b = np.asarray(m["w"], np.float32), np.asarray(m["b"], np.float32)
            if tuple(w.shape) != (lin.in_features, lin.out_features):
                raise ValueError(f"shape mismatch: {w.shape} vs Linear({lin.in_features}, {lin.out_features})")
            lin.weight.copy_(torch.from_numpy(w.T.copy()))
            lin.bias.copy_(torch.from_numpy(b))
    return net


def torch_to_haiku(net: ActorCritic) -> dict:
    """The inverse (numpy leaves): what ``pickle.dump(params)`` of the reference would hold for these weights."""
    tree = {}
    for i, lin in enumerate(_torch_linears(net)):
        name = "actor_critic/linear" + (f"_{i}" if i else "")
        tree[name] = {"w": lin.weight.detach().cpu().numpy().T.copy(), "b": lin.bias.detach().cpu().numpy().copy()}
    return tree
