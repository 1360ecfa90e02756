"""Host logic either side of the hot path, CPU only: opponent pool decisions (ppo.py:376-460), LUT
rotation bookkeeping (ppo.py:525-549), checkpoints, the Haiku-pickle reader, the opponent-index broadcast (gloo)."""
import os
import pickle
import socket
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from brl_amd import checkpoint as ckpt
from brl_amd.models import make_forward_pass
from brl_amd.train import DEFAULTS, LutRotation, choose_opponent, linear_schedule, parse_cli, pfsp_probabilities

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_defaults_match_ppo_config():
    # ppo.py:122-180
    want = dict(seed=0, lr=0.000001, num_envs=8192, num_steps=32, total_timesteps=2_621_440_000, update_epochs=10,
                minibatch_size=1024, hash_size=100_000, num_eval_envs=10000, num_eval_step=10, save_model_interval=1,
                ratio_model_zoo=0, num_model_zoo=100_000, threshold_model_zoo=-24, prioritized_fictitious=False,
                prior_t=0.1, num_prioritized_envs=100, gamma=1, gae_lambda=0.95, clip_eps=0.2, ent_coef=0.001,
                vf_coef=0.5, max_grad_norm=0.5, reward_scale=7600, actor_illegal_action_mask=True,
                actor_illegal_action_penalty=False, illegal_action_l2norm_coef=0, game_mode="competitive", self_play=True)
    for k, v in want.items():
        assert DEFAULTS[k] == v, k
    cfg = parse_cli(["num_envs=64", "lr=1e-4", "self_play=false", "game_mode=free-run", "dds_results_dir=/x"])
    assert cfg["num_envs"] == 64 and cfg["lr"] == 1e-4 and cfg["self_play"] is False and cfg["game_mode"] == "free-run"
    assert cfg["dds_results_dir"] == "/x"
    with pytest.raises(SystemExit):
        parse_cli(["no_such_key=1"])


def test_pfsp_probabilities_softmax_of_negative_imp():
    imps = np.array([2.0, -1.0, 0.5])
    p = pfsp_probabilities(imps, 0.1)               # ppo.py:424-435
    want = np.exp(-imps / 0.1) / np.exp(-imps / 0.1).sum()
    assert np.allclose(p, want) and abs(p.sum() - 1) < 1e-12 and p.argmax() == 1   # the opponent we lose to most
    assert np.allclose(pfsp_probabilities([3.0, 3.0], 0.1), [0.5, 0.5])


def test_choose_opponent_decision_tree():
    cfg = dict(DEFAULTS, threshold_model_zoo=0.5, ratio_model_zoo=1.0)
    rng = np.random.RandomState(0)
    assert choose_opponent(cfg, 0.4, ["a", "b"], rng) is None            # below the gate: keep the current opponent
    assert choose_opponent(cfg, 0.6, [], rng) == -1                      # empty pool: latest
    picks = {choose_opponent(cfg, 0.6, ["a", "b", "c"], rng) for _ in range(50)}
    assert picks == {0, 1, 2}                                             # FSP: uniform over the pool
    assert choose_opponent(dict(cfg, ratio_model_zoo=0.0), 0.6, ["a"], rng) == -1
    pf = dict(cfg, prioritized_fictitious=True, prior_t=0.01)
    calls = []
    def league():
        calls.append(1)
        return np.array([5.0, -3.0, 4.0])
    assert all(choose_opponent(pf, 0.6, ["a", "b", "c"], rng, league) == 1 for _ in range(10)) and len(calls) == 10


def test_lut_rotation_follows_hash_size():
    rng = np.random.RandomState(1)
    rot = LutRotation(3, 100, rng)
    assert rot.current == 0 and not rot.advance(99)
    assert rot.advance(100) and rot.current == 1 and rot.board_count == 100
    assert not rot.advance(150) and rot.advance(260) and rot.current == 2
    assert rot.advance(400) and rot.hash_index == 0 and sorted(rot.order) == [0, 1, 2]   # all used: reshuffled (ppo.py:529-532)
    cfg = dict(DEFAULTS, num_minibatches=4, update_epochs=2, num_updates=10, lr=1.0)
    assert linear_schedule(cfg, 0) == 1.0 and abs(linear_schedule(cfg, 8 * 5) - 0.5) < 1e-12


def test_checkpoint_roundtrip(tmp_path):
    fp = make_forward_pass("relu", "DeepMind")
    net = fp.init(3)
    ckpt.save_params(net, str(tmp_path / "params-00000001.pt"))
    ckpt.save_params(net, str(tmp_path / "params-00000002.pt"))
    assert ckpt.list_checkpoints(str(tmp_path)) == ["params-00000001.pt", "params-00000002.pt"]
    back = ckpt.load_params(str(tmp_path / "params-00000002.pt"), "relu", "DeepMind")
    for a, b in zip(net.parameters(), back.parameters()):
        assert torch.equal(a, b)


class _FakeJaxArray:
    """Pickles the way jax 0.4.23's ArrayImpl does: (jax._src.array._reconstruct_array, (fun, args, arr_state, aval_state))."""

    def __init__(self, a):
        self.a = np.asarray(a)

    def __reduce__(self):
        fun, args, arr_state = self.a.__reduce__()
        return (sys.modules["jax._src.array"]._reconstruct_array, (fun, args, arr_state, {"weak_type": False, "named_shape": {}}))


@pytest.mark.parametrize("model,activation", [("DeepMind", "relu"), ("FAIR", "tanh")])
def test_haiku_pickle_is_read_without_jax(tmp_path, monkeypatch, model, activation):
    """A parameter tree pickled with jax-array leaves loads with no jax installed and gives a torch
    module whose forward equals x @ w + b of the tree."""
    fp = make_forward_pass(activation, model)
    net = fp.init(7)
    tree = ckpt.torch_to_haiku(net)
    assert list(tree)[0] == "actor_critic/linear" and tree["actor_critic/linear"]["w"].shape[0] == 480   # [in, out]
    # write the pickle with a stand-in `jax._src.array` module present (the writer side has jax) ...
    fake = types.ModuleType("jax._src.array")
    fake._reconstruct_array = lambda fun, args, st, aval: fun(*args)
    fake._reconstruct_array.__module__ = "jax._src.array"
    fake._reconstruct_array.__qualname__ = fake._reconstruct_array.__name__ = "_reconstruct_array"
    for name in ("jax", "jax._src"):
        monkeypatch.setitem(sys.modules, name, types.ModuleType(name))
    monkeypatch.setitem(sys.modules, "jax._src.array", fake)
    path = tmp_path / "model.pkl"
    with open(path, "wb") as f:
        pickle.dump({k: {kk: _FakeJaxArray(vv) for kk, vv in v.items()} for k, v in tree.items()}, f)
    # ... and read it back with it gone (the reader side has none)
    for name in ("jax", "jax._src", "jax._src.array"):
        monkeypatch.delitem(sys.modules, name)
    with pytest.raises(Exception):
        pickle.load(open(path, "rb"))
    back = ckpt.load_params(str(path), activation, model)
    x = torch.rand(16, 480)
    with torch.no_grad():
        l0, v0 = net(x)
        l1, v1 = back(x)
    assert torch.equal(l0, l1) and torch.equal(v0, v1)
    if model == "DeepMind":   # the layer chain by hand from the tree: hk.Linear is x @ w + b
        h = x.numpy()
        for i in range(4):
            m = tree["actor_critic/linear" + (f"_{i}" if i else "")]
            h = np.maximum(h @ m["w"] + m["b"], 0)
        assert np.allclose(h @ tree["actor_critic/linear_4"]["w"] + tree["actor_critic/linear_4"]["b"], l0.numpy(), atol=1e-4)
    with pytest.raises(ValueError):
        ckpt.haiku_to_torch(tree, activation, "FAIR" if model == "DeepMind" else "DeepMind")


def test_haiku_pickle_reader_refuses_foreign_globals(tmp_path):
    """A model file is a program: the reader resolves numpy's array reconstruction and plain containers only."""
    class _Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    for payload in (_Evil(), {"actor_critic/linear": {"w": _Evil()}}):
        with pytest.raises(pickle.UnpicklingError, match="refusing"):
            ckpt.load_haiku_pickle(pickle.dumps(payload))
    # `_codecs.encode` (numpy's protocol <= 2 byte strings) resolves to a latin1-only stand-in: a file cannot name another codec
    # (the real function would import encodings.<name>: bz2, zlib, ...)
    class _Codec:
        def __reduce__(self):
            import _codecs
            return (_codecs.encode, ("abc", "bz2"))
    with pytest.raises(pickle.UnpicklingError, match="latin1"):
        ckpt.load_haiku_pickle(pickle.dumps({"actor_critic/linear": {"w": _Codec()}}, protocol=2))
    # a plain numpy tree (what a jax-free writer would dump) still loads
    tree = {"actor_critic/linear": {"w": np.ones((480, 4), np.float32), "b": np.zeros(4, np.float32)}}
    back = ckpt.load_haiku_pickle(pickle.dumps(tree))
    assert np.array_equal(back["actor_critic/linear"]["w"], tree["actor_critic/linear"]["w"])


def test_haiku_pickle_reader_accepts_every_pickle_protocol():
    """ndarray.__reduce_ex__ emits numpy(._)core.numeric._frombuffer under protocol 5 (HIGHEST_PROTOCOL): plain numpy trees
    written that way — opponent / initial model files of a jax-free writer — must keep loading."""
    rs = np.random.RandomState(0)
    tree = {"actor_critic/linear": {"w": rs.randn(480, 8).astype(np.float32), "b": rs.randn(8).astype(np.float32)},
            "actor_critic/linear_1": {"w": np.asfortranarray(rs.randn(8, 8).astype(np.float32)), "b": np.zeros(8, np.float32)}}
    for proto in range(2, pickle.HIGHEST_PROTOCOL + 1):
        back = ckpt.load_haiku_pickle(pickle.dumps(tree, protocol=proto))
        for k, v in tree.items():
            for kk, a in v.items():
                assert np.array_equal(back[k][kk], a), (proto, k, kk)


def test_weight_delta_log_field():
    """train.py's opp_weight_delta: one number; NaN (never an exception) for trees of different shape or the same object"""
    from brl_amd.models import make_forward_pass
    from brl_amd.train import _weight_delta
    fp = make_forward_pass("relu", "DeepMind")
    a, b = fp.init(0, device="cpu"), fp.init(0, device="cpu")
    assert _weight_delta(a, b) == 0.0 and np.isnan(_weight_delta(a, a))
    with torch.no_grad():
        list(b.parameters())[3].view(-1)[5] += 0.25
    assert abs(_weight_delta(a, b) - 0.25) < 1e-6
    fair = make_forward_pass("relu", "FAIR").init(0, device="cpu")
    assert np.isnan(_weight_delta(a, fair))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pool_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from brl_amd.dist import broadcast_int
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = dict(DEFAULTS, threshold_model_zoo=-24.0, ratio_model_zoo=1.0)
    rng = np.random.RandomState(100 + rank)          # different host RNG streams on purpose
    got = []
    for it in range(20):
        choice = choose_opponent(cfg, 1.0, [f"params-{k:08}.pt" for k in range(7)], rng) if rank == 0 else None
        got.append(broadcast_int(-2 if choice is None else choice))
    torch.save(got, os.path.join(out_dir, f"choices{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_opponent_index_is_broadcast_from_rank0(tmp_path):
    mp.start_processes(_pool_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True, start_method="spawn")
    a, b = torch.load(tmp_path / "choices0.pt"), torch.load(tmp_path / "choices1.pt")
    assert a == b and len(set(a)) > 1 and all(0 <= x < 7 for x in a)
