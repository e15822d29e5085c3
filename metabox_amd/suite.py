"""Device-resident problem suites and lock-step instance batches (thin host wrappers over the C-ABI).

``Suite``  ~ the list of problem objects returned by the reference's ``construct_problem_set``
            (src/utils.py:4-27), uploaded once to HBM.
``Batch``  ~ B independent ``PBO_Env(problem, optimizer)`` pairs (src/environment/basic_environment.py:6-22)
            stepped together by one kernel launch per ``step``.

PyTorch is used only for device memory and streams; the arithmetic lives in libmbx.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _abi


def _require_gpu():
    if not torch.cuda.is_available():
        raise _abi.MbxError('metabox_amd needs a ROCm GPU (MI355X); there is no CPU fallback for the hot path.')


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Suite:
    """A set of problem instances of one dimension, resident on the current GPU."""

    def __init__(self, problems):
        _require_gpu()
        self.lib = _abi.load_lib()
        self.problems = list(problems)
        self.dim = int(self.problems[0].dim)
        descs = [p.desc() for p in self.problems]
        arr, self._keep = _abi.pack_descs(descs)
        opts = [getattr(p, 'opt', None) for p in self.problems]
        opt_arr = None
        if all(o is not None for o in opts):
            opt_arr = np.ascontiguousarray(np.stack([np.asarray(o, dtype=np.float64) for o in opts]))
        h = C.c_void_p()
        _abi.check(self.lib.mbx_suite_create(arr, len(descs), opt_arr.ctypes.data_as(_abi.c_double_p) if opt_arr is not None
                                             else _abi.c_double_p(), C.byref(h)))
        self._h = h
        out = np.empty(len(descs))
        _abi.check(self.lib.mbx_suite_optimum(self._h, out.ctypes.data_as(_abi.c_double_p)))
        self._optimum = out
        for i, p in enumerate(self.problems):
            p._suite = self
            p._suite_index = i
            p._optimum = None if np.isnan(out[i]) else float(out[i])

    def __len__(self):
        return len(self.problems)

    def optimum(self, i):
        v = self._optimum[i]
        return None if np.isnan(v) else float(v)

    def eval_device(self, i, x_dev, noisy=False, seed=0, noise_draws=None):
        """x_dev: float64 CUDA tensor [n, dim] -> float64 CUDA tensor [n] (asynchronous on the current stream)."""
        assert x_dev.is_cuda and x_dev.dtype == torch.float64 and x_dev.is_contiguous() and x_dev.shape[1] == self.dim
        n = x_dev.shape[0]
        f = torch.empty(n, dtype=torch.float64, device=x_dev.device)
        _abi.check(self.lib.mbx_eval(self._h, int(i), _ptr(x_dev), n, _ptr(f), int(bool(noisy)), int(seed),
                                     _ptr(noise_draws), _stream()))
        return f

    _eval_calls = 0

    def eval(self, i, x, noisy=False, seed=None, noise_draws=None):
        """Host convenience used by ``problem.eval``: numpy in, numpy out."""
        xd = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float64)).cuda()
        if seed is None:                       # fresh Philox stream per call, like consuming a global RNG
            Suite._eval_calls += 1
            seed = (int(np.random.randint(0, 2 ** 31 - 1)) << 20) + Suite._eval_calls
        nd = None
        if noise_draws is not None:
            nd = torch.from_numpy(np.ascontiguousarray(noise_draws, dtype=np.float64)).cuda()
        return self.eval_device(i, xd, noisy, seed, nd).cpu().numpy()

    def close(self):
        if getattr(self, '_h', None):
            self.lib.mbx_suite_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Batch:
    """B lock-step optimizer instances on one GPU (``mbx_batch``)."""

    def __init__(self, suite, algo, problem_idx, seeds, np_, max_fes, log_interval, n_logpoint, early_stop=True, n_group=5, flags=0):
        """flags: ``mbx_algo_cfg.flags`` (``_abi.F_FDR_FAST | F_GENERIC_GEOMETRY | F_ROLLOUT_PER_GENERATION``), per batch; ``self.flags`` is what the
        library made of them (environment overrides included)."""
        _require_gpu()
        self.lib = suite.lib
        self.suite = suite
        self.cfg = _abi.AlgoCfg(int(algo), int(np_), suite.dim, int(max_fes), int(log_interval), int(n_logpoint),
                                int(bool(early_stop)), int(n_group), int(flags))
        pidx = np.ascontiguousarray(problem_idx, dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert pidx.shape == sd.shape and pidx.ndim == 1
        self.B = int(pidx.shape[0])
        h = C.c_void_p()
        _abi.check(self.lib.mbx_batch_create(suite._h, C.byref(self.cfg), pidx.ctypes.data_as(C.POINTER(C.c_int32)),
                                             sd.ctypes.data_as(C.POINTER(C.c_uint64)), self.B, C.byref(h)))
        self._h = h
        self.flags = int(self.lib.mbx_batch_flags(h))
        self.cfg.flags = self.flags            # the geometry queries below must see the flags the batch really has
        self.state_dim = self.lib.mbx_state_dim(C.byref(self.cfg))
        self.action_dim = self.lib.mbx_action_dim(C.byref(self.cfg))
        self.tape_stride = int(self.lib.mbx_tape_stride(C.byref(self.cfg)))
        dev = torch.device('cuda', torch.cuda.current_device())
        self.device = dev
        self.state = torch.zeros(self.B, self.state_dim, dtype=torch.float64, device=dev)
        self.reward = torch.zeros(self.B, dtype=torch.float64, device=dev)
        self.done = torch.zeros(self.B, dtype=torch.uint8, device=dev)
        self._tape = None

    def set_tape(self, tape):
        """tape: float64 CUDA tensor [B, tape_stride] or None (back to Philox)."""
        if tape is not None:
            assert tape.is_cuda and tape.dtype == torch.float64 and tape.is_contiguous()
            assert tuple(tape.shape) == (self.B, self.tape_stride)
        self._tape = tape
        _abi.check(self.lib.mbx_set_tape(self._h, _ptr(tape)))

    def reset(self):
        _abi.check(self.lib.mbx_reset(self._h, _ptr(self.state), _stream()))
        return self.state

    def step(self, actions):
        """actions: CUDA tensor [B, action_dim] (float32).  Returns (state, reward, done) device tensors that are
        overwritten by the next call."""
        if self.action_dim > 0:
            want = torch.int32 if self.cfg.algo in (_abi.ALGO_DEDDQN, _abi.ALGO_QLPSO) else torch.float32
            assert actions.is_cuda and actions.is_contiguous() and actions.dtype == want
            assert actions.numel() == self.B * self.action_dim
        else:
            actions = None                     # Random_search takes no action
        _abi.check(self.lib.mbx_step(self._h, _ptr(actions), _ptr(self.state), _ptr(self.reward), _ptr(self.done), _stream()))
        return self.state, self.reward, self.done

    def _net(self, weights, h1, h2, min_sigma, max_sigma):
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        variant = _abi.POLICY_RLPSO if self.cfg.algo == _abi.ALGO_RLPSO else _abi.POLICY_RLEPSO
        return _abi.GaussMlp(weights.data_ptr(), self.state_dim, int(h1), int(h2), self.action_dim, float(min_sigma), float(max_sigma), variant)

    def rlpso_rollout(self, weights, h1, h2, min_sigma, max_sigma, n_steps, want_actions=False):
        """`n_steps` RL-PSO env steps of every instance in ONE launch, actor evaluated in the kernel (``mbx_rlpso_rollout``).
        Returns (state, reward summed over the executed steps, done[, last actions])."""
        acts = None
        if want_actions:
            if getattr(self, '_actions', None) is None:
                self._actions = torch.empty(self.B, self.action_dim, dtype=torch.float32, device=self.device)
            acts = self._actions
        net = self._net(weights, h1, h2, min_sigma, max_sigma)
        _abi.check(self.lib.mbx_rlpso_rollout(self._h, C.byref(net), int(n_steps), _ptr(acts), _ptr(self.state), _ptr(self.reward),
                                              _ptr(self.done), _stream()))
        return (self.state, self.reward, self.done, acts) if want_actions else (self.state, self.reward, self.done)

    def gleet_policy(self, weights, min_sigma, max_sigma, want_mu_sigma=False):
        """GLEET's attention actor over the batch's current state in one launch (``mbx_gleet_policy``).  weights: the actor's
        state_dict flattened (``Actor.packed_weights``).  Returns [B, np] float32 actions (overwritten by the next call), plus
        [B, 2, np] (mu, sigma) if asked."""
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        if getattr(self, '_actions', None) is None:
            self._actions = torch.empty(self.B, self.action_dim, dtype=torch.float32, device=self.device)
        ms = torch.empty(self.B, 2, self.action_dim, dtype=torch.float32, device=self.device) if want_mu_sigma else None
        net = _abi.GleetActor(weights.data_ptr(), int(weights.numel()), float(min_sigma), float(max_sigma))
        _abi.check(self.lib.mbx_gleet_policy(self._h, C.byref(net), _ptr(self.state), _ptr(self._actions), _ptr(ms), _stream()))
        return (self._actions, ms) if want_mu_sigma else self._actions

    def qlpso_rollout(self, q_table, n_steps, want_actions=False):
        """`n_steps` QLPSO env steps of every instance in ONE launch, tabular policy evaluated in the kernel (``mbx_qlpso_rollout``).
        q_table: [4, 4] float64 CUDA tensor.  Returns (state, reward summed over the executed steps, done[, last actions int32])."""
        assert q_table.is_cuda and q_table.dtype == torch.float64 and q_table.is_contiguous() and tuple(q_table.shape) == (4, 4)
        acts = None
        if want_actions:
            if getattr(self, '_iactions', None) is None:
                self._iactions = torch.empty(self.B, dtype=torch.int32, device=self.device)
            acts = self._iactions
        _abi.check(self.lib.mbx_qlpso_rollout(self._h, _ptr(q_table), int(n_steps), _ptr(acts), _ptr(self.state), _ptr(self.reward),
                                              _ptr(self.done), _stream()))
        return (self.state, self.reward, self.done, acts) if want_actions else (self.state, self.reward, self.done)

    def ddqn_qnet(self, weights, width=100, depth=4, n_act=4, want_q=False):
        """DE-DDQN's greedy action over the batch's own state tensor in one launch (``mbx_ddqn_qnet``: the Q-network on the float32
        matrix cores).  weights: packed float32 CUDA tensor (``DE_DDQN_Agent.packed_weights``: per layer Wt [in][out], b [out]).
        Returns the int32 [B] action tensor (overwritten by the next call)[, Q values [B, n_act]]."""
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        need = self.state_dim * width + width + (depth - 1) * (width * width + width) + width * n_act + n_act      # qnet_floats (mbx_qnet.hpp)
        if weights.numel() != need:
            raise ValueError(f'packed Q-network has {weights.numel()} floats, {self.state_dim} -> {width} x {depth} -> {n_act} needs {need}')
        if getattr(self, '_iactions', None) is None:
            self._iactions = torch.empty(self.B, dtype=torch.int32, device=self.device)
        q = torch.empty(self.B, n_act, dtype=torch.float32, device=self.device) if want_q else None
        net = _abi.QNet(weights.data_ptr(), self.state_dim, int(width), int(depth), int(n_act))
        _abi.check(self.lib.mbx_ddqn_qnet(self._h, C.byref(net), _ptr(self.state), _ptr(self._iactions), _ptr(q), _stream()))
        return (self._iactions, q) if want_q else self._iactions

    def lde_policy(self, weights, hidden, h, c, want_mu_sigma=False, sample=True):
        """LDE's PolicyNet in one launch (``mbx_lde_policy``): reads the batch's own state tensor, updates h / c [B, hidden] (float32,
        contiguous) in place, returns the actions [B, 2 NP] (float32; None when ``sample`` is False)[, mu_sigma [B, 2, 2 NP]]."""
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        assert h.is_cuda and c.is_cuda and h.dtype == c.dtype == torch.float32 and h.is_contiguous() and c.is_contiguous()
        assert h.numel() == c.numel() == self.B * hidden
        need = (self.state_dim + hidden) * 4 * hidden + 4 * hidden + 2 * hidden * self.action_dim + 2 * self.action_dim   # lstm_policy_floats (mbx_lstm_policy.hpp)
        if weights.numel() != need:
            raise ValueError(f'packed LSTM policy has {weights.numel()} floats, in {self.state_dim} / hidden {hidden} / out {self.action_dim} needs {need}')
        if getattr(self, '_actions', None) is None:
            self._actions = torch.empty(self.B, self.action_dim, dtype=torch.float32, device=self.device)
        ms = torch.empty(self.B, 2, self.action_dim, dtype=torch.float32, device=self.device) if want_mu_sigma else None
        net = _abi.LstmPolicy(weights.data_ptr(), self.state_dim, int(hidden), self.action_dim)
        _abi.check(self.lib.mbx_lde_policy(self._h, C.byref(net), _ptr(self.state), _ptr(h), _ptr(c), _ptr(self._actions) if sample else None,
                                           _ptr(ms), _stream()))
        acts = self._actions if sample else None
        return (acts, ms) if want_mu_sigma else acts

    def lde_rollout_is_resident(self):
        """True when lde_rollout runs the resident kernel k_lde_run (``mbx_lde_rollout_resident``), False for mbx_lde_policy + mbx_step per generation."""
        return int(self.lib.mbx_lde_rollout_resident(self._h)) == 1

    def lde_rollout(self, weights, hidden, h, c, n_gens, trajectory=False):
        """Up to `n_gens` generations of PolicyNet.sampler + env.step per instance in ONE launch (``mbx_lde_rollout``): population, fitness order,
        features and (h, c) stay on chip in between; bit-identical to `n_gens` x (lde_policy + step).  Reads and updates the batch's own state
        tensor, h / c [B, hidden] in place.  Returns (state, reward summed over the executed generations, done) and, with ``trajectory=True``, a
        dict of per-generation records: actions [n_gens, B, 2 NP] float32, state [n_gens, B, NP + 10] / reward [n_gens, B] float64, done uint8."""
        assert weights.is_cuda and weights.dtype == torch.float32 and weights.is_contiguous()
        assert h.is_cuda and c.is_cuda and h.dtype == c.dtype == torch.float32 and h.is_contiguous() and c.is_contiguous()
        assert h.numel() == c.numel() == self.B * hidden
        need = (self.state_dim + hidden) * 4 * hidden + 4 * hidden + 2 * hidden * self.action_dim + 2 * self.action_dim
        if weights.numel() != need:
            raise ValueError(f'packed LSTM policy has {weights.numel()} floats, in {self.state_dim} / hidden {hidden} / out {self.action_dim} needs {need}')
        n_gens = int(n_gens)
        traj = None
        if trajectory:
            traj = {'actions': torch.zeros(n_gens, self.B, self.action_dim, dtype=torch.float32, device=self.device),
                    'state': torch.zeros(n_gens, self.B, self.state_dim, dtype=torch.float64, device=self.device),
                    'reward': torch.empty(n_gens, self.B, dtype=torch.float64, device=self.device),
                    'done': torch.empty(n_gens, self.B, dtype=torch.uint8, device=self.device)}
        t = traj or {}
        net = _abi.LstmPolicy(weights.data_ptr(), self.state_dim, int(hidden), self.action_dim)
        _abi.check(self.lib.mbx_lde_rollout(self._h, C.byref(net), _ptr(self.state), _ptr(h), _ptr(c), n_gens, _ptr(t.get('actions')),
                                            _ptr(t.get('state')), _ptr(t.get('reward')), _ptr(t.get('done')), _ptr(self.state), _ptr(self.reward),
                                            _ptr(self.done), _stream()))
        return (self.state, self.reward, self.done, traj) if trajectory else (self.state, self.reward, self.done)

    def gauss_policy(self, weights, h1, h2, min_sigma, max_sigma, want_mu_sigma=False):
        """RLEPSO / RL-PSO actor over the batch's current state in one launch (``mbx_gauss_policy``).  weights: packed float32 CUDA
        tensor (``Actor.packed_weights``).  Returns the [B, action_dim] float32 action tensor (overwritten by the next
        call), plus [B, 2, action_dim] (mu, sigma) if asked."""
        if getattr(self, '_actions', None) is None:
            self._actions = torch.empty(self.B, self.action_dim, dtype=torch.float32, device=self.device)
        ms = torch.empty(self.B, 2, self.action_dim, dtype=torch.float32, device=self.device) if want_mu_sigma else None
        net = self._net(weights, h1, h2, min_sigma, max_sigma)
        _abi.check(self.lib.mbx_gauss_policy(self._h, C.byref(net), _ptr(self.state), _ptr(self._actions), _ptr(ms), _stream()))
        return (self._actions, ms) if want_mu_sigma else self._actions

    def policy_table(self, weights, h1, h2, min_sigma, max_sigma):
        """(mu, sigma) of the actor at every reachable state fes/maxFEs -> [rows, 2, action_dim] float32 (``mbx_rlepso_policy_table``)."""
        rows = int(self.lib.mbx_rlepso_policy_table_rows(self._h))
        table = torch.empty(rows, 2, self.action_dim, dtype=torch.float32, device=self.device)
        net = self._net(weights, h1, h2, min_sigma, max_sigma)
        _abi.check(self.lib.mbx_rlepso_policy_table(self._h, C.byref(net), _ptr(table), _stream()))
        return table

    def _check_table(self, table):
        """The kernels index the actor table by fes, clamped to mbx_rlepso_policy_table_rows(batch): a table with fewer rows or another
        action width would be read out of bounds on the device, so its shape is checked here."""
        assert table.is_cuda and table.dtype == torch.float32 and table.is_contiguous()
        rows = int(self.lib.mbx_rlepso_policy_table_rows(self._h))
        if tuple(table.shape) != (rows, 2, self.action_dim):
            raise ValueError(f'actor table has shape {tuple(table.shape)}, this batch needs ({rows}, 2, {self.action_dim}) '
                             f'(mbx_rlepso_policy_table_rows x (mu, sigma) x action_dim)')

    def rollout_is_resident(self):
        """True when rlepso_rollout runs the resident kernel (one launch per call), False for the one-launch-per-generation route
        (``mbx_rlepso_rollout_resident``)."""
        return int(self.lib.mbx_rlepso_rollout_resident(self._h)) == 1

    def act_step(self, table, want_actions=False):
        """agent.act + env.step in one launch (``mbx_rlepso_act_step``): the action of every instance is drawn inside the generation
        kernel from row fes of `table`.  Returns (state, reward, done[, actions])."""
        self._check_table(table)
        acts = None
        if want_actions:
            if getattr(self, '_actions', None) is None:
                self._actions = torch.empty(self.B, self.action_dim, dtype=torch.float32, device=self.device)
            acts = self._actions
        _abi.check(self.lib.mbx_rlepso_act_step(self._h, _ptr(table), _ptr(acts), _ptr(self.state), _ptr(self.reward), _ptr(self.done),
                                                _stream()))
        return (self.state, self.reward, self.done, acts) if want_actions else (self.state, self.reward, self.done)

    def rlepso_rollout(self, table, n_gens, trajectory=False):
        """Up to `n_gens` generations of agent.act + env.step per instance in ONE launch with the state on chip in between
        (``mbx_rlepso_rollout``); bit-identical to `n_gens` calls of act_step.  Returns (state, reward summed over the executed
        generations, done) and, with ``trajectory=True``, a dict of per-generation records: actions [n_gens, B, action_dim] float32,
        state / reward [n_gens, B] float64, done [n_gens, B] uint8 (rows after an instance's termination: reward 0, done 1, actions
        not written)."""
        self._check_table(table)
        n_gens = int(n_gens)
        traj = None
        if trajectory:
            traj = {'actions': torch.zeros(n_gens, self.B, self.action_dim, dtype=torch.float32, device=self.device),
                    'state': torch.empty(n_gens, self.B, dtype=torch.float64, device=self.device),
                    'reward': torch.empty(n_gens, self.B, dtype=torch.float64, device=self.device),
                    'done': torch.empty(n_gens, self.B, dtype=torch.uint8, device=self.device)}
        t = traj or {}
        _abi.check(self.lib.mbx_rlepso_rollout(self._h, _ptr(table), n_gens, _ptr(t.get('actions')), _ptr(t.get('state')),
                                               _ptr(t.get('reward')), _ptr(t.get('done')), _ptr(self.state), _ptr(self.reward),
                                               _ptr(self.done), _stream()))
        return (self.state, self.reward, self.done, traj) if trajectory else (self.state, self.reward, self.done)

    def results(self):
        """-> dict of device tensors: cost [B, n_logpoint+1], fes [B], return [B], steps [B], cost_len [B]."""
        n = self.cfg.n_logpoint + 1
        cost = torch.empty(self.B, n, dtype=torch.float64, device=self.device)
        fes = torch.empty(self.B, dtype=torch.float64, device=self.device)
        ret = torch.empty(self.B, dtype=torch.float64, device=self.device)
        steps = torch.empty(self.B, dtype=torch.int32, device=self.device)
        clen = torch.empty(self.B, dtype=torch.int32, device=self.device)
        _abi.check(self.lib.mbx_results(self._h, _ptr(cost), _ptr(fes), _ptr(ret), _ptr(steps), _ptr(clen), _stream()))
        return {'cost': cost, 'fes': fes, 'return': ret, 'steps': steps, 'cost_len': clen}

    def rebind(self, problem_idx, seeds):
        """New (problem, seed) pairs for the same batch (``mbx_batch_rebind``): no allocation, the next reset() starts their episode 0."""
        pidx = np.ascontiguousarray(problem_idx, dtype=np.int32)
        sd = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert pidx.shape == sd.shape == (self.B,)
        _abi.check(self.lib.mbx_batch_rebind(self._h, pidx.ctypes.data_as(C.c_void_p), sd.ctypes.data_as(C.c_void_p)))

    def read_public(self, instance=0):
        """Scalar block + cost list of one instance (``mbx_read_public``): numpy [NSCALAR + n_logpoint + 1], one small D2H copy."""
        if getattr(self, '_pub', None) is None:
            self._pub = np.empty(16 + self.cfg.n_logpoint + 1)
        _abi.check(self.lib.mbx_read_public(self._h, int(instance), self._pub.ctypes.data_as(_abi.c_double_p), _stream()))
        return self._pub

    def read_state(self, instance):
        n = int(self.lib.mbx_instance_state_doubles(self._h))
        out = np.empty(n)
        _abi.check(self.lib.mbx_debug_read_state(self._h, int(instance), out.ctypes.data_as(_abi.c_double_p)))
        return out

    def write_state(self, instance, block):
        """Overwrite one instance's state block (the layout read_state returns): snapshot / resume, crafted swarms in tests."""
        n = int(self.lib.mbx_instance_state_doubles(self._h))
        block = np.ascontiguousarray(block, dtype=np.float64)
        assert block.shape == (n,)
        _abi.check(self.lib.mbx_debug_write_state(self._h, int(instance), block.ctypes.data_as(_abi.c_double_p)))

    def launch_info(self):
        """How the generation kernel is launched: threads per workgroup, LDS bytes, compile-time-geometry id, state stride in doubles."""
        import ctypes as C
        out = (C.c_int32 * 4)()
        _abi.check(self.lib.mbx_batch_launch_info(self._h, out))
        return {'threads': out[0], 'lds_bytes': out[1], 'fixed_geometry': out[2], 'state_doubles': out[3]}

    def close(self):
        if getattr(self, '_h', None):
            self.lib.mbx_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
