"""DE-DDQN agent: a double deep Q-network picks one of four DE mutation operators for every trial vector
(reference: src/agent/de_ddqn_agent.py).

Q-network: MLP 99 -> 100 -> 100 -> 100 -> 100 -> 4 with ReLU.  Rollout is greedy (argmax Q); training is epsilon-greedy
(eps = 0.1) with a 1e5-transition replay buffer, a 1e4-transition warm-up, mini-batches of 64, gamma = 0.99, AdamW
(lr 1e-4) and a target network refreshed every 1000 updates.  ``rollout_batch`` evaluates the Q-network once per step
over the whole instance batch (PyTorch-ROCm GEMMs) and feeds the argmax to the fused DE-DDQN step kernel.
"""
import copy

import numpy as np
import torch

from .basic_agent import Basic_Agent
from .networks import MLP
from .utils import ReplayBuffer, save_class

_HYPER = dict(state_size=99, n_act=4, lr=1e-4, batch_size=64, epsilon=0.1, gamma=0.99, update_target_steps=1000,
              memory_size=100000, warm_up_size=10000)


def _q_layers(n_in, n_out, width=100, depth=4):
    dims = [n_in] + [width] * depth
    layers = [{'in': a, 'out': b, 'drop_out': 0, 'activation': 'ReLU'} for a, b in zip(dims[:-1], dims[1:])]
    return layers + [{'in': width, 'out': n_out, 'drop_out': 0, 'activation': 'None'}]


class DE_DDQN_Agent(Basic_Agent):
    # Under torch.distributed the only training path Trainer drives is train_batch, whose gradients are synchronised over ranks: every rank holds the
    # same parameters, so rank 0 alone writes the checkpoints -- including the `checkpoint0` of __init__ / update_setting (agent/utils.save_class).
    # A class attribute: true before the first train_batch call and not part of the pickled instance state.
    _mbx_replicated = True

    def __init__(self, config):
        super().__init__(config)
        for key, value in _HYPER.items():               # the agent publishes its hyper-parameters on the shared config
            setattr(config, key, value)
        config.net_config = _q_layers(config.state_size, config.n_act)
        self.__config = config
        self.__device = config.device
        self.__pred_func = MLP(config.net_config).to(self.__device)
        self.__target_func = copy.deepcopy(self.__pred_func).to(self.__device)
        self.__optimizer = torch.optim.AdamW(self.__pred_func.parameters(), lr=config.lr)
        self.__criterion = torch.nn.MSELoss()
        self.__replay_buffer = ReplayBuffer(config.memory_size)
        self.__max_learning_step = config.max_learning_step
        self.__global_ls = 0
        self.__cur_checkpoint = 0
        self.__checkpoint()

    # ---- bookkeeping ---------------------------------------------------------------------------------
    def __checkpoint(self):
        if getattr(self.__config, 'agent_save_dir', None):
            save_class(self.__config.agent_save_dir, f'checkpoint{self.__cur_checkpoint}', self)
        self.__cur_checkpoint += 1

    @property
    def q_net(self):
        return self.__pred_func

    def load_exported_weights(self, npz):
        prefix = 'net/'
        self.__pred_func.load_state_dict({k[len(prefix):]: torch.as_tensor(np.asarray(npz[k]))
                                          for k in npz.files if k.startswith(prefix)})
        self.__target_func = copy.deepcopy(self.__pred_func)
        return self

    def to(self, device):
        self.__device = self.__config.device = device
        self.__pred_func.to(device)
        self.__target_func.to(device)
        return self

    def __getstate__(self):
        """Checkpoints carry the networks and the host-side replay like the reference's; the device-resident replay of train_batch
        (up to 80 MB) is rebuilt on demand."""
        return {k: v for k, v in self.__dict__.items() if k != '_dev_replay'}

    def update_setting(self, config):
        self.__max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__config.save_interval = config.save_interval
        self.__global_ls = 0
        self.__cur_checkpoint = 0
        self.__checkpoint()

    # ---- acting ---------------------------------------------------------------------------------------
    def __act(self, state, explore):
        with torch.no_grad():
            q = self.__pred_func(torch.as_tensor(np.asarray(state), dtype=torch.float32, device=self.__device))
        if explore and np.random.rand() < self.__config.epsilon:
            return int(np.random.randint(low=0, high=self.__config.n_act))
        return int(torch.argmax(q))

    def packed_weights(self):
        """float32 CUDA tensor in the layout ``mbx_qnet`` documents (include/mbx.h): per Linear layer the weight transposed, Wt [in][out], then
        the bias.  Rebuilt on every call (the network is 40 k parameters); callers that roll out with fixed weights keep the tensor."""
        parts = []
        for m in self.__pred_func.net:
            if isinstance(m, torch.nn.Linear):
                parts += [m.weight.detach().t().contiguous().reshape(-1), m.bias.detach().reshape(-1)]
        return torch.cat(parts).to(torch.float32).contiguous()

    def qnet_shape(self):
        """(in_dim, width, depth, n_act) of the Q-network."""
        lin = [m for m in self.__pred_func.net if isinstance(m, torch.nn.Linear)]
        return lin[0].in_features, lin[0].out_features, len(lin) - 1, lin[-1].out_features

    @torch.no_grad()
    def greedy_batch(self, states):
        """Greedy operator choice for a batch of states [B, 99] -> int32 [B] (argmax Q, de_ddqn_agent.py:59-68,108-117)."""
        return self.__pred_func(states.to(torch.float32)).argmax(dim=1).to(torch.int32)

    def rollout_episode(self, env):
        state, done, total = env.reset(), False, 0
        while not done:
            state, reward, done = env.step(self.__act(state, explore=False))
            total += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': total}

    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None, graph=False, policy=None):
        """Lock-step rollout: greedy action of the Q-network for the whole batch, then the fused DE-DDQN step kernel.
        policy = 'hip' (default): the Q-network + argmax as ONE launch on the float32 matrix cores (``mbx_ddqn_qnet``, reads the batch's own
        state tensor); 'torch': the PyTorch module (5 small GEMMs + element-wise launches; also what any non-reference architecture uses).
        ``graph=True`` captures the PyTorch forward once into a hipGraph and replays it every step (round 2: not faster than eager).
        Default: ``config.ddqn_policy`` (--ddqn_policy, 'hip').  The two routes evaluate the same float32 network with different summation orders (one
        fma chain per unit in ascending k on the matrix cores / torch's tiled GEMMs), so Q values agree to ~1e-6 and the greedy action can differ where
        two Q values are that close: trajectories are route-dependent (both valid); tests/test_ddqn.py bounds the disagreement.  'torch' reproduces
        results obtained before round 3."""
        if policy is None:
            policy = getattr(self.__config, 'ddqn_policy', 'hip')
        if max_steps is None:
            bc = env.batch.cfg
            max_steps = bc.max_fes - bc.np                # one evaluation per step
        state = env.reset()

        greedy_of = self.greedy_batch
        if policy == 'hip' and not graph and self.qnet_shape() == (99, 100, 4, 4) and state.is_cuda:
            packed = self.packed_weights()
            for _ in range(max_steps):
                env.step(env.batch.ddqn_qnet(packed))
            res = env.results()
            return {k: res[k] for k in ('cost', 'fes', 'return', 'steps', 'cost_len')}

        replay, static_action = None, None
        if graph:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(3):
                    greedy_of(state)
            torch.cuda.current_stream().wait_stream(side)
            replay = torch.cuda.CUDAGraph()
            with torch.cuda.graph(replay):
                static_action = greedy_of(state)             # `state` is the batch's own tensor: later steps overwrite it in place
        for _ in range(max_steps):
            if replay is not None:
                replay.replay()
                env.step(static_action)
            else:
                state, _, _ = env.step(greedy_of(state).contiguous())
        res = env.results()
        return {k: res[k] for k in ('cost', 'fes', 'return', 'steps', 'cost_len')}

    # ---- learning -------------------------------------------------------------------------------------
    def learn_from_batch(self, obs, act, rew, nxt, dn, sync_gradients=False):
        """One double-DQN update on a mini-batch (de_ddqn_agent.py:79-89): MSE between Q(s, a) and r + (1 - done) gamma max_a' Q_target(s', a'),
        AdamW step.  Shared by train_episode (host replay) and train_batch (device replay)."""
        cfg = self.__config
        q_taken = self.__pred_func(obs).gather(1, act.view(-1, 1).long()).squeeze(1)
        with torch.no_grad():
            target = rew + (1 - dn) * cfg.gamma * self.__target_func(nxt).max(1)[0]
        self.__optimizer.zero_grad()
        loss = self.__criterion(q_taken, target)
        loss.backward()
        if sync_gradients:
            from ..distributed import average_gradients
            average_gradients(list(self.__pred_func.parameters()))
        self.__optimizer.step()
        self.__global_ls += 1
        if getattr(cfg, 'agent_save_dir', None) and self.__global_ls >= cfg.save_interval * self.__cur_checkpoint:
            self.__checkpoint()
        return loss

    def __learn_from_replay(self):
        cfg = self.__config
        obs, act, rew, nxt, dn = (t.to(self.__device) for t in self.__replay_buffer.sample(cfg.batch_size))
        self.learn_from_batch(obs, act, rew, nxt, dn)

    def train_batch(self, env, max_updates=None, updates_per_step=1):
        """Double-DQN training over a lock-step BatchedPBO_Env.  Every env step all B instances act epsilon-greedily on the device,
        their B transitions go into a device-resident FIFO replay (capacity memory_size, the oldest rows are overwritten), and once
        it holds warm_up_size transitions `updates_per_step` mini-batch updates (batch_size = 64, Huber-free MSE on the TD target with
        the target network, refreshed every update_target_steps updates) follow -- the reference's loop (de_ddqn_agent.py:70-106)
        with a batch axis.  By construction the data : update ratio is B times the reference's.  Gradients are averaged across ranks.
        Returns (exceed_max_learning_step, {'normalizer', 'gbest', 'return', 'learn_steps'})."""
        from ..distributed import all_ranks_any
        cfg, dev = self.__config, env.batch.device
        net, tgt = self.__pred_func, self.__target_func
        B, S = env.B, cfg.state_size
        cap = (cfg.memory_size // B) * B if cfg.memory_size >= B else B
        if getattr(self, '_dev_replay', None) is None or self._dev_replay['obs'].shape[0] != cap or self._dev_replay['obs'].device != dev:
            self._dev_replay = dict(obs=torch.empty(cap, S, device=dev), nxt=torch.empty(cap, S, device=dev),
                                    act=torch.empty(cap, dtype=torch.int64, device=dev), rew=torch.empty(cap, device=dev),
                                    done=torch.empty(cap, device=dev), size=0, head=0)
        rb = self._dev_replay
        state = env.reset().to(torch.float32).clone()
        alive = torch.ones(B, dtype=torch.bool, device=dev)
        ret_sum = torch.zeros(B, dtype=torch.float64, device=dev)
        updates, exceed = 0, False
        while all_ranks_any(bool(alive.any()), dev) and not exceed:      # global loop control: every rank issues the same collectives
            with torch.no_grad():
                greedy = net(state).argmax(dim=1)
                explore = torch.rand(B, device=dev) < cfg.epsilon
                action = torch.where(explore, torch.randint(0, cfg.n_act, (B,), device=dev), greedy)
            nstate, reward, done = env.step(action.to(torch.int32).contiguous())
            nstate = nstate.to(torch.float32).clone()
            ret_sum += reward * alive
            live = alive.nonzero(as_tuple=True)[0]                  # finished instances contribute no transitions
            n = int(live.numel())
            if n:
                slots = (rb['head'] + torch.arange(n, device=dev)) % cap
                rb['obs'][slots] = state[live]; rb['nxt'][slots] = nstate[live]; rb['act'][slots] = action[live]
                rb['rew'][slots] = reward[live].to(torch.float32); rb['done'][slots] = (done[live] != 0).to(torch.float32)
                rb['head'] = (rb['head'] + n) % cap
                rb['size'] = min(cap, rb['size'] + n)
            alive = alive & (done == 0)
            state = nstate
            # warm-up reached on ANY rank starts the updates on EVERY rank (same number of gradient all-reduces everywhere); a rank that is
            # still short samples its mini-batch, with replacement, from what it has
            if all_ranks_any(rb['size'] >= min(cfg.warm_up_size, cap), dev) and rb['size'] >= 1:
                for _ in range(updates_per_step):
                    idx = torch.randint(0, rb['size'], (cfg.batch_size,), device=dev)
                    self.learn_from_batch(rb['obs'][idx], rb['act'][idx], rb['rew'][idx], rb['nxt'][idx], rb['done'][idx], sync_gradients=True)
                    updates += 1
                    if self.__global_ls % cfg.update_target_steps == 0:
                        tgt.load_state_dict(net.state_dict())
                    if self.__global_ls >= self.__max_learning_step or (max_updates is not None and updates >= max_updates):
                        exceed = True
                        break
        res = env.results()
        return self.__global_ls >= self.__max_learning_step, {
            'normalizer': float(res['cost'][:, 0].mean()), 'gbest': float(res['cost'][:, -1].mean()),
            'return': float(ret_sum.mean()), 'learn_steps': self.__global_ls}

    def train_episode(self, env):
        """One episode of epsilon-greedy interaction with a replay update after every step once the buffer holds
        warm_up_size transitions (reference loop: de_ddqn_agent.py:70-106)."""
        cfg = self.__config
        state, done, total = env.reset(), False, 0
        while not done:
            action = self.__act(state, explore=True)
            nxt, reward, done = env.step(action)
            total += reward
            self.__replay_buffer.append((state, action, reward, nxt, done))
            if len(self.__replay_buffer) >= cfg.warm_up_size:
                self.__learn_from_replay()
                if self.__global_ls >= self.__max_learning_step:
                    break
            if self.__global_ls % cfg.update_target_steps == 0:
                self.__target_func.load_state_dict(self.__pred_func.state_dict())
            state = nxt
        return self.__global_ls >= self.__max_learning_step, {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1],
                                                              'return': total, 'learn_steps': self.__global_ls}
