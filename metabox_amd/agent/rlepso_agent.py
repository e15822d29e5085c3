"""RLEPSO agent: PPO actor-critic over the scalar state fes/maxFEs (reference: src/agent/rlepso_agent.py).

Actor : two MLPs 1 -> 64 -> 32 -> 35 (ReLU); mu = (tanh+1)/2, sigma = (tanh+1)/2*(max_sigma-min_sigma)+min_sigma,
        action = clamp(Normal(mu, sigma).sample(), 0, 1)                      (rlepso_agent.py:9-47)
Critic: MLP 1 -> 16 -> 8 -> 1                                                  (rlepso_agent.py:50-61)

``rollout_episode`` / ``train_episode`` keep the reference's single-environment semantics.  ``rollout_batch``
is the MI355X path: one policy forward over the whole instance batch per generation (PyTorch-ROCm GEMMs), one
fused generation kernel, no host synchronisation inside the episode.
"""
import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from .basic_agent import Basic_Agent
from .networks import MLP
from .utils import save_class


class Actor(nn.Module):
    def __init__(self, config):
        super().__init__()
        net = [{'in': config.feature_dim, 'out': 64, 'drop_out': 0, 'activation': 'ReLU'},
               {'in': 64, 'out': 32, 'drop_out': 0, 'activation': 'ReLU'},
               {'in': 32, 'out': config.action_dim, 'drop_out': 0, 'activation': 'None'}]
        self.mu_net = MLP(net)
        self.sigma_net = MLP(net)
        self.max_sigma = config.max_sigma
        self.min_sigma = config.min_sigma

    def distribution(self, x):
        mu = (torch.tanh(self.mu_net(x)) + 1.) / 2.
        sigma = (torch.tanh(self.sigma_net(x)) + 1.) / 2. * (self.max_sigma - self.min_sigma) + self.min_sigma
        return mu, sigma

    def forward(self, x, fixed_action=None, require_entropy=False):
        mu, sigma = self.distribution(x)
        policy = Normal(mu, sigma)
        action = fixed_action if fixed_action is not None else torch.clamp(policy.sample(), min=0, max=1)
        log_prob = torch.sum(policy.log_prob(action))
        if require_entropy:
            return action, log_prob, policy.entropy()
        return action, log_prob

    def _fused_weights(self):
        """The mu- and sigma-networks share their input and their shapes, so one generation needs 3 (batched) GEMMs instead
        of 6: layer 1 of both nets is one [1 -> 128] affine map, layers 2 and 3 are 2-batch bmm's."""
        # keyed on (storage, in-place version): optimizer.step() and load_state_dict() modify parameters in place
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + (str(next(self.parameters()).device),)
        if getattr(self, '_fw_key', None) != key:
            nets = (self.mu_net.net, self.sigma_net.net)
            lin = [[n._modules[f'layer{i}-linear'] for n in nets] for i in range(3)]
            self._fw = dict(
                w1=torch.cat([l.weight[:, 0] for l in lin[0]])[None, :].contiguous(), b1=torch.cat([l.bias for l in lin[0]])[None, :].contiguous(),
                w2=torch.stack([l.weight.t() for l in lin[1]]).contiguous(), b2=torch.stack([l.bias for l in lin[1]])[:, None, :].contiguous(),
                w3=torch.stack([l.weight.t() for l in lin[2]]).contiguous(), b3=torch.stack([l.bias for l in lin[2]])[:, None, :].contiguous())
            # (tanh + 1)/2 and (tanh + 1)/2 * (max_sigma - min_sigma) + min_sigma as ONE affine map of the stacked tanh output:
            # row 0 (mu) = 0.5 t + 0.5 (bit-identical to (t + 1)/2), row 1 (sigma) = s t + (s + min_sigma), s = (max - min)/2
            half = 0.5 * (self.max_sigma - self.min_sigma)
            dev = self._fw['w1'].device
            self._fw['scale'] = torch.tensor([0.5, half], dtype=torch.float32, device=dev).view(2, 1, 1)
            self._fw['shift'] = torch.tensor([0.5, half + self.min_sigma], dtype=torch.float32, device=dev).view(2, 1, 1)
            self._fw_key = key
        return self._fw

    def packed_weights(self):
        """float32 CUDA tensor in the layout ``mbx_gauss_mlp`` documents (include/mbx.h): per net W1^T | b1 | W2^T | b2 | W3^T | b3.
        Re-packed whenever a parameter has been modified in place (optimizer step, load)."""
        ps = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_pw_key', None) != key:
            parts = []
            for n in (self.mu_net.net, self.sigma_net.net):
                for i in range(3):
                    lin = n._modules[f'layer{i}-linear']
                    parts += [lin.weight.detach().t().reshape(-1), lin.bias.detach().reshape(-1)]
            self._pw = torch.cat(parts).to(torch.float32).contiguous()
            self._pw_key = key
        return self._pw

    def hidden_sizes(self):
        n = self.mu_net.net
        return n._modules['layer0-linear'].out_features, n._modules['layer1-linear'].out_features

    @torch.no_grad()
    def act_batch(self, states):
        """[B, 1] float32 -> [B, 35] float32 actions for B independent environments (both MLPs evaluated every call)."""
        f = self._fused_weights()
        B = states.shape[0]
        h = torch.relu_(torch.addcmul(f['b1'], states, f['w1']))                    # [B, 128]
        h = h.view(B, 2, -1).transpose(0, 1)                                        # [2, B, 64]
        h = torch.relu_(torch.baddbmm(f['b2'], h, f['w2']))                         # [2, B, 32]
        t = torch.tanh_(torch.baddbmm(f['b3'], h, f['w3']))                         # [2, B, 35]
        ms = torch.addcmul(f['shift'], t, f['scale'])                               # [mu ; sigma]
        # mu + sigma * eps == Normal(mu, sigma).sample(); unlike torch.normal(tensor, tensor) it has no host-side check and
        # can be captured into a hipGraph
        return torch.addcmul(ms[0], ms[1], torch.randn_like(ms[0])).clamp_(0, 1)


class ActorTable:
    """Batched sampling for the RLEPSO actor.

    The optimizer's state is the scalar fes/maxFEs (rlepso_optimizer.py:170-171) and fes is an integer below
    maxFEs + 2 NP, so (mu, sigma) take at most that many distinct values.  They are evaluated ONCE with one batched
    forward over every possible state; a generation then costs one row gather and the Normal sampling instead of two
    3-layer MLPs over the whole instance batch (~23 small kernels -> 5).  Same float32 arithmetic, same weights."""

    def __init__(self, actor, max_fes, np_, device):
        self.max_fes = int(max_fes)
        k = torch.arange(0, self.max_fes + 2 * int(np_) + 2, dtype=torch.float64, device=device)
        with torch.no_grad():
            mu, sigma = actor.distribution((k / self.max_fes).to(torch.float32)[:, None])
        self.table = torch.cat([mu, sigma], dim=1).contiguous()        # [K, 2*action_dim]
        self.adim = mu.shape[1]

    @torch.no_grad()
    def act(self, state):
        """state: [B, 1] float64 device tensor (fes/maxFEs) -> [B, action_dim] float32 actions."""
        idx = torch.round(state[:, 0] * self.max_fes).to(torch.int64).clamp_(0, self.table.shape[0] - 1)
        ms = self.table.index_select(0, idx)
        mu, sigma = ms[:, :self.adim], ms[:, self.adim:]
        return torch.addcmul(mu, sigma, torch.randn_like(mu)).clamp_(0, 1)


class Critic(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.value_head = MLP([{'in': config.feature_dim, 'out': 16, 'drop_out': 0, 'activation': 'ReLU'},
                               {'in': 16, 'out': 8, 'drop_out': 0, 'activation': 'ReLU'},
                               {'in': 8, 'out': 1, 'drop_out': 0, 'activation': 'None'}])

    def forward(self, h):
        v = self.value_head(h)
        return v.detach().squeeze(), v.squeeze()


_REF_PREFIX = {'actor/_Actor__mu_net.': 'mu_net.', 'actor/_Actor__sigma_net.': 'sigma_net.',
               'critic/_Critic__value_head.': 'value_head.'}


class RLEPSO_Agent(Basic_Agent):
    # Under torch.distributed the only training path Trainer drives is train_batch, whose gradients are synchronised over ranks: every rank holds the
    # same parameters, so rank 0 alone writes the checkpoints -- including the `checkpoint0` of __init__ / update_setting (agent/utils.save_class).
    # A class attribute: true before the first train_batch call and not part of the pickled instance state.
    _mbx_replicated = True

    def __init__(self, config):
        super().__init__(config)
        # the agent writes its hyper-parameters into the shared config like the reference (rlepso_agent.py:69-78)
        config.feature_dim = 1
        config.action_dim = 35
        config.action_shape = (35,)
        config.n_step = 10
        config.K_epochs = 3
        config.eps_clip = 0.1
        config.gamma = 0.999
        config.max_sigma = 0.7
        config.min_sigma = 0.01
        config.lr = 1e-5
        self.__config = config
        self.__device = config.device
        self.__actor = Actor(config).to(self.__device)
        self.__critic = Critic(config).to(self.__device)
        self.__optimizer_actor = torch.optim.Adam([{'params': self.__actor.parameters(), 'lr': config.lr}])
        self.__optimizer_critic = torch.optim.Adam([{'params': self.__critic.parameters(), 'lr': config.lr}])
        self.__learning_time = 0
        self.__cur_checkpoint = 0
        if getattr(config, 'agent_save_dir', None):
            save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    # ---- weights -----------------------------------------------------------------------------------
    @property
    def actor(self):
        return self.__actor

    @property
    def critic(self):
        return self.__critic

    def actor_table(self, max_fes, np_, device):
        key = (int(max_fes), int(np_), str(device))
        cache = self.__dict__.setdefault('_tables', {})
        if key not in cache:
            cache[key] = ActorTable(self.__actor, max_fes, np_, device)
        return cache[key]

    def load_exported_weights(self, npz):
        """Load the arrays exported from a reference checkpoint by tools/gen_golden.py (`policy` section)."""
        sd_a, sd_c = {}, {}
        for k in npz.files if hasattr(npz, 'files') else npz:
            for pre, new in _REF_PREFIX.items():
                if k.startswith(pre):
                    (sd_c if pre.startswith('critic') else sd_a)[new + k[len(pre):]] = torch.as_tensor(np.asarray(npz[k]))
        self.__actor.load_state_dict(sd_a)
        self.__critic.load_state_dict(sd_c)
        self.__dict__['_tables'] = {}
        return self

    def to(self, device):
        self.__device = device
        self.__config.device = device
        self.__actor.to(device)
        self.__critic.to(device)
        self.__dict__['_tables'] = {}
        return self

    def update_setting(self, config):
        self.__config.max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__learning_time = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    # ---- rollout -----------------------------------------------------------------------------------
    def rollout_episode(self, env):
        """Single environment, reference loop (rlepso_agent.py:294-303)."""
        is_done = False
        state = env.reset()
        R = 0
        while not is_done:
            state = torch.FloatTensor(state).to(self.__device)
            with torch.no_grad():
                action = self.__actor(state)[0].cpu().numpy()
            state, reward, is_done = env.step(action)
            R += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': R}

    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None, policy='resident'):
        """Rollout of a BatchedPBO_Env: no host sync inside the episode.  ``policy`` selects how the actor runs:
        'resident' the WHOLE episode in one launch (``mbx_rlepso_rollout``): the actor is evaluated once per rollout at every reachable
                state (``mbx_rlepso_policy_table``), every workgroup draws its own actions and keeps its instance's state on chip from the
                first generation to the last -- the default; bit-identical to 'fused';
        'fused' act + step in ONE launch per generation (``mbx_rlepso_act_step``), same table, same draws;
        'hip'   one ``mbx_gauss_policy`` launch per generation (weights in LDS), then ``mbx_step``; bit-identical to 'fused';
        'torch' the two MLPs as batched PyTorch GEMMs (``Actor.act_batch``, torch's generator);
        'table' (mu, sigma) gathered from the per-fes table of the actor evaluated once (``ActorTable``).
        All of them sample the same distribution; 'torch' and 'table' use torch's generator instead of the instance's Philox stream.

        Every update() bills at least NP evaluations, so after ceil((maxFEs-NP)/NP) generations every instance has
        reached ``fes >= maxFEs``; instances that finish earlier idle inside the kernel.
        """
        bc = env.batch.cfg                           # the optimizer's own NP / maxFEs (it may differ from the agent's config copy)
        if max_steps is None:
            max_steps = -(-(bc.max_fes - bc.np) // bc.np)
        actor = self.__actor
        if policy in ('fused', 'resident'):
            h1, h2 = actor.hidden_sizes()
            table = env.batch.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
            env.reset()
            if policy == 'resident':
                env.batch.rlepso_rollout(table, max_steps)
            else:
                for _ in range(max_steps):
                    env.batch.act_step(table)
            res = env.results()
            return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'],
                    'cost_len': res['cost_len']}
        if policy == 'hip':
            h1, h2 = actor.hidden_sizes()

            def act(_state):
                return env.batch.gauss_policy(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        elif policy == 'table':
            act = self.actor_table(bc.max_fes, bc.np, env.batch.device).act
        elif policy == 'torch':
            def act(state):
                return actor.act_batch(state.to(torch.float32))
        else:
            raise ValueError(f"policy must be 'resident', 'fused', 'hip', 'torch' or 'table', not {policy!r}")
        state = env.reset()
        for _ in range(max_steps):
            state, _, _ = env.step(act(state))
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'],
                'cost_len': res['cost_len']}

    @torch.no_grad()
    def collect_segment_resident(self, env, state, alive, n_step):
        """The n_step transitions of one PPO segment (rlepso_agent.py:143-190) from ONE ``mbx_rlepso_rollout`` launch with the CURRENT
        weights: -> (S [T, B, 1] float32 state before each generation, A [T, B, 35] float32 actions the kernel drew, M [T, B] bool alive
        before each generation, R [T, B] float64 rewards, state after the segment [B, 1] float32, alive after the segment [B])."""
        actor, batch = self.__actor, env.batch
        h1, h2 = actor.hidden_sizes()
        table = batch.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
        _, _, _, traj = batch.rlepso_rollout(table, n_step, trajectory=True)
        after = traj['state'].to(torch.float32)                                   # state after generation t
        done_after = traj['done'] != 0                                            # absorbing: stays 1 once the instance terminated
        S = torch.cat([state[None], after[:-1, :, None]], 0)
        M = torch.cat([alive[None], alive[None] & ~done_after[:-1]], 0)
        return S, traj['actions'], M, traj['reward'], after[-1][:, None].clone(), alive & ~done_after[-1]

    # ---- training: PPO (reference rlepso_agent.py:113-292), one implementation for B >= 1 -------------------------------------
    def train_batch(self, env, max_updates=None, forced_actions=None, collect='auto'):
        """PPO over a lock-step batch of environments: n_step = 10 segments, K_epochs = 3 optimizer steps per segment, clipped
        surrogate + clipped value loss, n-step returns bootstrapped from the critic at the segment's last state (rlepso_agent.py:140-276).
        Every quantity carries a leading [T, B] shape and the losses average over the (step, instance) pairs that were still running;
        at B = 1 this IS the reference's update (tests/test_training_parity.py replays reference segments: same gradients, same weights).
        For B > 1 one optimizer step consumes B trajectories -- a semantic difference by construction -- and `learn_steps` counts
        optimizer steps.  With torch.distributed initialised, gradients are averaged over ranks and the loop control is global
        (every rank runs the same number of segments and optimizer steps; a rank whose shard has finished contributes zero gradients).

        env: BatchedPBO_Env-like (B, reset() -> [B, 1], step(actions [B, 35]) -> (state, reward, done), results()).
        forced_actions: optional [T_total, B, 35] tensor replayed instead of sampling (parity tests).
        collect: how a segment's n_step transitions are gathered.  'step': actor forward + sampling in PyTorch and one env.step per
                 generation (the reference's loop, rlepso_agent.py:143-190).  'resident': ONE ``mbx_rlepso_rollout`` launch per segment --
                 the current weights' (mu, sigma) table is rebuilt (``mbx_rlepso_policy_table``), the kernel draws the actions itself
                 (the instance's Philox stream instead of torch's generator: same distribution), keeps the state on chip for the n_step
                 generations and hands back the per-generation records (actions, state, reward, done); log-probabilities and values are
                 then evaluated over the whole [n_step, B] block at once.  'auto' (default): 'resident' when the environment is a
                 BatchedPBO_Env on the GPU and no actions are forced, else 'step'.
        Returns (exceed_max_learning_step, {'normalizer', 'gbest', 'return', 'learn_steps', 'last_losses'})."""
        from ..distributed import all_ranks_any, average_gradients
        config = self.__config
        gamma, n_step, K_epochs, eps_clip = config.gamma, config.n_step, config.K_epochs, config.eps_clip
        actor, critic = self.__actor, self.__critic
        params = list(actor.parameters()) + list(critic.parameters())
        B = env.B
        state = env.reset().to(torch.float32).clone()                     # [B, 1]
        dev = state.device
        alive = torch.ones(B, dtype=torch.bool, device=dev)
        ret_sum = torch.zeros(B, dtype=torch.float64, device=dev)
        updates, exceed, t_all = 0, False, 0
        baseline_loss = reinforce_loss = torch.zeros((), device=dev)

        def evaluate(states, actions):
            mu, sigma = actor.distribution(states)
            return Normal(mu, sigma).log_prob(actions).sum(-1), critic.value_head(states).squeeze(-1)

        batch = getattr(env, 'batch', None)
        if collect == 'auto':
            collect = 'resident' if (forced_actions is None and hasattr(batch, 'rlepso_rollout') and state.is_cuda) else 'step'
        if collect not in ('step', 'resident') or (collect == 'resident' and (forced_actions is not None or not hasattr(batch, 'rlepso_rollout'))):
            raise ValueError("collect must be 'auto', 'step' or 'resident' (the latter needs a BatchedPBO_Env and no forced actions)")

        while all_ranks_any(bool(alive.any()), dev) and not exceed:
            S, A, LP, V, R, M = [], [], [], [], [], []
            if collect == 'resident':
                S_blk, A_blk, M_blk, R_blk, state, alive_next = self.collect_segment_resident(env, state, alive, n_step)
                logp_blk, val_blk = evaluate(S_blk.reshape(-1, S_blk.shape[-1]), A_blk.reshape(-1, A_blk.shape[-1]))
                S, A, M = list(S_blk), list(A_blk), list(M_blk)
                LP, V = list(logp_blk.view(M_blk.shape)), list(val_blk.view(M_blk.shape))
                R = list(R_blk.to(torch.float32))
                ret_sum += (R_blk * M_blk).sum(0)
                alive = alive_next
                t_all += n_step
            for _ in range(n_step if collect == 'step' else 0):
                if forced_actions is not None:
                    action = forced_actions[t_all].to(dev)
                else:
                    with torch.no_grad():
                        mu, sigma = actor.distribution(state)
                        action = torch.clamp(mu + sigma * torch.randn_like(mu), 0, 1)
                t_all += 1
                logp, val = evaluate(state, action)
                S.append(state); A.append(action); LP.append(logp); V.append(val); M.append(alive.clone())
                nstate, reward, done = env.step(action.contiguous())
                R.append(reward.to(torch.float32).clone())
                ret_sum += reward * alive
                alive = alive & (done == 0)
                state = nstate.to(torch.float32).clone()
                if not bool(alive.any()):
                    break
            S, A, M = torch.stack(S), torch.stack(A), torch.stack(M).to(torch.float32)      # [T, B, ...]
            R = torch.stack(R) * M
            old_logp = torch.stack(LP).detach()
            n_live = M.sum().clamp_min(1.)
            old_value = None
            for k in range(K_epochs):
                if k == 0:
                    logp, val = torch.stack(LP), torch.stack(V)
                else:
                    logp, val = evaluate(S.reshape(-1, S.shape[-1]), A.reshape(-1, A.shape[-1]))
                    logp, val = logp.view(M.shape), val.view(M.shape)
                with torch.no_grad():                                        # n-step bootstrapped returns (:225-237)
                    Rt = critic.value_head(state).squeeze(-1)
                    returns = []
                    for t in reversed(range(R.shape[0])):
                        Rt = torch.where(M[t] > 0, Rt * gamma + R[t], Rt)
                        returns.append(Rt)
                    returns = torch.stack(returns[::-1])
                ratios = torch.exp(logp - old_logp)
                adv = returns - val.detach()
                surr = torch.min(ratios * adv, torch.clamp(ratios, 1 - eps_clip, 1 + eps_clip) * adv)
                reinforce_loss = -(surr * M).sum() / n_live
                if old_value is None:
                    baseline_loss = (((val - returns) ** 2) * M).sum() / n_live
                    old_value = val.detach()
                else:
                    vclip = old_value + torch.clamp(val - old_value, -eps_clip, eps_clip)
                    baseline_loss = (torch.max((val - returns) ** 2, (vclip - returns) ** 2) * M).sum() / n_live
                self.__optimizer_actor.zero_grad()
                self.__optimizer_critic.zero_grad()
                (baseline_loss + reinforce_loss).backward()
                average_gradients(params, weight=M.sum())          # weight = live (step, instance) pairs: the global mean loss for any world size
                self.__optimizer_actor.step()
                self.__optimizer_critic.step()
                self.__dict__['_tables'] = {}
                self.__learning_time += 1
                updates += 1
                if getattr(config, 'agent_save_dir', None) and self.__learning_time >= config.save_interval * self.__cur_checkpoint:
                    save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
                    self.__cur_checkpoint += 1
                if self.__learning_time >= config.max_learning_step or (max_updates is not None and updates >= max_updates):
                    exceed = True
                    break
        res = env.results()
        return self.__learning_time >= config.max_learning_step, {
            'normalizer': float(res['cost'][:, 0].mean()), 'gbest': float(res['cost'][:, -1].mean()),
            'return': float(ret_sum.mean()), 'learn_steps': self.__learning_time,
            'last_losses': (float(baseline_loss.detach()), float(reinforce_loss.detach()))}

    def train_episode(self, env):
        """The reference's single-environment entry point (rlepso_agent.py:113-292): the B = 1 case of train_batch over a one-instance
        view of the PBO_Env."""
        exceed, info = self.train_batch(SingleEnvBatch(env, self.__device))
        info.pop('last_losses', None)
        info['normalizer'], info['gbest'] = env.optimizer.cost[0], env.optimizer.cost[-1]
        return exceed, info


class SingleEnvBatch:
    """B = 1 lock-step view of a reference-protocol PBO_Env (reset() -> state, step(action) -> (state, reward, done) with numpy /
    Python scalars), for the agents' batched training loops."""

    def __init__(self, env, device):
        self.env, self.B, self.device = env, 1, torch.device(device)

    def _state(self, s):
        return torch.as_tensor(np.asarray(s, dtype=np.float64).reshape(1, -1), device=self.device)

    def reset(self):
        return self._state(self.env.reset())

    def step(self, actions):
        s, r, d = self.env.step(actions[0].detach().cpu().numpy())
        return (self._state(s), torch.as_tensor([float(np.mean(r))], dtype=torch.float64, device=self.device),
                torch.as_tensor([1 if d else 0], dtype=torch.uint8, device=self.device))

    def results(self):
        c = np.asarray(self.env.optimizer.cost, dtype=np.float64).reshape(1, -1)
        return {'cost': torch.as_tensor(c)}
