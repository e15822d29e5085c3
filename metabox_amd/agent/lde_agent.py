"""LDE agent: REINFORCE over an LSTM controller that emits per-individual F and CR
(reference: src/agent/lde_agent.py).  PolicyNet = LSTM(NP+10 -> 50, 1 layer) + Linear(50 -> 2NP) for mu and
for sigma (sigmoid); action = clip(Normal(mu, sigma).sample(), 0, 1).

``rollout_batch`` steps a whole BatchedPBO_Env per generation: one LSTM cell + two linear heads over [B, NP+10]
(PyTorch-ROCm GEMMs) and one fused DE generation kernel; (h, c) stay on the device.
"""
import numpy as np
import torch
import torch.nn as nn

from .basic_agent import Basic_Agent
from .utils import save_class


class PolicyNet(nn.Module):
    def __init__(self, config):
        super().__init__()
        self.lstm = nn.LSTM(input_size=config.node_dim, hidden_size=config.CELL_SIZE, num_layers=config.LAYERS_NUM)
        self.mu = nn.Linear(config.CELL_SIZE, config.output_dim_actor)
        self.sigma = nn.Linear(config.CELL_SIZE, config.output_dim_actor)
        self.action_shape = config.action_shape

    def forward(self, x, h, c):
        cell_out, (h_, c_) = self.lstm(x, (h, c))
        return self.mu(cell_out), torch.sigmoid(self.sigma(cell_out)), h_, c_

    def sampler(self, inputs, ht, ct):
        mu, sigma, ht_, ct_ = self.forward(inputs, ht, ct)
        sample_w = torch.clip(torch.distributions.Normal(mu, sigma).sample(), 0, 1).reshape(self.action_shape)
        return sample_w, ht_, ct_

    def packed_weights(self):
        """float32 CUDA tensor in the layout ``mbx_lstm_policy`` documents (include/mbx.h): every matrix transposed, the two LSTM biases
        summed.  Re-packed whenever a parameter has been modified in place (optimizer step, load)."""
        ps = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_pw_key', None) != key:
            l = self.lstm
            parts = [l.weight_ih_l0.detach().t().reshape(-1), l.weight_hh_l0.detach().t().reshape(-1), (l.bias_ih_l0 + l.bias_hh_l0).detach().reshape(-1),
                     self.mu.weight.detach().t().reshape(-1), self.sigma.weight.detach().t().reshape(-1), self.mu.bias.detach().reshape(-1),
                     self.sigma.bias.detach().reshape(-1)]
            self._pw = torch.cat(parts).to(torch.float32).contiguous()
            self._pw_key = key
        return self._pw

    @torch.no_grad()
    def act_batch(self, states, h, c):
        """states [B, NP+10] float32, h/c [1, B, 50] -> actions [B, 2NP], h', c'."""
        mu, sigma, h_, c_ = self.forward(states[None], h, c)
        return torch.clip(mu[0] + sigma[0] * torch.randn_like(mu[0]), 0, 1), h_, c_


_REF_KEYS = {'net/_PolicyNet__lstm.': 'lstm.', 'net/_PolicyNet__mu.': 'mu.', 'net/_PolicyNet__sigma.': 'sigma.'}


class LDE_Agent(Basic_Agent):
    # Under torch.distributed the only training path Trainer drives is train_batch, whose gradients are synchronised over ranks: every rank holds the
    # same parameters, so rank 0 alone writes the checkpoints -- including the `checkpoint0` of __init__ / update_setting (agent/utils.save_class).
    # A class attribute: true before the first train_batch call and not part of the pickled instance state.
    _mbx_replicated = True

    def __init__(self, config):
        super().__init__(config)
        self.__config = config
        self.__BATCH_SIZE = 1
        config.NP = int(getattr(config, 'NP_override', None) or 50)       # lde_agent.py:37
        config.TRAJECTORY_NUM = 20
        config.TRAJECTORY_LENGTH = 50
        config.CELL_SIZE = 50
        config.BINS = 5
        config.LAYERS_NUM = 1
        config.lr_model = 0.005
        config.lr_decay = 1
        config.gamma = 0.99
        config.output_dim_actor = config.NP * 2
        config.action_shape = (1, self.__BATCH_SIZE, config.NP * 2,)
        config.node_dim = config.NP + 2 * config.BINS
        self.__feature_shape = (self.__BATCH_SIZE, config.node_dim,)
        self.__net = PolicyNet(config).to(config.device)
        self.__optimizer = torch.optim.Adam(self.__net.parameters(), lr=config.lr_model)
        self.__learn_steps = 0
        self.__cur_checkpoint = 0
        if getattr(config, 'agent_save_dir', None):
            save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    @property
    def net(self):
        return self.__net

    def load_exported_weights(self, npz):
        sd = {}
        for k in npz.files:
            for pre, new in _REF_KEYS.items():
                if k.startswith(pre):
                    sd[new + k[len(pre):]] = torch.as_tensor(np.asarray(npz[k]))
        self.__net.load_state_dict(sd)
        return self

    def to(self, device):
        self.__config.device = device
        self.__net.to(device)
        return self

    def update_setting(self, config):
        self.__config.max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__learn_steps = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    def __zeros(self, B=1):
        c = self.__config
        return (torch.zeros(c.LAYERS_NUM, B, c.CELL_SIZE, device=c.device), torch.zeros(c.LAYERS_NUM, B, c.CELL_SIZE, device=c.device))

    def rollout_episode(self, env):
        """Single environment, reference loop (lde_agent.py:147-163)."""
        c = self.__config
        is_done = False
        input_net = env.reset()
        h0, c0 = self.__zeros()
        R = 0
        while not is_done:
            with torch.no_grad():
                action, h_, c_ = self.__net.sampler(torch.FloatTensor(input_net[None, :]).to(c.device), h0, c0)
            action = np.squeeze(action.reshape(1, self.__BATCH_SIZE, -1).cpu().numpy(), axis=0)
            next_input, reward, is_done = env.step(action)
            R += np.mean(reward)
            h0, c0 = h_, c_
            input_net = next_input.copy()
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': R}

    @torch.no_grad()
    def policy_step(self, env, state, h, c, policy='hip'):
        """One lock-step generation of a BatchedPBO_Env: act (LSTM cell + heads + sampling over the whole batch) and env.step.
        'hip' (default): the whole PolicyNet as one launch (``mbx_lde_policy``; h, c [1, B, 50] contiguous, updated in place; Philox
        sampling); 'torch': the PyTorch modules (``PolicyNet.act_batch``, torch's generator).  Same distribution either way."""
        if policy == 'hip':
            net = self.__net
            actions = env.batch.lde_policy(net.packed_weights(), net.lstm.hidden_size, h, c)
            state, _, _ = env.step(actions)
            return state, h, c
        actions, h, c = self.__net.act_batch(state.to(torch.float32), h, c)
        state, _, _ = env.step(actions.contiguous())
        return state, h, c

    def policy_route(self, policy='hip'):
        return {'resident': 'mbx_lde_rollout: PolicyNet + update() for up to n generations per launch inside one resident kernel (k_lde_run), LSTM weights from L2',
                'hip': 'mbx_lde_policy: LSTM cell + both heads + sampling in ONE hand-written launch per generation (gate and head products on the float32 matrix cores, weights in L2)',
                'torch': 'PyTorch-ROCm: one LSTM cell + two linear heads over [B, NP + 10] per generation (rocBLAS / hipBLASLt GEMMs)'}[policy]

    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None, policy='resident', gens_per_launch=50):
        """Whole episodes of a lock-step batch.  'resident' (default): ``mbx_lde_rollout`` -- up to `gens_per_launch` generations of PolicyNet.sampler +
        env.step per launch, population / fitness order / features / (h, c) on chip in between (k_lde_run; batches whose geometry or objective kinds
        that kernel does not build are stepped per generation behind the same call); 'hip' / 'torch': one policy launch + one generation launch per
        generation (``policy_step``).  'resident' and 'hip' give bit-identical trajectories."""
        if max_steps is None:
            bc = env.batch.cfg
            max_steps = -(-(bc.max_fes - bc.np) // bc.np)
        state = env.reset()
        h, cc = self.__zeros(env.B)
        if policy == 'resident':
            net = self.__net
            h, cc = h[0].contiguous(), cc[0].contiguous()
            g = 0
            while g < max_steps:
                n = min(int(gens_per_launch), max_steps - g)
                env.batch.lde_rollout(net.packed_weights(), net.lstm.hidden_size, h, cc, n)
                g += n
            res = env.results()
            return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}
        for _ in range(max_steps):
            state, h, cc = self.policy_step(env, state, h, cc, policy)
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}

    def __discounted(self, rewards, n_traj):
        """Discounted returns (gamma = 0.99) exactly as the reference slices them (lde_agent.py:70-83): the flat reward vector is cut into
        n_traj segments of length len(rewards) // n_traj -- which are the trajectories only when all of them have the same length; with
        ragged trajectories (an early `done`) the reference's segments straddle trajectory boundaries and the last len % n_traj rewards
        are dropped.  Kept as is: the loss then averages over the first n_traj * length samples (torch broadcasting would fail otherwise,
        so the caller trims log_prob to the same length)."""
        gamma = self.__config.gamma
        r = np.asarray(rewards, dtype=np.float64)
        length = r.shape[0] // n_traj
        out = r[:n_traj * length].reshape(n_traj, length).copy()
        for t in range(length - 2, -1, -1):
            out[:, t] += gamma * out[:, t + 1]
        return out.reshape(-1)

    def __reinforce_step(self, inputs, hs, cs, actions, returns):
        """One REINFORCE update: loss = -mean(log pi(a + 1e-8 | s, h, c) * discounted return)  (lde_agent.py:124-133)."""
        mean, std, _, _ = self.__net.forward(inputs[None], hs[None], cs[None])
        log_prob = torch.distributions.Normal(mean[0], std[0]).log_prob(actions + 1e-8).sum(1)
        loss = -(log_prob * returns).mean()
        loss.backward()
        return loss

    def __after_update(self):
        c = self.__config
        self.__learn_steps += 1
        if getattr(c, 'agent_save_dir', None) and self.__learn_steps >= c.save_interval * self.__cur_checkpoint:
            save_class(c.agent_save_dir, f'checkpoint{self.__cur_checkpoint}', self)
            self.__cur_checkpoint += 1

    def train_episode(self, env, forced_actions=None):
        """REINFORCE over TRAJECTORY_NUM = 20 trajectories of at most TRAJECTORY_LENGTH = 50 steps, every trajectory
        restarting the environment with a zero LSTM state (reference: lde_agent.py:85-145).  `forced_actions` ([n_steps, 2 NP]) replaces
        the sampled actions (parity test against a recorded reference update)."""
        c = self.__config
        dev = c.device
        self.__optimizer.zero_grad()
        feats, acts, hs, cs, rews = [], [], [], [], []
        total, k = 0, 0
        for _ in range(c.TRAJECTORY_NUM):
            obs = env.reset()
            h, cell = self.__zeros()
            for _step in range(c.TRAJECTORY_LENGTH):
                obs = np.asarray(obs).reshape(self.__feature_shape)
                # no torch.no_grad() here, like the reference: the (h, c) fed to the update below stay connected to the LSTM steps that
                # produced them, so loss.backward() also back-propagates through time along every trajectory (lde_agent.py:99-111, 124-133)
                a, h_next, c_next = self.__net.sampler(torch.FloatTensor(obs[None, :]).to(dev), h, cell)
                a = np.squeeze(a.reshape(1, self.__BATCH_SIZE, -1).detach().cpu().numpy(), axis=0)
                if forced_actions is not None:
                    a = np.asarray(forced_actions[k], dtype=np.float32).reshape(a.shape)
                k += 1
                nxt, reward, done = env.step(a)
                feats.append(obs[0]); acts.append(a[0]); hs.append(h[0, 0]); cs.append(cell[0, 0])
                rews.append(float(np.mean(reward)))
                total += np.mean(reward)
                h, cell, obs = h_next, c_next, np.array(nxt, copy=True)
                if done:
                    break
        returns = torch.FloatTensor(self.__discounted(rews, c.TRAJECTORY_NUM)).to(dev)
        n = returns.shape[0]                                 # == len(rews) unless the trajectories are ragged (see __discounted)
        self.__reinforce_step(torch.FloatTensor(np.stack(feats))[:n].to(dev), torch.stack(hs)[:n], torch.stack(cs)[:n],
                              torch.FloatTensor(np.stack(acts))[:n].to(dev), returns)
        self.__optimizer.step()
        self.__after_update()
        return self.__learn_steps >= c.max_learning_step, {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1],
                                                          'return': total, 'learn_steps': self.__learn_steps}

    def train_batch(self, env, max_updates=None):
        """Batched REINFORCE: the B instances of a lock-step BatchedPBO_Env are the trajectories (the
        reference collects 20 of them one after the other); TRAJECTORY_LENGTH = 50 steps each, then one update, repeated
        until every instance is done.  Finished instances are masked out; gradients are averaged over ranks.  The update's arithmetic
        (log-probability of action + 1e-8, discounted returns, back-propagation through the collection-phase LSTM steps) is the one
        tests/test_training_parity.py pins against the reference for train_episode."""
        from ..distributed import all_ranks_any, average_gradients
        c = self.__config
        dev = env.batch.device
        state = env.reset().to(torch.float32).clone()
        B = env.B
        h, cell = self.__zeros(B)
        alive = torch.ones(B, dtype=torch.bool, device=dev)
        ret_sum = torch.zeros(B, dtype=torch.float64, device=dev)
        updates, exceed, loss = 0, False, torch.zeros(())
        while all_ranks_any(bool(alive.any()), dev) and not exceed:       # global loop control: every rank issues the same collectives
            S, H, C_, A, R, M = [], [], [], [], [], []
            h, cell = h.detach(), cell.detach()                     # the previous segment's graph was consumed by its update
            for _ in range(c.TRAJECTORY_LENGTH):
                # the LSTM steps of the collection phase stay in the autograd graph (see train_episode): the update back-propagates through time
                mu_, sg_, h2, c2 = self.__net.forward(state[None], h, cell)
                a = torch.clip(mu_[0] + sg_[0] * torch.randn_like(mu_[0]), 0, 1).detach()
                S.append(state); H.append(h[0]); C_.append(cell[0]); A.append(a); M.append(alive.clone())
                nstate, reward, done = env.step(a.contiguous())
                r = torch.nan_to_num(reward.to(torch.float32), nan=0.0, posinf=0.0, neginf=0.0) * alive
                R.append(r)
                ret_sum += r.to(torch.float64)
                alive = alive & (done == 0)
                h, cell, state = h2, c2, nstate.to(torch.float32).clone()
                if not bool(alive.any()):
                    break
            M = torch.stack(M).to(torch.float32)
            G = torch.stack(R)
            for t in range(G.shape[0] - 2, -1, -1):
                G[t] += c.gamma * G[t + 1]
            T = G.shape[0]
            self.__optimizer.zero_grad()
            mean, std, _, _ = self.__net.forward(torch.stack(S).view(1, T * B, -1), torch.stack(H).view(1, T * B, -1),
                                                 torch.stack(C_).view(1, T * B, -1))
            logp = torch.distributions.Normal(mean[0], std[0]).log_prob(torch.stack(A).view(T * B, -1) + 1e-8).sum(1).view(T, B)
            loss = -(logp * G * M).sum() / M.sum().clamp_min(1.)
            loss.backward()
            average_gradients(list(self.__net.parameters()), weight=M.sum())
            self.__optimizer.step()
            self.__after_update()
            updates += 1
            if self.__learn_steps >= c.max_learning_step or (max_updates is not None and updates >= max_updates):
                exceed = True
        res = env.results()
        return self.__learn_steps >= c.max_learning_step, {
            'normalizer': float(res['cost'][:, 0].mean()), 'gbest': float(res['cost'][:, -1].mean()),
            'return': float(ret_sum.mean()), 'learn_steps': self.__learn_steps, 'last_losses': (float(loss.detach()),)}
