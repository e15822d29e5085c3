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

    @torch.no_grad()
    def act_batch(self, states, h, c):
        """states [B, NP+10] float32, h/c [1, B, 50] -> actions [B, 2NP], h', c'."""
        mu, sigma, h_, c_ = self.forward(states[None], h, c)
        return torch.clip(mu[0] + sigma[0] * torch.randn_like(mu[0]), 0, 1), h_, c_


_REF_KEYS = {'net/_PolicyNet__lstm.': 'lstm.', 'net/_PolicyNet__mu.': 'mu.', 'net/_PolicyNet__sigma.': 'sigma.'}


class LDE_Agent(Basic_Agent):
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
    def rollout_batch(self, env, max_steps=None):
        if max_steps is None:
            bc = env.batch.cfg
            max_steps = -(-(bc.max_fes - bc.np) // bc.np)
        state = env.reset()
        h, cc = self.__zeros(env.B)
        for _ in range(max_steps):
            actions, h, cc = self.__net.act_batch(state.to(torch.float32), h, cc)
            state, _, _ = env.step(actions.contiguous())
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}

    def __discounted_norm_rewards(self, r):
        c = self.__config
        out = []
        length = r.shape[0] // c.TRAJECTORY_NUM
        for ep in range(c.TRAJECTORY_NUM * self.__BATCH_SIZE):
            single = r[ep * length: ep * length + length]
            disc = np.zeros_like(single)
            run = 0.
            for t in reversed(range(length)):
                run = run * c.gamma + single[t]
                disc[t] = run
            out.append(disc)
        return np.hstack(out)

    def train_episode(self, env):
        """REINFORCE over 20 trajectories x 50 steps, each trajectory restarting the environment (lde_agent.py:85-145)."""
        c = self.__config
        self.__optimizer.zero_grad()
        inputs_b, action_b, hs_b, cs_b, rewards_b = [], [], [], [], []
        R = 0
        for _ in range(c.TRAJECTORY_NUM):
            input_net = env.reset()
            h0, c0 = self.__zeros()
            for _t in range(c.TRAJECTORY_LENGTH):
                input_net = input_net.reshape(self.__feature_shape)
                with torch.no_grad():
                    action, h_, c_ = self.__net.sampler(torch.FloatTensor(input_net[None, :]).to(c.device), h0, c0)
                action = np.squeeze(action.reshape(1, self.__BATCH_SIZE, -1).cpu().numpy(), axis=0)
                inputs_b.append(input_net)
                action_b.append(action)
                next_input, reward, is_done = env.step(action)
                hs_b.append(torch.squeeze(h0, axis=0))
                cs_b.append(torch.squeeze(c0, axis=0))
                rewards_b.append(np.asarray(reward).reshape(self.__BATCH_SIZE))
                R += np.mean(reward)
                h0, c0 = h_, c_
                input_net = next_input.copy()
                if is_done:
                    break
        inputs = np.stack(inputs_b, axis=0).transpose((1, 0, 2)).reshape(-1, c.node_dim)
        actions = np.stack(action_b, axis=0).transpose((1, 0, 2)).reshape(-1, c.output_dim_actor)
        hs = torch.stack(hs_b, axis=0).permute(1, 0, 2).reshape(-1, c.CELL_SIZE)
        cs = torch.stack(cs_b, axis=0).permute(1, 0, 2).reshape(-1, c.CELL_SIZE)
        rewards = np.stack(rewards_b, axis=0).transpose((1, 0)).flatten()
        mean, std, _, _ = self.__net.forward(torch.FloatTensor(inputs[None, :]).to(c.device), hs[None, :], cs[None, :])
        actions = torch.FloatTensor(actions).to(c.device)
        normal = torch.distributions.Normal(torch.squeeze(mean, 0), torch.squeeze(std, 0))
        log_prob = torch.sum(normal.log_prob(actions + 1e-8), 1)
        loss = -torch.mean(log_prob * torch.FloatTensor(self.__discounted_norm_rewards(rewards)).to(c.device))
        loss.backward()
        self.__optimizer.step()
        self.__learn_steps += 1
        if self.__learn_steps >= (c.save_interval * self.__cur_checkpoint):
            save_class(c.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
            self.__cur_checkpoint += 1
        return self.__learn_steps >= c.max_learning_step, {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1],
                                                          'return': R, 'learn_steps': self.__learn_steps}
