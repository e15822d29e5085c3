"""RL-PSO agent: a Gaussian policy outputs, for the particle that moves next, the weight of its attraction towards gbest
(reference: src/agent/rl_pso_agent.py).

PolicyNetwork: two MLPs 2D -> 32 -> 8 -> 1 (ReLU) sharing their input; mu = (tanh + 1)/2, sigma = clamp((tanh + 1)/2, 0.01,
0.7); the action is a Normal sample, re-folded by ``(a + 3 sigma - mu) * (1/6 sigma)`` when it leaves [0, 1).  Training is
per-step REINFORCE (loss = -log_prob * reward, one Adam step per env step, lr 1e-5).

``rollout_episode`` / ``train_episode`` keep the reference's single-environment protocol.  ``rollout_batch`` runs the whole
episode of a lock-step batch with the actor INSIDE the step kernel, `chunk` env steps per launch (``mbx_rlpso_rollout``):
an RL-PSO step is D elements of arithmetic and one evaluation, so anything per-step on the host is pure launch latency.
"""
import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from .basic_agent import Basic_Agent
from .networks import MLP
from .utils import save_class


class PolicyNetwork(nn.Module):
    def __init__(self, config):
        super().__init__()
        net = [{'in': config.feature_dim, 'out': 32, 'drop_out': 0, 'activation': 'ReLU'},
               {'in': 32, 'out': 8, 'drop_out': 0, 'activation': 'ReLU'},
               {'in': 8, 'out': config.action_dim, 'drop_out': 0, 'activation': 'None'}]
        self.mu_net = MLP(net)
        self.sigma_net = MLP(net)
        self.max_sigma = config.max_sigma
        self.min_sigma = config.min_sigma

    def distribution(self, x):
        mu = (torch.tanh(self.mu_net(x)) + 1.) / 2.
        sigma = torch.clamp((torch.tanh(self.sigma_net(x)) + 1.) / 2., min=self.min_sigma, max=self.max_sigma)
        return mu, sigma

    def forward(self, x, require_entropy=False, require_musigma=False):
        mu, sigma = self.distribution(x)
        policy = Normal(mu, sigma)
        action = policy.sample()
        outside = torch.abs(action - 0.5) >= 0.5
        action = torch.where(outside, (action + 3 * sigma.detach() - mu.detach()) * (1. / 6 * sigma.detach()), action)
        log_prob = policy.log_prob(action)
        if require_entropy:
            return action, log_prob, policy.entropy()
        if require_musigma:
            return action, log_prob, mu, sigma
        return action, log_prob

    def packed_weights(self):
        """float32 CUDA tensor in the ``mbx_gauss_mlp`` layout (include/mbx.h): per net W1^T | b1 | W2^T | b2 | W3^T | b3."""
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


_REF_PREFIX = {'nets/_PolicyNetwork__mu_net.': 'mu_net.', 'nets/_PolicyNetwork__sigma_net.': 'sigma_net.'}


class RL_PSO_Agent(Basic_Agent):
    def __init__(self, config):
        super().__init__(config)
        config.feature_dim = 2 * config.dim            # rl_pso_agent.py:53-58
        config.action_dim = 1
        config.action_shape = (1,)
        config.max_sigma = 0.7
        config.min_sigma = 0.01
        config.lr = 1e-5
        self.__config = config
        self.__device = config.device
        self.__nets = PolicyNetwork(config).to(self.__device)
        self.__optimizer = torch.optim.Adam([{'params': self.__nets.parameters(), 'lr': config.lr}])
        self.__learning_time = 0
        self.__cur_checkpoint = 0
        if getattr(config, 'agent_save_dir', None):
            save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    @property
    def nets(self):
        return self.__nets

    def load_exported_weights(self, npz):
        """Load the arrays exported from a reference checkpoint by tools/gen_golden.py (`rlpso` section)."""
        sd = {}
        for k in npz.files if hasattr(npz, 'files') else npz:
            for pre, new in _REF_PREFIX.items():
                if k.startswith(pre):
                    sd[new + k[len(pre):]] = torch.as_tensor(np.asarray(npz[k]))
        self.__nets.load_state_dict(sd)
        return self

    def to(self, device):
        self.__device = device
        self.__config.device = device
        self.__nets.to(device)
        return self

    def update_setting(self, config):
        self.__config.max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__learning_time = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    # ---- reference protocol (one environment) ------------------------------------------------------------------------
    def train_episode(self, env):
        """Per-step REINFORCE (rl_pso_agent.py:77-116)."""
        config = self.__config
        state = torch.FloatTensor(env.reset()).to(self.__device)
        exceed, R = False, 0
        while True:
            action, log_prob = self.__nets(state)
            state, reward, is_done = env.step(action.reshape(config.action_shape).detach().cpu().numpy())
            R += reward
            state = torch.FloatTensor(state).to(self.__device)
            loss = (-log_prob * reward).mean()
            self.__optimizer.zero_grad()
            loss.backward()
            self.__optimizer.step()
            self.__learning_time += 1
            if self.__learning_time >= config.save_interval * self.__cur_checkpoint:
                save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
                self.__cur_checkpoint += 1
            if self.__learning_time >= config.max_learning_step:
                exceed = True
                break
            if is_done:
                break
        return exceed, {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1], 'return': R,
                        'learn_steps': self.__learning_time}

    @torch.no_grad()
    def rollout_episode(self, env):
        is_done, R = False, 0
        state = env.reset()
        while not is_done:
            action, _ = self.__nets(torch.FloatTensor(state).to(self.__device))
            state, reward, is_done = env.step(action.cpu().numpy())
            R += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': R}

    # ---- lock-step batch -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None, chunk=256, policy='fused'):
        """Whole episodes of a BatchedPBO_Env.  policy = 'fused': `chunk` env steps per launch with the actor inside the kernel
        (``mbx_rlpso_rollout``); 'hip': ``mbx_gauss_policy`` + ``mbx_step`` per step (bit-identical to 'fused'); 'torch': the
        PyTorch modules per step (torch's generator)."""
        bc = env.batch.cfg
        if max_steps is None:
            max_steps = bc.max_fes - bc.np                     # every step bills exactly one evaluation
        nets = self.__nets
        h1, h2 = nets.hidden_sizes()
        net = (nets.packed_weights(), h1, h2, nets.min_sigma, nets.max_sigma)
        state = env.reset()
        if policy == 'fused':
            left = max_steps
            while left > 0:
                env.batch.rlpso_rollout(*net, min(chunk, left))
                left -= chunk
        elif policy == 'hip':
            for _ in range(max_steps):
                env.step(env.batch.gauss_policy(*net))
        elif policy == 'torch':
            for _ in range(max_steps):
                action, _ = nets(state.to(torch.float32))
                state, _, _ = env.step(action.contiguous())
        else:
            raise ValueError(f"policy must be 'fused', 'hip' or 'torch', not {policy!r}")
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}
