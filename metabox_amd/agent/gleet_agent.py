"""GLEET agent: an attention policy that gives every particle of a PSO swarm its own exploration / exploitation balance
(reference: src/agent/gleet_agent.py, attention blocks from src/agent/networks.py:47-365).

Actor, per swarm of ps particles with 27 state values each (gleet_optimizer.py:111-124):
  h   = Embed_9->16(population features)                                      [ps, 16]
  h   = EncoderLayer(h)                 4-head self-attention + residual + swarm-wide normalisation, FF 16-16-16 + same
  q   = Embed_32->16([Embed_9->16(exploration memory) | Embed_9->16(exploitation memory)])
  z   = EncoderLayer(h, queries = q)    the particles' memories attend over the encoded swarm
  mu, sigma = two MLPs 16-32-8-1 (LeakyReLU) on z, squashed like RLEPSO's; action = clamp(N(mu, sigma), 0, 1) per particle.
Critic: MLP 16-32-16-1 on the swarm mean of z.  Training: PPO with n_step = 10, K_epochs = 3, clipped value loss, Adam 1e-4,
gradient-norm clipping at 0.1 per parameter group.

The modules keep the reference's parameter names, so a reference ``state_dict`` loads unchanged.  The forward already carries a
leading batch axis, so ``rollout_batch`` evaluates the policy for a whole lock-step batch [B, ps, 27] at once (chunked over B: the
attention maps are [4, B, ps, ps]) and feeds [B, ps] actions to the fused GLEET generation kernel.
"""
import math

import numpy as np
import torch
from torch import nn
from torch.distributions import Normal

from .basic_agent import Basic_Agent
from .networks import MLP
from .utils import save_class


def swarm_norm(x):
    """'layer' normalisation of the reference (networks.py:69-72): statistics over the whole [ps, E] block of a sample, unbiased
    variance, no affine parameters."""
    mean = x.mean((1, 2)).view(-1, 1, 1)
    return (x - mean) / torch.sqrt(x.var((1, 2)).view(-1, 1, 1) + 1e-05)


class _Embed(nn.Module):
    def __init__(self, n_in, n_out):
        super().__init__()
        self.embedder = nn.Linear(n_in, n_out, bias=False)

    def forward(self, x):
        return self.embedder(x)


class _Attention(nn.Module):
    """Multi-head attention with per-head projection tensors drawn from U[0, 1) (networks.py:113-190)."""

    def __init__(self, n_heads, dim):
        super().__init__()
        self.n_heads, self.dim, self.dk = n_heads, dim, dim // n_heads
        self.W_query = nn.Parameter(torch.rand(n_heads, dim, self.dk))
        self.W_key = nn.Parameter(torch.rand(n_heads, dim, self.dk))
        self.W_val = nn.Parameter(torch.rand(n_heads, dim, self.dk))
        self.W_out = nn.Parameter(torch.rand(n_heads, self.dk, dim))

    def forward(self, h, q=None):
        q = h if q is None else q
        # [H, B, n, dk] projections; scores [H, B, nq, n]
        Q = torch.einsum('bne,hek->hbnk', q, self.W_query)
        K = torch.einsum('bne,hek->hbnk', h, self.W_key)
        V = torch.einsum('bne,hek->hbnk', h, self.W_val)
        attn = torch.softmax((1 / math.sqrt(self.dk)) * torch.matmul(Q, K.transpose(2, 3)), dim=-1)
        heads = torch.matmul(attn, V)                                           # [H, B, nq, dk]
        return torch.einsum('hbnk,hke->bne', heads, self.W_out)


class _AttnBlock(nn.Module):
    def __init__(self, n_heads, dim):
        super().__init__()
        self.MHA = _Attention(n_heads, dim)

    def forward(self, x, q=None):
        return swarm_norm(self.MHA(x, q) + x)


class _FFBlock(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.FF = nn.Sequential(nn.Linear(dim, hidden), nn.ReLU(), nn.Linear(hidden, dim)) if hidden > 0 else nn.Linear(dim, dim)

    def forward(self, x):
        return swarm_norm(self.FF(x) + x)


class EncoderLayer(nn.Module):
    def __init__(self, n_heads, dim, hidden):
        super().__init__()
        self.MHA_sublayer = _AttnBlock(n_heads, dim)
        self.FFandNorm_sublayer = _FFBlock(dim, hidden)

    def forward(self, x, q=None):
        return self.FFandNorm_sublayer(self.MHA_sublayer(x, q))


def _head(dim, h1, h2):
    return MLP([{'in': dim, 'out': h1, 'drop_out': 0, 'activation': 'LeakyReLU'},
                {'in': h1, 'out': h2, 'drop_out': 0, 'activation': 'LeakyReLU'},
                {'in': h2, 'out': 1, 'drop_out': 0, 'activation': 'None'}])


class Actor(nn.Module):
    def __init__(self, embedding_dim=16, hidden_dim=16, n_heads=4, n_layers=1, node_dim=9, hidden_dim1=32, hidden_dim2=8,
                 max_sigma=0.7, min_sigma=0.01):
        super().__init__()
        self.node_dim = node_dim
        self.embedder = _Embed(node_dim, embedding_dim)
        self.encoder = nn.ModuleList([EncoderLayer(n_heads, embedding_dim, hidden_dim) for _ in range(n_layers)])
        self.embedder_for_decoder = _Embed(2 * embedding_dim, embedding_dim)
        self.decoder = nn.ModuleList([EncoderLayer(n_heads, embedding_dim, hidden_dim) for _ in range(n_layers)])
        self.mu_net = _head(embedding_dim, hidden_dim1, hidden_dim2)
        self.sigma_net = _head(embedding_dim, hidden_dim1, hidden_dim2)
        self.max_sigma, self.min_sigma = max_sigma, min_sigma

    _PACK_ORDER = ('embedder.embedder.weight',) + tuple(
        f'{blk}.0.{name}' for blk in ('encoder', 'decoder') for name in (
            'MHA_sublayer.MHA.W_query', 'MHA_sublayer.MHA.W_key', 'MHA_sublayer.MHA.W_val', 'MHA_sublayer.MHA.W_out',
            'FFandNorm_sublayer.FF.0.weight', 'FFandNorm_sublayer.FF.0.bias', 'FFandNorm_sublayer.FF.2.weight', 'FFandNorm_sublayer.FF.2.bias'))

    def packed_weights(self):
        """float32 CUDA tensor in the layout ``mbx_gleet_actor`` documents (include/mbx.h); re-packed after in-place updates."""
        ps = list(self.parameters())
        key = tuple((p.data_ptr(), p._version) for p in ps)
        if getattr(self, '_pw_key', None) != key:
            sd = self.state_dict()
            enc = [k for k in self._PACK_ORDER if k.startswith(('embedder.', 'encoder.'))]
            dec = [k for k in self._PACK_ORDER if k.startswith('decoder.')]
            heads = [f'{n}.net.layer{i}-linear.{w}' for n in ('mu_net', 'sigma_net') for i in range(3) for w in ('weight', 'bias')]
            order = enc + ['embedder_for_decoder.embedder.weight'] + dec + heads
            self._pw = torch.cat([sd[k].detach().reshape(-1) for k in order]).to(torch.float32).contiguous()
            self._pw_key = key
        return self._pw

    def features(self, x):
        """[B, ps, 27] -> decoder output z [B, ps, E] (what the critic consumes)."""
        n = self.node_dim
        h = self.embedder(x[:, :, :n])
        for layer in self.encoder:
            h = layer(h)
        q = self.embedder_for_decoder(torch.cat((self.embedder(x[:, :, n:2 * n]), self.embedder(x[:, :, 2 * n:])), dim=-1))
        for k, layer in enumerate(self.decoder):
            h = layer(h, q) if k == 0 else layer(h)          # mySequential hands the query to the first layer only
        return h

    def distribution(self, z):
        mu = (torch.tanh(self.mu_net(z)) + 1.) / 2.
        sigma = (torch.tanh(self.sigma_net(z)) + 1.) / 2. * (self.max_sigma - self.min_sigma) + self.min_sigma
        return mu, sigma

    def forward(self, x_in, fixed_action=None, require_entropy=False, to_critic=False, only_critic=False):
        z = self.features(x_in)
        if only_critic:
            return z
        mu, sigma = self.distribution(z)
        policy = Normal(mu, sigma)
        action = fixed_action if fixed_action is not None else torch.clamp(policy.sample(), min=0, max=1)
        log_prob = torch.sum(policy.log_prob(action), dim=1)          # joint action of the swarm
        out = (action, log_prob, z if to_critic else None)
        return out + (policy.entropy(),) if require_entropy else out


class Critic(nn.Module):
    def __init__(self, input_dim=16, hidden_dim1=32, hidden_dim2=16):
        super().__init__()
        self.value_head = MLP([{'in': input_dim, 'out': hidden_dim1, 'drop_out': 0, 'activation': 'LeakyReLU'},
                               {'in': hidden_dim1, 'out': hidden_dim2, 'drop_out': 0, 'activation': 'LeakyReLU'},
                               {'in': hidden_dim2, 'out': 1, 'drop_out': 0, 'activation': 'None'}])

    def forward(self, z):
        v = self.value_head(torch.mean(z, dim=-2))
        return v.detach().squeeze(), v.squeeze()


_HYPER = dict(embedding_dim=16, encoder_head_num=4, decoder_head_num=4, n_encode_layers=1, normalization='layer', v_range=6,
              hidden_dim=16, node_dim=9, hidden_dim1_actor=32, hidden_dim2_actor=8, max_sigma=0.7, min_sigma=0.01,
              hidden_dim1_critic=32, hidden_dim2_critic=16, gamma=0.999, n_step=10, K_epochs=3, eps_clip=0.1, lr_model=1e-4,
              lr_decay=0.9862327, max_grad_norm=0.1)


class GLEET_Agent(Basic_Agent):
    # Under torch.distributed the only training path Trainer drives is train_batch, whose gradients are synchronised over ranks: every rank holds the
    # same parameters, so rank 0 alone writes the checkpoints -- including the `checkpoint0` of __init__ / update_setting (agent/utils.save_class).
    # A class attribute: true before the first train_batch call and not part of the pickled instance state.
    _mbx_replicated = True

    def __init__(self, config):
        super().__init__(config)
        for k, v in _HYPER.items():                       # the agent publishes its hyper-parameters on the shared config (:31-53)
            setattr(config, k, v)
        self.__config = config
        self.actor = Actor(config.embedding_dim, config.hidden_dim, config.encoder_head_num, config.n_encode_layers, config.node_dim,
                           config.hidden_dim1_actor, config.hidden_dim2_actor, config.max_sigma, config.min_sigma).to(config.device)
        self.critic = Critic(config.embedding_dim, config.hidden_dim1_critic, config.hidden_dim2_critic).to(config.device)
        self.optimizer = torch.optim.Adam([{'params': self.actor.parameters(), 'lr': config.lr_model},
                                           {'params': self.critic.parameters(), 'lr': config.lr_model}])
        self.lr_scheduler = torch.optim.lr_scheduler.ExponentialLR(self.optimizer, config.lr_decay, last_epoch=-1)
        self.__learning_time = 0
        self.__cur_checkpoint = 0
        if getattr(config, 'agent_save_dir', None):
            save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    def load_exported_weights(self, npz):
        """Arrays exported by tools/gen_golden.py (`gleet_policy` section): keys 'actor/<name>' and 'critic/<name>'."""
        keys = npz.files if hasattr(npz, 'files') else list(npz)
        self.actor.load_state_dict({k[6:]: torch.as_tensor(np.asarray(npz[k])) for k in keys if k.startswith('actor/')})
        self.critic.load_state_dict({k[7:]: torch.as_tensor(np.asarray(npz[k])) for k in keys if k.startswith('critic/')})
        return self

    def to(self, device):
        self.__config.device = device
        self.actor.to(device)
        self.critic.to(device)
        return self

    def update_setting(self, config):
        self.__config.max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__learning_time = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    def __after_update(self):
        c = self.__config
        self.__learning_time += 1
        if self.__learning_time >= c.save_interval * self.__cur_checkpoint:
            save_class(c.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
            self.__cur_checkpoint += 1
        return self.__learning_time >= c.max_learning_step

    # ---- reference protocol (one environment) ------------------------------------------------------------------------
    def train_episode(self, env):
        """PPO over one environment (gleet_agent.py:113-290): segments of n_step generations, K_epochs passes per segment (the first
        pass re-uses the rollout's own log-probabilities and values), bootstrapped n-step returns, clipped surrogate and clipped
        value loss."""
        c = self.__config
        dev = c.device
        state = torch.FloatTensor(env.reset()[None, :]).to(dev)
        ret, done = 0., False
        info = lambda: {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1], 'return': ret,
                        'learn_steps': self.__learning_time}
        while not done:
            states, actions, logps, values, rewards = [], [], [], [], []
            while len(states) < c.n_step:
                states.append(state.clone())
                action, logp, z, _ = self.actor(state, require_entropy=True, to_critic=True)
                actions.append(action.clone())
                logps.append(logp)
                values.append(self.critic(z))
                state, reward, done = env.step(action.detach().cpu().numpy().squeeze())
                rewards.append(torch.FloatTensor([reward]).to(dev))
                ret += float(np.squeeze(reward))
                state = torch.FloatTensor(state[None, :]).to(dev)
                if done:
                    break
            T = len(states)
            old_logp = torch.stack(logps).detach().view(-1)
            old_value = None
            for k in range(c.K_epochs):
                if k > 0:                                  # re-evaluate the stored actions under the current parameters
                    logps, values = [], []
                    for t in range(T):
                        _, logp, z, _ = self.actor(states[t], fixed_action=actions[t], require_entropy=True, to_critic=True)
                        logps.append(logp)
                        values.append(self.critic(z))
                logp = torch.stack(logps).view(-1)
                v_det = torch.stack([v[0] for v in values]).view(-1)
                v = torch.stack([v[1] for v in values]).view(-1)
                R = self.critic(self.actor(state, only_critic=True))[0]
                targets = []
                for r in rewards[::-1]:
                    R = R * c.gamma + r
                    targets.append(R)
                target = torch.stack(targets[::-1], 0).view(-1)
                ratio = torch.exp(logp - old_logp)
                adv = target - v_det
                policy_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1 - c.eps_clip, 1 + c.eps_clip) * adv).mean()
                if old_value is None:
                    value_loss = ((v - target) ** 2).mean()
                    old_value = v.detach()
                else:
                    v_clip = old_value + torch.clamp(v - old_value, -c.eps_clip, c.eps_clip)
                    value_loss = torch.max((v - target) ** 2, (v_clip - target) ** 2).mean()
                self.optimizer.zero_grad()
                (value_loss + policy_loss).backward()
                for group in self.optimizer.param_groups:
                    torch.nn.utils.clip_grad_norm_(group['params'], c.max_grad_norm if c.max_grad_norm > 0 else math.inf, norm_type=2)
                self.optimizer.step()
                if self.__after_update():
                    return True, info()
        return self.__learning_time >= c.max_learning_step, info()

    # ---- batched training (SURVEY §8 N3 applied to GLEET) -----------------------------------------------------------------
    def train_batch(self, env, max_updates=None):
        """PPO over a lock-step BatchedPBO_Env: the reference's n_step = 10 segments / K_epochs = 3 / clipped surrogate + clipped value
        loss / per-group gradient-norm clipping (gleet_agent.py:113-290) with a leading batch axis; losses are averaged over the
        (step, swarm) pairs still running.  By construction one optimizer step consumes B trajectories instead of one.  Gradients are
        averaged across ranks when torch.distributed is initialised.
        Returns (exceed_max_learning_step, {'normalizer', 'gbest', 'return', 'learn_steps'}) with per-batch means."""
        from ..distributed import all_ranks_any, average_gradients
        c = self.__config
        actor, critic = self.actor, self.critic
        params = list(actor.parameters()) + list(critic.parameters())
        B = env.B

        def as_input(st):
            return st.view(B, -1, 27).to(torch.float32).clone()

        def evaluate(x, action):
            z = actor.features(x)
            mu, sigma = actor.distribution(z)
            return Normal(mu, sigma).log_prob(action).sum(dim=(1, 2)), critic.value_head(z.mean(dim=-2)).squeeze(-1)

        state = as_input(env.reset())
        alive = torch.ones(B, dtype=torch.bool, device=state.device)
        ret_sum = torch.zeros(B, dtype=torch.float64, device=state.device)
        updates, exceed = 0, False
        while all_ranks_any(bool(alive.any()), state.device) and not exceed:   # global loop control: every rank issues the same collectives
            S, A, LP, V, R, M = [], [], [], [], [], []
            for _ in range(c.n_step):
                with torch.no_grad():
                    mu, sigma = actor.distribution(actor.features(state))
                    action = torch.clamp(mu + sigma * torch.randn_like(mu), 0, 1)
                logp, val = evaluate(state, action)
                S.append(state); A.append(action); LP.append(logp); V.append(val); M.append(alive.clone())
                nstate, reward, done = env.step(action.squeeze(-1).contiguous())
                R.append(reward.to(torch.float32).clone())
                ret_sum += reward * alive
                alive = alive & (done == 0)
                state = as_input(nstate)
                if not bool(alive.any()):
                    break
            T = len(S)
            M = torch.stack(M).to(torch.float32)
            R = torch.stack(R) * M
            old_logp = torch.stack(LP).detach()
            n_live = M.sum().clamp_min(1.)
            old_value = None
            for k in range(c.K_epochs):
                if k == 0:
                    logp, val = torch.stack(LP), torch.stack(V)
                else:
                    pairs = [evaluate(S[t], A[t]) for t in range(T)]
                    logp, val = torch.stack([q[0] for q in pairs]), torch.stack([q[1] for q in pairs])
                with torch.no_grad():
                    Rt = critic.value_head(actor.features(state).mean(dim=-2)).squeeze(-1)
                    returns = []
                    for t in reversed(range(T)):
                        Rt = torch.where(M[t] > 0, Rt * c.gamma + R[t], Rt)
                        returns.append(Rt)
                    returns = torch.stack(returns[::-1])
                ratio = torch.exp(logp - old_logp)
                adv = returns - val.detach()
                policy_loss = -(torch.min(ratio * adv, torch.clamp(ratio, 1 - c.eps_clip, 1 + c.eps_clip) * adv) * M).sum() / n_live
                if old_value is None:
                    value_loss = (((val - returns) ** 2) * M).sum() / n_live
                    old_value = val.detach()
                else:
                    v_clip = old_value + torch.clamp(val - old_value, -c.eps_clip, c.eps_clip)
                    value_loss = (torch.max((val - returns) ** 2, (v_clip - returns) ** 2) * M).sum() / n_live
                self.optimizer.zero_grad()
                (value_loss + policy_loss).backward()
                average_gradients(params, weight=M.sum())
                for group in self.optimizer.param_groups:
                    torch.nn.utils.clip_grad_norm_(group['params'], c.max_grad_norm if c.max_grad_norm > 0 else math.inf, norm_type=2)
                self.optimizer.step()
                updates += 1
                if self.__after_update() or (max_updates is not None and updates >= max_updates):
                    exceed = True
                    break
        res = env.results()
        return self.__learning_time >= c.max_learning_step, {
            'normalizer': float(res['cost'][:, 0].mean()), 'gbest': float(res['cost'][:, -1].mean()),
            'return': float(ret_sum.mean()), 'learn_steps': self.__learning_time}

    @torch.no_grad()
    def rollout_episode(self, env):
        done, ret = False, 0
        state = env.reset()
        while not done:
            x = torch.FloatTensor(state[None, :]).to(self.__config.device)
            action = self.actor(x)[0]
            state, reward, done = env.step(action.cpu().numpy().squeeze())
            ret += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': ret}

    # ---- lock-step batch -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def act_batch(self, state, chunk=1024):
        """[B, ps * 27] float64 device state -> [B, ps] float32 actions; chunked over B to bound the [4, B, ps, ps] attention maps."""
        B = state.shape[0]
        x = state.view(B, -1, 27).to(torch.float32)
        out = torch.empty(B, x.shape[1], dtype=torch.float32, device=state.device)
        for a in range(0, B, chunk):
            mu, sigma = self.actor.distribution(self.actor.features(x[a:a + chunk]))
            out[a:a + chunk] = torch.addcmul(mu, sigma, torch.randn_like(mu)).clamp_(0, 1).squeeze(-1)
        return out

    def rollout_batch(self, env, max_steps=None, policy='hip'):
        """Lock-step rollout of a BatchedPBO_Env, two launches per generation: policy = 'hip' (default) evaluates the attention actor
        with ``mbx_gleet_policy`` (one workgroup per swarm), 'torch' with the PyTorch modules (``act_batch``); then the fused
        generation kernel.  Both sample the same distribution; 'torch' uses torch's generator instead of the instance's Philox stream."""
        bc = env.batch.cfg
        if max_steps is None:
            max_steps = -(-(bc.max_fes - bc.np) // bc.np)
        state = env.reset()
        actor = self.actor
        for _ in range(max_steps):
            if policy == 'hip':
                actions = env.batch.gleet_policy(actor.packed_weights(), actor.min_sigma, actor.max_sigma)
            else:
                actions = self.act_batch(state)
            state, _, _ = env.step(actions)
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}
