"""DE-DDQN agent: double deep Q-network choosing one of four DE mutation operators per trial vector
(reference: src/agent/de_ddqn_agent.py).  Q-network = MLP 99 -> 100 x4 (ReLU) -> 4; rollout is greedy,
training is epsilon-greedy with a 1e5 replay buffer, 1e4 warm-up, batch 64, target sync every 1000 updates.

``rollout_batch`` evaluates the Q-network once per step over the whole instance batch (PyTorch-ROCm GEMMs) and
feeds the argmax actions to the fused DE-DDQN step kernel.
"""
import copy

import numpy as np
import torch

from .basic_agent import Basic_Agent
from .networks import MLP
from .utils import ReplayBuffer, save_class


class DE_DDQN_Agent(Basic_Agent):
    def __init__(self, config):
        super().__init__(config)
        config.state_size = 99
        config.n_act = 4
        config.lr = 1e-4
        config.batch_size = 64
        config.epsilon = 0.1
        config.gamma = 0.99
        config.update_target_steps = 1000
        config.memory_size = 100000
        config.warm_up_size = 10000
        config.net_config = [{'in': config.state_size, 'out': 100, 'drop_out': 0, 'activation': 'ReLU'},
                             {'in': 100, 'out': 100, 'drop_out': 0, 'activation': 'ReLU'},
                             {'in': 100, 'out': 100, 'drop_out': 0, 'activation': 'ReLU'},
                             {'in': 100, 'out': 100, 'drop_out': 0, 'activation': 'ReLU'},
                             {'in': 100, 'out': config.n_act, 'drop_out': 0, 'activation': 'None'}]
        self.__config = config
        self.__device = config.device
        self.__pred_func = MLP(config.net_config).to(self.__device)
        self.__target_func = copy.deepcopy(self.__pred_func).to(self.__device)
        self.__optimizer = torch.optim.AdamW(self.__pred_func.parameters(), lr=config.lr)
        self.__criterion = torch.nn.MSELoss()
        self.__n_act = config.n_act
        self.__epsilon = config.epsilon
        self.__gamma = config.gamma
        self.__update_target_steps = config.update_target_steps
        self.__batch_size = config.batch_size
        self.__replay_buffer = ReplayBuffer(config.memory_size)
        self.__warm_up_size = config.warm_up_size
        self.__max_learning_step = config.max_learning_step
        self.__global_ls = 0
        self.__cur_checkpoint = 0
        if getattr(config, 'agent_save_dir', None):
            save_class(config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
        self.__cur_checkpoint += 1

    @property
    def q_net(self):
        return self.__pred_func

    def load_exported_weights(self, npz):
        sd = {k[len('net/'):]: torch.as_tensor(np.asarray(npz[k])) for k in npz.files if k.startswith('net/')}
        self.__pred_func.load_state_dict(sd)
        self.__target_func = copy.deepcopy(self.__pred_func)
        return self

    def to(self, device):
        self.__device = device
        self.__config.device = device
        self.__pred_func.to(device)
        self.__target_func.to(device)
        return self

    def update_setting(self, config):
        self.__max_learning_step = config.max_learning_step
        self.__config.agent_save_dir = config.agent_save_dir
        self.__global_ls = 0
        save_class(self.__config.agent_save_dir, 'checkpoint0', self)
        self.__config.save_interval = config.save_interval
        self.__cur_checkpoint = 1

    def __get_action(self, state, options=None):
        state = torch.Tensor(state).to(self.__device)
        action = None
        with torch.no_grad():
            Q_list = self.__pred_func(state)
        if options['epsilon_greedy'] and np.random.rand() < self.__epsilon:
            action = np.random.randint(low=0, high=self.__n_act)
        if action is None:
            action = int(torch.argmax(Q_list).detach().cpu().numpy())
        return action, Q_list[action].detach().cpu().numpy()

    def rollout_episode(self, env):
        state = env.reset()
        done, R = False, 0
        while not done:
            action, _ = self.__get_action(state, {'epsilon_greedy': False})
            state, reward, done = env.step(action)
            R += reward
        return {'cost': env.optimizer.cost, 'fes': env.optimizer.fes, 'return': R}

    @torch.no_grad()
    def rollout_batch(self, env, max_steps=None):
        if max_steps is None:
            bc = env.batch.cfg
            max_steps = bc.max_fes - bc.np                # one evaluation per step
        state = env.reset()
        for _ in range(max_steps):
            actions = torch.argmax(self.__pred_func(state.to(torch.float32)), dim=1).to(torch.int32)
            state, _, _ = env.step(actions.contiguous())
        res = env.results()
        return {'cost': res['cost'], 'fes': res['fes'], 'return': res['return'], 'steps': res['steps'], 'cost_len': res['cost_len']}

    def train_episode(self, env):
        """Reference training loop (de_ddqn_agent.py:70-106)."""
        state = env.reset()
        done, R = False, 0
        while not done:
            action, _ = self.__get_action(state, {'epsilon_greedy': True})
            next_state, reward, done = env.step(action)
            R += reward
            self.__replay_buffer.append((state, action, reward, next_state, done))
            if len(self.__replay_buffer) >= self.__warm_up_size:
                obs, act, rew, nxt, dn = self.__replay_buffer.sample(self.__batch_size)
                pred_Vs = self.__pred_func(obs.to(self.__device))
                onehot = torch.nn.functional.one_hot(act.to(self.__device), self.__n_act)
                predict_Q = (pred_Vs * onehot).sum(1)
                target_Q = rew.to(self.__device) + (1 - dn.to(self.__device)) * self.__gamma * \
                    self.__target_func(nxt.to(self.__device)).max(1)[0]
                self.__optimizer.zero_grad()
                loss = self.__criterion(predict_Q, target_Q.detach())
                loss.backward()
                self.__optimizer.step()
                self.__global_ls += 1
                if self.__global_ls >= (self.__config.save_interval * self.__cur_checkpoint):
                    save_class(self.__config.agent_save_dir, 'checkpoint' + str(self.__cur_checkpoint), self)
                    self.__cur_checkpoint += 1
                if self.__global_ls >= self.__max_learning_step:
                    break
            if self.__global_ls % self.__update_target_steps == 0:
                for tp, pp in zip(self.__target_func.parameters(), self.__pred_func.parameters()):
                    tp.data.copy_(pp.data)
            state = next_state
        return self.__global_ls >= self.__max_learning_step, {'normalizer': env.optimizer.cost[0], 'gbest': env.optimizer.cost[-1],
                                                              'return': R, 'learn_steps': self.__global_ls}
