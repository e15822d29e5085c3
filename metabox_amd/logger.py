"""Result aggregation metrics (reference: src/logger.py): the Random_search baseline (:94-120), AEI (:574-645) and the
CEC-style score (:83-91, 647-680).  Presentation (xlsx tables, matplotlib figures) is out of scope.

The inputs are the ``test.pkl`` / ``random_search_baseline.pkl`` dictionaries written by ``Tester``:
``{'cost': {problem: {agent: [runs][n_logpoint+1]}}, 'fes': {problem: {agent: [runs]}}, 'T0': ms, 'T1': {agent: ms},
'T2': {agent: ms}}``.
"""
import numpy as np


def to_label(agent_name):
    label = agent_name
    if len(label) > 6 and label[-6:] in ('_Agent', '_agent'):
        label = label[:-6]
    return label


def get_random_baseline(random, fes):
    """Statistics of the Random_search runs that normalise AEI.  Note the operator precedence kept from the reference:
    ``log10(1 / (T2 - T1) / T0)`` = -log10((T2 - T1) * T0)."""
    base = {}
    t1 = random['T1']['Random_search'] if isinstance(random['T1'], dict) else random['T1']
    base['complexity_avg'] = np.log10(1 / (random['T2']['Random_search'] - t1) / random['T0'])
    base['complexity_std'] = 0.005
    problems = random['cost'].keys()
    g = [np.log10(fes / np.array(random['fes'][p]['Random_search'])) for p in problems]
    base['fes_avg'] = np.mean([x.mean() for x in g])
    base['fes_std'] = np.mean([x.std() for x in g])
    g = [np.log10(1 / (np.array(random['cost'][p]['Random_search'])[:, -1] + 1)) for p in problems]
    base['cost_avg'] = np.mean([x.mean() for x in g])
    base['cost_std'] = np.mean([x.std() for x in g])
    return base


def cal_scores1(D, maxf):
    sne = np.array([0.5 * np.sum(np.min(D[a], -1) / maxf) for a in D.keys()])
    return (1 - (sne - np.min(sne)) / sne) * 50


class Logger:
    def __init__(self, config):
        self.config = config

    def aei_metric(self, data, random, maxFEs=20000, ignore=None):
        """Aggregated Evaluation Indicator: mean over problems of Z_complexity * Z_cost[p] * Z_fes[p] (exp-normalised
        against Random_search); returns ({agent: mean}, {agent: std})."""
        base = get_random_baseline(random, maxFEs)
        problems = list(data['fes'].keys())
        if 'complexity' not in data:
            data['complexity'] = {}
            agents = list(data['fes'][problems[0]].keys())
        else:
            agents = list(data['complexity'].keys())
        z_complex, z_fes, z_cost = {}, {}, {}
        for key in agents:
            if ignore is not None and key in ignore:
                continue
            if key not in data['complexity']:
                t1 = data['T1'][key] if isinstance(data['T1'], dict) else data['T1']
                data['complexity'][key] = (data['T2'][key] - t1) / data['T0']
            z_complex[key] = np.exp((np.log10(1 / data['complexity'][key]) - base['complexity_avg']) / base['complexity_std'] / 1000)
        for agent in agents:
            budget = 100 if agent == 'L2L_Agent' else (self.config.bo_maxFEs if agent == 'BayesianOptimizer' else maxFEs)
            f = [np.log10(budget / np.array(data['fes'][p][agent])).mean() for p in problems]
            z_fes[agent] = np.exp(np.array(f) - base['fes_avg'])
            c = [np.log10(1 / (np.array(data['cost'][p][agent])[:, -1] + 1)).mean() for p in problems]
            z_cost[agent] = np.exp(np.array(c) - base['cost_avg'])
        mean, std = {}, {}
        for agent in agents:
            if (ignore is not None and agent in ignore) or agent == 'Random_search':
                continue
            aei = z_complex[agent] * z_cost[agent] * z_fes[agent]
            mean[agent] = np.mean(aei)
            std[agent] = np.std(aei) * 5. if self.config.problem in ['protein', 'protein-torch'] else np.std(aei) / 5.
        return mean, std

    def cec_metric(self, data, ignore=None):
        score, M, R = {}, [], []
        cost, fes = data['cost'], data['fes']
        for problem in list(cost.keys()):
            maxf, avg_cost, avg_fes = 0, [], []
            for agent in list(cost[problem].keys()):
                if ignore is not None and agent in ignore:
                    continue
                key = to_label(agent)
                values = np.array(cost[problem][agent])[:, -1]
                score.setdefault(key, []).append(values)
                maxf = max(maxf, np.max(values))
                avg_cost.append(np.mean(values))
                avg_fes.append(np.mean(fes[problem][agent]))
            M.append(maxf)
            order = np.lexsort((avg_fes, avg_cost))
            rank = np.zeros(len(avg_cost))
            rank[order] = np.arange(len(avg_cost)) + 1
            R.append(rank)
        sr = 0.5 * np.sum(R, 0)
        score2 = (1 - (sr - np.min(sr)) / sr) * 50
        score1 = cal_scores1(score, M)
        for i, key in enumerate(score.keys()):
            score[key] = score1[i] + score2[i]
        return score
