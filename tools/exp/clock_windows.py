"""What makes a window of the headline kernel run at 110 or at 127 us per generation?  Same 20 generations (6..25 of an episode, every instance live) of
k_rlepso_run<256,100,10,5> on 4096 instances, in modes: plain / marks (two one-thread kernels on the launch stream) / probe (a one-wave sampler on a side
stream) / both / idle_ms (a host sleep before the window) / back-to-back bursts.   python tools/exp/clock_windows.py"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from metabox_amd.suite import Batch, Suite
from metabox_amd._abi import ALGO_RLEPSO
from metabox_amd.problem.bbob import BBOB_Dataset

cfg = bench.make_config(); cfg.device = 'cuda'
agent = bench.load_agent(cfg, 'cuda'); actor = agent.actor; h1, h2 = actor.hidden_sizes()
tr, te = BBOB_Dataset.get_datasets('bbob', 10, 5.0)
ps = sorted(tr.data + te.data, key=lambda p: p.func_id)
B = 4096
b = Batch(Suite(ps), ALGO_RLEPSO, np.arange(B) % len(ps), np.arange(B, dtype=np.uint64) + 1000, 100, 20000, 400, 50)
table = b.policy_table(actor.packed_weights(), h1, h2, actor.min_sigma, actor.max_sigma)
lib = b.lib
dev = torch.device('cuda', 0)
side = torch.cuda.Stream()
main_s = torch.cuda.current_stream()


def window(marks=False, probe=False, idle_ms=0, K=20, W=5):
    buf = torch.zeros(2048, 2, dtype=torch.int64, device=dev); mk = torch.zeros(2, 2, dtype=torch.int64, device=dev)
    b.reset(); b.rlepso_rollout(table, W); torch.cuda.synchronize()
    if idle_ms:
        time.sleep(idle_ms * 1e-3)
    if probe:
        lib.mbx_debug_clock_probe(C.c_void_p(buf.data_ptr()), 2048, 127, C.c_void_p(side.cuda_stream))
    if marks:
        lib.mbx_debug_clock_mark(C.c_void_p(mk[0].data_ptr()), C.c_void_p(main_s.cuda_stream))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); b.rlepso_rollout(table, K); e1.record()
    if marks:
        lib.mbx_debug_clock_mark(C.c_void_p(mk[1].data_ptr()), C.c_void_p(main_s.cuda_stream))
    torch.cuda.synchronize()
    out = {'us_per_gen': round(e0.elapsed_time(e1) * 1e3 / K, 2)}
    if probe:
        bb = buf.cpu().numpy().astype(np.float64)
        ok = bb[:, 1] > 0
        per = np.diff(bb[ok, 0]) / np.diff(bb[ok, 1]) * 0.1
        out['probe_clock_ghz_all_samples_median_min_max'] = [round(float(np.median(per)), 3), round(float(per.min()), 3), round(float(per.max()), 3)]
        if marks:
            m = mk.cpu().numpy().astype(np.float64)
            ins = (bb[:, 1] >= m[0, 1]) & (bb[:, 1] <= m[1, 1])
            if ins.sum() > 4:
                out['clock_ghz_inside'] = round(float((bb[ins, 0][-1] - bb[ins, 0][0]) / (bb[ins, 1][-1] - bb[ins, 1][0]) * 0.1), 3)
    return out


for rep in range(2):
    for name, kw in (('plain', {}), ('plain', {}), ('marks', dict(marks=True)), ('probe', dict(probe=True)), ('both', dict(marks=True, probe=True)), ('plain', {}),
                     ('idle 50 ms, plain', dict(idle_ms=50)), ('idle 500 ms, plain', dict(idle_ms=500)), ('idle 500 ms, both', dict(idle_ms=500, marks=True, probe=True)),
                     ('plain', {}), ('plain K=100', dict(K=100)), ('both K=100', dict(K=100, marks=True, probe=True))):
        print(json.dumps({'mode': name, **window(**kw)}), flush=True)
# 30 plain windows back to back (the bench's repeats): per-window times
ts = [window()['us_per_gen'] for _ in range(30)]
print(json.dumps({'mode': '30 plain windows back to back', 'us_per_gen': ts}))
