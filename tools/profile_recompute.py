#!/usr/bin/env python
"""Recompute the roofline fraction of a profile snapshot from the rocprofv3 kernel trace alone, next to what bench.py printed in the same
run:   python tools/profile_recompute.py gpurun_out/<tag>   (tools/profile_round.sh calls it; the paragraph goes into profiles/README.md)

The traced command is `bench.py --steps 199 --warmup 0 --repeats R`: every k_rlepso_run launch of it is a timed launch (no warm-up launches
to subtract), R x 199 lock-step generations in R whole-episode launches (round 6; R x 4 launches of <= 50 generations before).
  fraction = 54 057 B x env-steps of all repeats / total duration of k_rlepso_run in the trace / 8e12 B/s."""
import csv, json, os, sys

d = sys.argv[1]
line = None
for l in open(os.path.join(d, 'trace.log'), errors='replace'):
    if l.startswith('{"metric"'):
        line = json.loads(l)
assert line is not None, 'no bench line in trace.log'
row = None
for r in csv.DictReader(open(os.path.join(d, 'kernel_stats.csv'))):
    if 'k_rlepso_run<256, 100, 10, 5, true>' in r['Name']:              # the exact-FDR (default) resident kernel
        row = r
assert row is not None, 'k_rlepso_run<256, 100, 10, 5, true> not in kernel_stats.csv'
calls, total_ns = int(row['Calls']), float(row['TotalDurationNs'])
R, K = int(line['repeats']), int(line['steps'])
env_steps = float(line['config']['live_env_steps']) * R                      # the reported repeat's live env-steps x repeats (episodes differ by < 1 %)
per_gen_us = total_ns / (R * K) / 1e3
frac = 54057.0 * env_steps / (total_ns * 1e-9) / 8e12
rl = line['roofline']
print(f"* recomputation from the trace (`{os.path.basename(d)}`): `k_rlepso_run<256, 100, 10, 5, true>` {calls} calls, {total_ns / 1e6:.3f} ms in total = {R} repeats x {K} generations "
      f"-> **{per_gen_us:.1f} us per generation** under the profiler (bench line of the same run: {rl['avg_generation_us']:.1f} us by HIP events); "
      f"54 057 B x {env_steps:.0f} env-steps / {total_ns / 1e6:.3f} ms / 8e12 B/s = **{frac:.4f}** (bench line of the same run: {rl['frac']:.4f}).")
