#!/usr/bin/env python
"""Whole-step hipGraph replay vs eager launch of the RVT-S training step (TrainEngine, precision mode bf16 unless LEOD_PRECISION says otherwise).
usage: python tools/graph_ab.py [eager|eager1|graph|graph_side|graph_head|graph_both] ...   (each variant runs in its own process)"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (the graph_side / graph_head / plan_both variants of profiles/r04_a_graph_ab.txt needed the LEOD_GRAPH_SIDE / LEOD_GRAPH_HEAD_STREAMS
# experiment switches, removed after the measurement)
VARIANTS = {'eager': {}, 'eager1': {'LEOD_WGRAD_STREAM': '0', 'LEOD_HEAD_STREAMS': '0'}, 'graph': {},
            'plan1': {'LEOD_PLAN_LANES': '1'}, 'plan_side': {}}


def child(variant):
    import torch
    import bench
    from leod_amd import ops
    from leod_amd.config import full_config, dynamically_modify_train_config
    from leod_amd.models.detection.yolox_extension.models.detector import YoloXDetector
    from leod_amd.engine import TrainEngine
    ops.set_precision(os.environ.get('LEOD_PRECISION', 'bf16'))
    dev = torch.device('cuda', 0)
    cfg = dynamically_modify_train_config(full_config('gen1', 'small'))
    torch.manual_seed(0)
    eng = TrainEngine(YoloXDetector(cfg.model).to(dev), lr=cfg.training.learning_rate)
    T, B = 21, 8
    ev, labels, label_tb, _ = bench.make_batch(T, B, (240, 304), 2, 0, dev, (4, 9, 14, 19))
    first = torch.zeros(B, dtype=torch.bool, device=dev)
    graph = variant.startswith('graph') or variant.startswith('plan')
    plan = variant.startswith('plan')
    for _ in range(3):
        eng.step(ev, labels, label_tb, first)
    if graph:
        t0 = time.perf_counter()
        if plan and variant == 'plan1':
            eng.wgrad_side = False
        eng.capture(ev, labels, label_tb, first, plan=plan, max_lanes=int(os.environ.get('LEOD_PLAN_LANES', '8')))
        print(f'{variant}: capture {time.perf_counter() - t0:.2f} s' + (f' plan {eng._plan.info}' if plan else ''), flush=True)
        if plan and os.environ.get('LEOD_PLAN_DUMP'):
            eng._plan.dump(os.environ['LEOD_PLAN_DUMP'] + '_' + variant + '.txt')
        run = lambda: eng.step_graph(ev, labels, first)
    else:
        run = lambda: eng.step(ev, labels, label_tb, first)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    n = 40
    enq = 0.0
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        run()
        enq += time.perf_counter() - a
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'{variant}: {1e3 * dt / n:.2f} ms/step, host enqueue {1e3 * enq / n:.2f} ms/step, loss {float(eng.last_losses["loss"]):.4f}', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 2 and sys.argv[1] == '--child':
        child(sys.argv[2])
    else:
        for v in (sys.argv[1:] or list(VARIANTS)):
            env = dict(os.environ, **VARIANTS[v])
            try:
                r = subprocess.run([sys.executable, __file__, '--child', v], env=env, capture_output=True, text=True, timeout=420)
                out = [l for l in r.stdout.splitlines() if l.startswith(v)]
                print('\n'.join(out) if out else f'{v}: rc {r.returncode}: {r.stderr[-600:]}', flush=True)
            except subprocess.TimeoutExpired:
                print(f'{v}: timeout', flush=True)
