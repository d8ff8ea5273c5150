#!/usr/bin/env python
"""Dump the four launch plans of a training step recorded with every collective of the N > 1 path issued by one rank
(LEOD_FORCE_COLLECTIVES=1): gpurun_out/plan_{bbf,hf,hb,bbb}_<segment>.txt -- lane, kernel, waits per op, host callbacks in place."""
import os, sys
os.environ.setdefault('LEOD_FORCE_COLLECTIVES', '1')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.argv = [sys.argv[0], '--steps', '3', '--warmup', '3', '--no-cpu-baseline', '--no-second-dtype', '--no-roofline']
import bench
from leod_amd.modules import step_plan
orig = step_plan.TrainStepPlans.info


def info(self):
    r = orig(self)
    bbs = [v for v in self.entries.values() if isinstance(v, step_plan.BackbonePlan)]
    if bbs:
        bb = bbs[-1]
        hd = [h for h in bb.heads.values() if isinstance(h, step_plan.HeadPlan)][-1]
        os.makedirs('gpurun_out', exist_ok=True)
        for name, rec in (('bbf', bb.fwd), ('hf', hd.fwd), ('hb', hd.bwd), ('bbb', bb.bwd)):
            for k, (pl, _cb) in enumerate(rec.segments):
                if pl is not None:
                    pl.dump(f'gpurun_out/plan_{name}_{k}.txt')
    return r


step_plan.TrainStepPlans.info = info
bench.main()
