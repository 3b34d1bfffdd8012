#!/usr/bin/env python
"""Ordered kernel timeline of ONE replay of the captured train step (developer tool): start offset, duration, gap to the previous kernel's end and
launch geometry of every kernel, from torch.profiler's roctracer records.

    python tools/step_timeline.py [workload] [--dtype bf16x3] [--batch 64]

Prints the sequence once (the median replay of 5 by total span) plus the sum of kernel time, of the gaps and the span of the replay."""
import argparse
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import bench                                                         # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="cfg2")
    ap.add_argument("--dtype", default="bf16x3")
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--frames", type=int, default=0)
    a = ap.parse_args()
    bench.set_dtype(a.dtype)
    from aesrc2020_amd import parallel
    parallel.init_from_env()                                            # SAR_DP_FORCE=1 (+ MASTER_ADDR / MASTER_PORT): the data-parallel step on one rank
    from aesrc2020_amd.model import SAR_Net
    from aesrc2020_amd.utils import synthetic_batch
    wl = bench.WORKLOADS[a.workload]
    T = a.frames or wl.get("frames") or (2000 if a.workload == "cfg5" else 400)
    B = a.batch or wl.get("batch") or 64
    kw = wl["kw"]
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        model, tm = SAR_Net([T, 80, 1], lr=0.01, gpus=1, seed=0, use_graph=True, **kw)
    xi, yi = synthetic_batch([T, 80, 1], B, kw.get("ctc_enable", False), kw.get("disc_enable", False), bpe_classes=kw.get("bpe_classes", 1000),
                             max_ctc_len=kw.get("max_ctc_len", 72), seed=1234)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in {**xi, **yi}.items()}
    for i in range(60):
        tm.set_device_batch(d); tm.train_step_device()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    N = 5
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for i in range(N):
                tm.set_device_batch(d); tm.train_step_device()
                torch.cuda.synchronize()
    ev = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    ev.sort(key=lambda e: e.time_range.start)
    # split into replays at the copy_ranges kernel (first launch of every step)
    starts = [i for i, e in enumerate(ev) if "copy_ranges" in e.name]
    reps = [ev[starts[i]:(starts[i + 1] if i + 1 < len(starts) else len(ev))] for i in range(len(starts))]
    reps = [r for r in reps if len(r) == max(len(x) for x in reps)]
    span = lambda r: r[-1].time_range.end - r[0].time_range.start
    reps.sort(key=span)
    r = reps[len(reps) // 2]
    t0 = r[0].time_range.start
    prev_end, ksum, gsum = t0, 0.0, 0.0
    for e in r:
        s, en = e.time_range.start, e.time_range.end
        name = e.name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0][:58]
        gap = s - prev_end
        print("%8.1f  %7.2f  gap %6.2f  %s" % (s - t0, en - s, gap, name))
        ksum += en - s; gsum += max(gap, 0.0); prev_end = max(prev_end, en)
    print("kernels %d, kernel time %.1f us, positive gaps %.1f us, span %.1f us" % (len(r), ksum, gsum, span(r)))


if __name__ == "__main__":
    main()
