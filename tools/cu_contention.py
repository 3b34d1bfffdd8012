#!/usr/bin/env python
"""What the data-parallel step loses when a collective's kernels sit on the same GPU (developer tool; VERDICT r5 "Next" item 4a).

    hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/cu_hog.hip -o tools/libcu_hog.so && python tools/cu_contention.py [cfg3|cfg2]

No N > 1 box was available to any round: the all-reduce of the 33.6 MB gradient range that the step overlaps with the backward of stages 1-0 and
the stem (model.train_step_device) has never run beside it over real links.  What one GPU can tell: how much slower that backward
segment -- the graph the collective hides under -- gets when k CUs are not its own.  The step is built in its forced one-rank three-graph form
(SAR_DP_FORCE=1, SAR_DP_SEGMENTS=3) so that the segment is a graph of its own; a hog kernel (tools/cu_hog.hip) is started on a second stream, the
segment is replayed while it runs, and the replay is timed with events on its own stream.  Two hogs:
  exclusive  k workgroups that take a whole CU each (all of its LDS): the CU is gone, as under a CU mask -- the upper bound of what k RCCL channels cost;
  channel    k workgroups of 512 threads / 32 KB of LDS that stream a buffer (load, add, store) for the whole time: a ring channel's footprint without
             the links -- they share their CU with the step's blocks and take memory bandwidth.
Prints the segment's time per k and the penalty against k = 0; DESIGN section 6 quotes the table and the channel budget it implies."""
import argparse
import ctypes
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
os.environ.update(SAR_DP_FORCE="1", SAR_DP_SEGMENTS="3", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")

import torch  # noqa: E402

import bench  # noqa: E402
from aesrc2020_amd import parallel  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
hog = ctypes.CDLL(os.path.join(HERE, "libcu_hog.so"))
vp = ctypes.c_void_p
hog.hog_exclusive.argtypes = [ctypes.c_int, ctypes.c_uint, vp, vp]
hog.hog_channel.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_long, ctypes.c_uint, vp, vp]

rank, world, local = parallel.init_from_env()
torch.cuda.set_device(local)
args = argparse.Namespace(steps=5, warmup=3, regions=1, settle=20, batch=None, frames=None, no_graph=False, gpus=1, workload=wl, dtype="bf16x3")
bench.set_dtype("bf16x3")
r, model = bench.run_workload(wl, args, rank, world, False)
assert isinstance(model.graph, dict) and "trunk_s10" in model.graph, "the three-graph data-parallel step was not built: %r" % (model.graph,)
seg = model.graph["trunk_s10"]
side = torch.cuda.Stream()
src = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda"); dst = torch.empty_like(src); sink = torch.zeros(4, dtype=torch.int32, device="cuda")


def timed(launch_hog, reps=15):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if launch_hog is not None:
            launch_hog()
            time.sleep(0.0003)                                   # the hog's workgroups are resident before the segment's first launch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); seg.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts))


for _ in range(5):
    seg.replay()
t0 = timed(None)
dur = int(t0 * 1.6 + 600)                                        # microseconds the hog stays: the host sleep + the whole segment
print("%s, forced one-rank three-graph step (%.3f ms per step): backward segment 'stages 1-0 + stem' alone %.1f us" % (wl, r["ms_per_step"], t0))
print("%-10s %4s %12s %10s" % ("hog", "k", "segment us", "penalty"))
for kind in ("exclusive", "channel"):
    for k in (8, 16, 32, 64):
        if kind == "exclusive":
            fn = lambda k=k: hog.hog_exclusive(k, dur, sink.data_ptr(), side.cuda_stream)
        else:
            fn = lambda k=k: hog.hog_channel(k, 512, 32 * 1024, src.data_ptr(), dst.data_ptr(), src.numel() // 16, dur, sink.data_ptr(), side.cuda_stream)
        t = timed(fn)
        print("%-10s %4d %12.1f %+9.1f%%" % (kind, k, t, 100.0 * (t / t0 - 1.0)))
torch.cuda.synchronize()
import torch.distributed as dist  # noqa: E402
if dist.is_initialized():
    dist.destroy_process_group()
