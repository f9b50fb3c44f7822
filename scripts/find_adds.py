"""Which autograd nodes issue the framework's elementwise adds in an eager attention step (torch profiler, with stacks)."""
import subprocess, sys, os, json, tempfile
import torch
from torch.profiler import profile, ProfilerActivity

sys.argv = ["bench.py", "--combine", "attention", "--no-graph", "--steps", "2", "--warmup", "2", "--no-cpu-baseline", "--no-roofline"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    bench.main()
seen = {}
for e in prof.events():
    if e.name in ("aten::add", "aten::add_") and e.input_shapes and len(e.input_shapes[0]) == 2 and e.input_shapes[0][0] > 10000:
        key = (e.name, str(e.input_shapes[:2]), tuple(str(s) for s in (e.stack or [])[:6]))
        seen[key] = seen.get(key, 0) + 1
for (name, shp, st), c in sorted(seen.items(), key=lambda kv: -kv[1]):
    print(c, name, shp)
    for s in st:
        print("     ", s)
