"""Does torch still see the GPU after liblapwarm_hip.so initialised HIP in this process?"""
import sys
sys.path.insert(0, "gnn-accelerated-lap-warm-start-pipeline_amd")
import numpy as np


def maps():
    seen = set()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line or "libhsa-runtime" in line:
            seen.add(line.split()[-1])
    return sorted(seen)


mode = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
if mode == "lib_first":
    import lap
    C = np.random.RandomState(0).uniform(size=(64, 64))
    print("lapjv", lap.lapjv(C)[0])
    if len(sys.argv) > 2:
        from gnn import compute_row_features
        print("feat", compute_row_features(C).shape)
    print(maps())
    import torch
    print("torch sees GPU:", torch.cuda.is_available(), torch.cuda.device_count())
    print(maps())
else:
    import torch
    print("torch sees GPU:", torch.cuda.is_available())
    import lap
    print("lapjv", lap.lapjv(np.random.RandomState(0).uniform(size=(64, 64)))[0])
    print(maps())
