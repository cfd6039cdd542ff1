"""Host-to-device copy rate of a pinned 12.5 MB buffer allocated (first touched) from each NUMA node:
python tools/micro/h2d_numa.py"""
import glob, os, time
import torch

def cpus_of(path):
    out = []
    for part in open(path).read().strip().split(","):
        if not part: continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out

dev = torch.device("cuda:0")
p = torch.cuda.get_device_properties(0)
bus = None
try:
    bus = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
    print("GPU", bus, "numa_node", open("/sys/bus/pci/devices/%s/numa_node" % bus).read().strip())
except Exception as e:
    print("no pci info:", e)
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes", len(nodes), "affinity now", len(os.sched_getaffinity(0)), "cpus")
allowed = os.sched_getaffinity(0)
n = 12_500_000 // 8
d = torch.zeros(n, dtype=torch.float64, device=dev)
for nd in nodes:
    cpus = set(cpus_of(nd + "/cpulist")) & allowed
    if not cpus:
        print(os.path.basename(nd), "no allowed cpus"); continue
    os.sched_setaffinity(0, cpus)
    h = torch.empty(n, dtype=torch.float64).pin_memory()
    h.fill_(1.0)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(8):
        t0 = time.perf_counter()
        d.copy_(h, non_blocking=True)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(os.path.basename(nd), "cpus", len(cpus), "H2D %.1f GB/s (%.0f us)" % (n * 8 / best / 1e9, best * 1e6))
    del h
os.sched_setaffinity(0, allowed)
