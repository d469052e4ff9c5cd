import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, bench
def one(name, K=1500, W=100):
    leg = bench.Leg(name, 0, 0)
    sync = lambda: (leg.hp.synchronize(), torch.cuda.synchronize())
    el, pos, prof, *_ = bench.timed_run(leg, K, W, sync, 8, age_frames=200, export=False)
    n = max(prof["steps"], 1)
    print(f"{name:8s} step {el/K*1e6:7.1f} us  K1 {bench.k1_ms(prof)[0]*1e3:6.1f}  blob {prof['blob_ms']/n*1e3:6.1f} total {prof['total_ms']/n*1e3:6.1f}", flush=True)
    leg.close(); del leg; torch.cuda.empty_cache()
for name in sys.argv[1:]:
    one(name, 1500 if name.endswith("p1") else 300)
