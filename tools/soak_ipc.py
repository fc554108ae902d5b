"""Soak: 3 ranks (C2 512^2) and 2 ranks (C4 slab pair) on one GPU through the IPC transport, many device-resident steps, against the single engine."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["DFLO_RANK_TRANSPORT"] = "ipc"
import torch.multiprocessing as mp
import test_gpu_multi_large as L
L.RESIDENT = int(sys.argv[1]) if len(sys.argv) > 1 else 300

def worker(rank, world, port, name, ret):
    import test_gpu_multi_large as T
    T.RESIDENT = L.RESIDENT
    T._worker(rank, world, port, name, ret, "ipc")

if __name__ == "__main__":
    for name, world, reps in (("c2", 3, 3), ("c2", 2, 2), ("c4", 2, 2)):
        for r in range(reps):
            mgr = mp.get_context("spawn").Manager()
            ret = mgr.dict()
            mp.spawn(worker, args=(world, 33000 + random.randint(0, 2000), name, ret), nprocs=world, join=True)
            print(name, world, "rep", r, dict(ret), flush=True)
