"""Developer probe: can two ranks share one GPU under RCCL (for rehearsing world_size 2 on a 1-GPU box)?"""
import sys, os, time, multiprocessing as mp
sys.path.insert(0, ".")
def worker(rank, world, path):
    import safebo_amd
    from safebo_amd import synthetic
    import numpy as np
    eng = safebo_amd.SweepEngine(0)
    if rank == 0:
        uid = safebo_amd.SweepEngine.comm_unique_id()
        open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
    else:
        while not os.path.exists(path): time.sleep(0.05)
        uid = open(path, "rb").read()
    try:
        eng.comm_init(world, rank, uid)
        print(rank, "comm ok", flush=True)
        cfg = synthetic.make_config("A")
        eng.set_model(cfg["ds"])
        eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], [50, 50])
        r = eng.sweep_safeopt(cfg["b"])
        print(rank, eng.first, eng.n_local, r["minimizer_index"], r["count_S"], r["count_G"], flush=True)
    except Exception as e:
        print(rank, "FAILED", type(e), e, flush=True)
if __name__ == "__main__":
    mp.set_start_method("spawn")
    path = "/tmp/sbo_uid_%d" % os.getpid()
    ps = [mp.Process(target=worker, args=(r, 2, path)) for r in range(2)]
    [p.start() for p in ps]
    for p in ps:
        p.join(120)
        if p.is_alive(): p.terminate()
