#!/usr/bin/env python3
"""Guard-band widths and counts per rank of a sharded sweep on ONE GPU (host relay): tools/dev_rank_guard.py CONFIG WORLD [SWEEPS]"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 3 and sys.argv[1] == "--rank":
    rank, world, port, name, sweeps = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6])
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import safebo_amd
    from safebo_amd import synthetic, distributed
    dist = distributed.init_from_env()
    cfg = synthetic.make_config(name)
    q = cfg["q"]
    eng = safebo_amd.SweepEngine(0)
    distributed.join(eng, dist, relay=True)
    eng.set_model(cfg["ds"], dtype="f64", use_invK=True)
    eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], list(cfg["count"]))
    for k in range(sweeps):
        r = eng.sweep_safeopt(cfg["b"])
        p = eng.profile()
        ys = np.maximum(1.0, cfg["ds"]["Y_std"])
        print(f"rank {rank} sweep {k + 1}: kernel {p['posterior_kernel']} dm/ys {np.array(p['guard_dm'][:q]) / ys} dv/ys^2 {np.array(p['guard_dv'][:q]) / ys ** 2} "
              f"rl {np.array(p['guard_rl'][:q])} guard {r['guard_band']}/{r['guard_rechecks']}/{r['guard_passes']} first {eng.first} n_local {eng.n_local}", flush=True)
    dist.barrier()
    eng.close()
    dist.destroy()
    sys.exit(0)
name, world = sys.argv[1], int(sys.argv[2])
sweeps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import socket
with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
procs = [subprocess.Popen([sys.executable, __file__, "--rank", str(r), str(world), port, name, str(sweeps)]) for r in range(world)]
sys.exit(max(p.wait(timeout=600) for p in procs))
