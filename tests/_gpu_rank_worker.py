"""Worker for the 2-rank GPU rehearsal in tests/test_gpu_parity.py: both ranks drive the single GPU of the
test box (RCCL refuses duplicate devices, so the collectives ride the host relay over the package's TCP rendezvous); the sweep code
path -- shard offsets, whole-grid U mask, exchange kernels, host merge -- is the multi-GPU one."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out_path, cfg_name, n, count, b = (int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4],
                                                          sys.argv[5], int(sys.argv[6]), json.loads(sys.argv[7]), json.loads(sys.argv[8]))
    if isinstance(b, list):
        return sequence(rank, world, port, out_path, cfg_name, n, count, b)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import safebo_amd
    from safebo_amd import synthetic, distributed

    dist = distributed.init_from_env()
    dtype, guard = "f64", None
    if cfg_name.endswith(":f32"):                      # an fp32 model (library Cholesky): the fp64 recheck across ranks
        cfg_name, dtype = cfg_name[:-4], "f32"
    if cfg_name.endswith(":guard"):                    # every sweep re-evaluates its guard-band candidates (option guard_band 2)
        cfg_name, guard = cfg_name[:-6], 2
    cfg = synthetic.make_config(cfg_name, n=n)
    eng = safebo_amd.SweepEngine(0)
    distributed.join(eng, dist, relay=True)
    if guard is not None:
        eng.set_option("guard_band", guard)
    eng.set_model(cfg["ds"], dtype=dtype, use_invK=(dtype == "f64"))
    eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
    res = eng.sweep_safeopt(b, want_masks=True)
    res["fp64_rechecks"] = int(eng.profile()["fp64_rechecks"])
    res["posterior_kernel"] = int(eng.profile()["posterior_kernel"])
    res["comm_bytes"], res["comm_calls"] = int(eng.profile()["comm_bytes"]), int(eng.profile()["comm_calls"])
    masks = {k: eng.mask(k) for k in ("S", "U", "M")}
    masks.update({f"G{c}": eng.mask("G", c) for c in range(1, cfg["q"])})
    gres = None
    tres = None
    if cfg["q"] > 1:
        try:
            gres = eng.sweep_goose(b, want_masks=True, posterior_ready=(dtype == "f64" and guard is None))
            gres["fp64_rechecks"] = int(eng.profile()["fp64_rechecks"])
            masks.update({f"O{c}": eng.mask("O", c) for c in range(1, cfg["q"])})
        except safebo_amd.EmptySafeSetError:
            gres = {"empty_safe_set": True}
    if dtype == "f32" or guard is not None:
        # the trust-region sweep of GP_TR (ball of radius 0.3 of the box around its centre) across the ranks
        x0 = cfg["bound"].mean(axis=1)
        tres = eng.sweep_tr(b, x0, 0.3 * float(np.min(cfg["bound"][:, 1] - cfg["bound"][:, 0])))
    np.savez(out_path + f".rank{rank}.npz", first=eng.first, n_local=eng.n_local, kernel=res["posterior_kernel"], **masks)
    if rank == 0:
        def plain(r):
            return {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in r.items()}
        json.dump({**plain(res), "goose": plain(gres) if gres is not None else None, "tr": plain(tres) if tres is not None else None},
                  open(out_path, "w"))
    dist.barrier()
    eng.close()
    dist.destroy()


def sequence(rank, world, port, out_path, cfg_name, n, count, bs):
    """Several sweeps in a row on one sharded grid, each with its own confidence multiplier (a larger b = larger radii): the
    first sizes its halo windows from its own keys (the host waits for them), the later ones from the previous sweep's keys
    (speculative: no wait inside the sweep; a window that turns out too narrow reruns the set phase)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
    import safebo_amd
    from safebo_amd import synthetic, distributed

    dist = distributed.init_from_env()
    cfg = synthetic.make_config(cfg_name, n=n)
    eng = safebo_amd.SweepEngine(0)
    distributed.join(eng, dist, relay=True)
    eng.set_grid_sharded(cfg["bound"][:, 0], cfg["bound"][:, 1], count)
    rows, masks = [], {}
    for i, b in enumerate(bs):
        eng.set_model(cfg["ds"], dtype="f64")
        res = eng.sweep_safeopt(b, want_masks=True)
        p = eng.profile()
        for k in ("S", "M"):
            masks[f"{i}_{k}"] = eng.mask(k)
        for c in range(1, cfg["q"]):
            masks[f"{i}_G{c}"] = eng.mask("G", c)
        if os.environ.get("SBO_TEST_SEQ_SAFEOPT_ONLY"):
            g, pg = {"target_index": -1, "explore_index": -1, "count_O": []}, p
        else:
            g = eng.sweep_goose(b, want_masks=True, posterior_ready=True)
            pg = eng.profile()
            for c in range(1, cfg["q"]):
                masks[f"{i}_O{c}"] = eng.mask("O", c)
        rows.append({"b": b, "host_syncs": int(p["host_syncs"]), "goose_host_syncs": int(pg["host_syncs"]),
                     "halo_reruns": int(p["halo_reruns"]), "goose_halo_reruns": int(pg["halo_reruns"]), "comm_calls": int(p["comm_calls"]),
                     "minimizer_index": res["minimizer_index"], "expander_index": res["expander_index"], "count_S": res["count_S"],
                     "count_M": res["count_M"], "count_G": [int(x) for x in res["count_G"]], "u_star": res["u_star"],
                     "target_index": g["target_index"], "explore_index": g["explore_index"], "count_O": [int(x) for x in g["count_O"]]})
    np.savez(out_path + f".rank{rank}.npz", **masks)
    if rank == 0:
        json.dump(rows, open(out_path, "w"))
    dist.barrier()
    eng.close()
    dist.destroy()


if __name__ == "__main__":
    main()
