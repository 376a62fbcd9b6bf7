#!/bin/bash
# rehearsal of bench.py --gpus 2 / 4 on the ONE GPU of the test box (RCCL refuses duplicate devices -> host relay) against the
# single-rank results of the same workloads (H strong, D strong): results must be identical
cd ${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-gpurun_out/ranks}
mkdir -p $out
python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-extra > $out/H_1.json 2> $out/H_1.err || exit 1
python bench.py --config D --steps 2 --warmup 1 --cpu-sample 0 --no-extra > $out/D_1.json 2> $out/D_1.err || exit 1
for n in 2 4; do
  SBO_BENCH_DEVICE=0 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29617 + n)) \
    bench.py --gpus $n --steps 3 --warmup 1 > $out/N_$n.json 2> $out/N_$n.err || { tail -5 $out/N_$n.err; exit 1; }
done
python - $out <<'PY'
import json, sys
out = sys.argv[1]
h1 = json.load(open(f"{out}/H_1.json")); d1 = json.load(open(f"{out}/D_1.json"))
ok = True
for n in (2, 4):
    line = [l for l in open(f"{out}/N_{n}.json") if l.startswith("{")][-1]
    r = json.loads(line)
    same_h = r["config"]["result"] == h1["config"]["result"]
    same_d = r["extra"][0]["result"] == d1["config"]["result"]
    ok = ok and same_h and same_d
    print(f"N={n}: H identical {same_h}, D identical {same_d}; transport {r['config']['collectives']}; H {r['ms_per_step']:.3f} ms/step, "
          f"comm {r['comm']['collectives_per_sweep']} calls {r['comm']['bytes_sent_per_rank']} B host_syncs {r['comm']['host_syncs_per_sweep']}; "
          f"D {r['extra'][0]['ms_per_step']:.1f} ms/step comm {r['extra'][0]['comm']['collectives_per_sweep']} calls {r['extra'][0]['comm']['bytes_sent_per_rank']} B")
    if not same_h: print("  H:", r["config"]["result"], "vs", h1["config"]["result"])
    if not same_d: print("  D:", r["extra"][0]["result"], "vs", d1["config"]["result"])
print("OK" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
PY
