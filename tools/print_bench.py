"""Print the headline fields of bench.py JSON lines: python tools/print_bench.py file.json [...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.load(open(f))
    r = d["roofline"]
    line = f"{f}: value {d['value']:.4e} ms/step {d['ms_per_step']:.4f} K1 {r['kernel_ms']:.4f} device {r['device_ms_per_step']:.4f} set {r['hbm']['set_phase_ms']:.4f} (chain {r['hbm'].get('chain_ms', 0):.4f} exposed {r['hbm'].get('exposed', {}).get('set_phase_ms', 0):.4f}) frac {r['frac']:.3f} hbm {r['hbm']['frac']:.3f}"
    if "iteration" in d:
        it = d["iteration"]
        line += f" | iteration {it['ms_per_step']:.3f} ms (set_model {it['set_model_ms']:.3f}, sweep {it['sweep_call_ms']:.3f}) slow {it.get('slow_steps')}"
    print(line)
    for e in d.get("extra", []):
        line = f"   {e['config'][:28]}: value {e['value']:.4e} ms {e['ms_per_step']:.4f} K1 {e['roofline']['kernel_ms']:.3f} frac {e['roofline']['frac']:.3f} hbm {e['roofline_hbm']['frac']:.3f} set {e['roofline_hbm']['set_phase_ms']:.3f} (chain {e['roofline_hbm'].get('chain_ms', 0):.3f} exposed {e['roofline_hbm'].get('exposed', {}).get('ms', 0):.3f}) device {e['roofline']['device_ms_per_step']:.4f}"
        if "iteration" in e:
            it = e["iteration"]
            line += f" | iteration {it['ms_per_step']:.3f} ms (set_model {it['set_model_ms']:.3f}, sweep {it['sweep_call_ms']:.3f})"
        if "fp64_recheck" in e:
            line += f" | fp64 recheck {e['fp64_recheck']['candidates_reevaluated']} candidates, {e['fp64_recheck']['ms']:.2f} ms"
        print(line)
