import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"])
print(json.dumps(d.get("roofline_sa_linear"), indent=0)[:1800])
print({k: d["roofline"][k] for k in ("achieved", "frac", "launches_per_step", "ms_per_step_in_kernel")})
