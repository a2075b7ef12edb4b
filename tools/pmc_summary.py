"""Summarise the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (tools/profile_round.sh <tag> ->
gpurun_out/prof_<tag>/{fetch,write}_<WL>/) into profiles/<round>/<tag>_pmc_summary.json and refresh
profiles/<round>/pmc_traffic.json (FETCH doubled per MI355X_MICROARCH.md).
usage: python tools/pmc_summary.py <tag> [round dir, default r2]"""
import collections, csv, json, os, sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r2"
out = {}
for wl in ("S3", "S2"):
    for kind in ("fetch", "write"):
        agg = collections.defaultdict(list)
        with open(f"gpurun_out/prof_{tag}/{kind}_{wl}/p_counter_collection.csv") as fh:
            for r in csv.DictReader(fh):
                name = r["Kernel_Name"]
                if "mmb::" not in name:
                    continue
                short = name.split("mmb::")[1].split("(")[0].split("<")[0]
                if "multi_kernel<" in name:
                    short += "<" + name.split("multi_kernel<", 1)[1].split(">")[0] + ">"
                agg[short].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out.setdefault(wl, {}).setdefault(k, {})[kind] = (sum(v) / len(v), len(v))
json.dump(out, open(f"profiles/{rnd}/{tag}_pmc_summary.json", "w"), indent=1)
p = f"profiles/{rnd}/pmc_traffic.json"
d = json.load(open(p)) if os.path.exists(p) else {}
gen = {"S3": "fused_generate_kernel", "S2": "fused_uni_generate_kernel"}
step = {}
for wl in ("S3", "S2"):
    v = out[wl][gen[wl]]
    d[wl] = int(2 * v["fetch"][0] * 1024 + v["write"][0] * 1024)
    n_steps = max(x["fetch"][1] for k, x in out[wl].items() if "train_forward" in k)
    tot, det = 0, {}
    for k, x in out[wl].items():
        if "train" in k or k.startswith("grad_weight") or k.startswith("weight_norm"):
            per_step = x["fetch"][1] / n_steps                      # launches per training step
            f = int(2 * x["fetch"][0] * 1024 * per_step)
            w = int(x["write"][0] * 1024 * per_step)
            det[k] = {"fetch_x2": f, "write": w, "launches_per_step": per_step}
            tot += f + w
    step[wl] = {"bytes_per_step": tot, "kernels": det}
    print(wl, "generation launch", d[wl], "training step", tot)
d["training_step"] = step
d["_source"] = f"profiles/{rnd}/{tag}_pmc_summary.json"
json.dump(d, open(p, "w"), indent=1)
