import csv, sys, collections, re
path = sys.argv[1]; steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(lambda: [0, 0.0])
with open(path) as f:
    rows = sorted(csv.DictReader(f), key=lambda r: int(r["Start_Timestamp"]))
# everything before the first fbank launch is process setup (model.to(device), the ParamArena re-homing 620 parameter tensors:
# ~1000 __amd_rocclr_copyBuffer launches), not part of any step
first = next((i for i, r in enumerate(rows) if "fbank_kernel" in (r.get("Kernel_Name") or r.get("Name"))), 0)
if True:
    for r in rows[first:]:
        name = r.get("Kernel_Name") or r.get("Name")
        if "spin_kernel" in name:          # torch.cuda._sleep: parks the GPU during bench.py's event-bracketed steps
            continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        name = re.sub(r"\(.*", "", name)[:110]
        agg[name][0] += 1; agg[name][1] += d
adam = [v[0] for k, v in agg.items() if "adam_kernel" in k]
if adam:
    steps = float(adam[0])                 # optimizer steps in the trace = launches of the Adam kernel
tot = sum(v[1] for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over trace; per step {tot/1e3/steps:.2f} ms")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{t/steps/1e3:8.3f} ms/step {c/steps:7.1f} calls/step {t/c:8.1f} us  {n}")
