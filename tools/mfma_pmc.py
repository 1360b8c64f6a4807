"""MFMA utilisation per kernel name from two rocprofv3 --pmc passes of the same command (development tool):
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d A -o run -- <cmd>
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d B -o run -- <cmd>
  python tools/mfma_pmc.py A B out.json "<cmd>"
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) summed over the launches of a
kernel name (both passes see the same launches); wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES."""
import collections, csv, glob, json, sys


def load(d):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.Counter())
    calls = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r.get("Dispatch_Id"), name)
        if key not in seen:
            seen.add(key)
            calls[name] += 1
    return acc, calls


a, calls = load(sys.argv[1])
b, _ = load(sys.argv[2])
out = {"commands": ["rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -- " + sys.argv[4],
                    "rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -- " + sys.argv[4]],
       "note": "separate --pmc passes (--kernel-trace only). mfma_utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x "
               "1024 SIMDs), counters summed over all launches of the kernel name in the run (dispatches are serialised under "
               "--pmc, so concurrent launches of the fit run one after the other here); wait_any_fraction = SQ_WAIT_ANY / "
               "SQ_WAVE_CYCLES", "kernels": {}}
for name in sorted(a, key=lambda k: -b.get(k, {}).get("GRBM_GUI_ACTIVE", 0)):
    gui = b.get(name, {}).get("GRBM_GUI_ACTIVE", 0.0)
    if gui <= 0:
        continue
    mf = a[name].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    if mf <= 0 and len(out["kernels"]) > 12:
        continue
    out["kernels"][name[:110]] = {
        "launches": calls[name], "gui_active_cycles": gui,
        "mfma_utilisation": round(mf / (gui / 8.0 * 1024.0), 4),
        "wait_any_fraction": round(a[name].get("SQ_WAIT_ANY", 0.0) / max(a[name].get("SQ_WAVE_CYCLES", 0.0), 1.0), 4)}
    if len(out["kernels"]) >= 24:
        break
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out["kernels"].items():
    print(f"{k[:80]:80s} {v['launches']:6d} mfma {v['mfma_utilisation']:.3f} wait {v['wait_any_fraction']:.3f}")
