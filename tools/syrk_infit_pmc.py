"""Fabric traffic of the trailing update INSIDE a fit (round 6): FETCH_SIZE and WRITE_SIZE of every syrk_mirror_kernel<64>
launch of `bench.py --steps 1 --warmup 1`, collected in two separate rocprofv3 --pmc passes (the counters do not fit one
pass), against the algorithmic bytes of the same launches -- 24 bytes per element of the 128 x 64 tiles a launch covers
(8 read, 8 written, 8 written again as the mirror; a launch's tiles = its workgroups), i.e. the 12 m^2 of a whole
triangle. KiB -> bytes; no gfx950 doubling: this kernel's 8-byte-per-lane loads read the known 4 m^2 bytes exactly in the
isolated probe (profiles/r05/r05u_syrk_traffic_pmc.json).

    python tools/syrk_infit_pmc.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> out.json "<command>"
"""
import csv, glob, json, os, sys


def collect(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "syrk_mirror_kernel<64" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        key = int(r["Dispatch_Id"])
        wg = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
        by[key] = (wg, by.get(key, (0, 0.0))[1] + 1024.0 * float(r["Counter_Value"]))
    return [by[k] for k in sorted(by)]


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    n = min(len(fetch), len(write))
    # (warm-up fit + timed fit: the second half of the launches is the timed fit; both halves are the same sequence)
    half = n // 2
    res = {"command": sys.argv[4] if len(sys.argv) > 4 else "", "launches_per_fit": half,
           "note": "sum over all syrk_mirror_kernel<64> launches of the second (timed) fit; algorithmic = 24 B x 128 x 64 x workgroups"}
    groups = {}
    tot_f = tot_w = tot_a = 0.0
    for (wg, fb), (wg2, wb) in zip(fetch[half:n], write[half:n]):
        alg = 24.0 * 128 * 64 * wg
        tot_f += fb; tot_w += wb; tot_a += alg
    res.update(FETCH_SIZE_bytes=tot_f, WRITE_SIZE_bytes=tot_w, algorithmic_bytes=tot_a,
               traffic_over_algorithmic=round((tot_f + tot_w) / max(tot_a, 1.0), 4))
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
