"""Per-kernel means of rocprofv3 counter passes, as a small CSV for profiles/.

    python tools/pmc_summary.py OUT.csv DIR [DIR ...] [--only SUBSTR] [--traffic KEY --kernel SUBSTR]

Every DIR is the output directory of one `rocprofv3 --pmc <counters> --kernel-trace -d DIR -- <cmd>`
pass (counters collected in passes of their own: gpurun refuses --pmc together with trace domains
other than the kernel trace).  Rows: kernel, counter, mean value per dispatch, dispatches, mean
duration (us).  With --traffic the corrected HBM bytes per launch of the kernel matching --kernel
are stored in profiles/traffic_select.json under KEY (bench.py's `roofline.traffic_key`):
bytes = 2 x FETCH_SIZE + WRITE_SIZE, both KiB - on gfx950 FETCH_SIZE tallies the 128-byte requests
of 16-B-per-lane reads at 64 B (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("az::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    cut = name.find("(")
    return name[:cut] if cut > 0 else name


def main():
    args = sys.argv[1:]
    out, dirs, only, key, ksub = args[0], [], None, None, None
    i = 1
    while i < len(args):
        if args[i] == "--only":
            only = args[i + 1]; i += 2
        elif args[i] == "--traffic":
            key = args[i + 1]; i += 2
        elif args[i] == "--kernel":
            ksub = args[i + 1]; i += 2
        else:
            dirs.append(args[i]); i += 1
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for r in csv.DictReader(fh):
                    k = short(r["Kernel_Name"])
                    if only and only not in k:
                        continue
                    a = acc[(k, r["Counter_Name"])]
                    a[0] += float(r["Counter_Value"]); a[1] += 1
                    a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    rows = sorted(acc.items())
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "counter", "mean_per_dispatch", "dispatches", "mean_duration_us"])
        for (k, c), (tot, n, us) in rows:
            w.writerow([k, c, "%.3f" % (tot / n), n, "%.2f" % (us / n)])
    print("wrote", out, len(rows), "rows")
    if key:
        fetch = [v for (k, c), v in acc.items() if c == "FETCH_SIZE" and ksub in k]
        write = [v for (k, c), v in acc.items() if c == "WRITE_SIZE" and ksub in k]
        if not fetch or not write:
            raise SystemExit("no FETCH_SIZE / WRITE_SIZE rows for a kernel matching %r" % ksub)
        f_kib = sum(v[0] for v in fetch) / sum(v[1] for v in fetch)
        w_kib = sum(v[0] for v in write) / sum(v[1] for v in write)
        path = os.path.join(ROOT, "profiles", "traffic_select.json")
        doc = json.load(open(path)) if os.path.exists(path) else {}
        doc.setdefault("by_config", {})[key] = {
            "kernel": ksub, "hbm_bytes_per_launch": int((2 * f_kib + w_kib) * 1024), "fetch_size_kib_raw": round(f_kib, 1),
            "write_size_kib": round(w_kib, 1), "launches": int(sum(v[1] for v in fetch)), "source": os.path.basename(out)}
        doc["correction"] = ("gfx950 FETCH_SIZE reports half the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, HBM "
                             "section): bytes = 2 x FETCH_SIZE + WRITE_SIZE.  The selection kernel reads 16-byte halves of "
                             "scattered 32-byte records, a pattern the guide calls uncalibrated: +-20 %.")
        json.dump(doc, open(path, "w"), indent=1)
        print("traffic[%s] = %d bytes per launch" % (key, doc["by_config"][key]["hbm_bytes_per_launch"]))


if __name__ == "__main__":
    main()
