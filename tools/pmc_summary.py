"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv, sys, collections
def main(paths):
    for path in paths:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        dur = collections.defaultdict(dict)
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"]
                if "cfnerf" not in name:
                    continue
                short = name.split("(")[0].replace("void ", "").replace("cfnerf::", "")
                acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
                dur[short][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"# {path}")
        for k in sorted(acc, key=lambda k: -sum(dur[k].values())):
            d = list(dur[k].values())
            line = f"{k:42s} n={len(d):3d} avg_us={sum(d)/len(d):9.1f}"
            for c, v in sorted(acc[k].items()):
                line += f"  {c}={sum(v)/len(v):.4g}"
            print(line)
if __name__ == "__main__":
    main(sys.argv[1:])
