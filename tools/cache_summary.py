"""Per-kernel averages (per launch) of the counters collected by tools/profile_cache.sh."""
import collections
import csv
import glob
import sys

KERNELS = ("level_kernel", "fast_kernel", "distribute_kernel", "describe_kernel", "stereo_frame_kernel", "bf_knn2_mfma_kernel")


def main(out):
    rows = collections.defaultdict(dict)
    for g in sorted(glob.glob(out + "/g*/p_counter_collection.csv")):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        n = collections.defaultdict(set)
        for r in csv.DictReader(open(g)):
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
                    n[k].add(r["Dispatch_Id"])
        for k, v in acc.items():
            for c, x in v.items():
                rows[k][c] = x / len(n[k])
    cols = sorted({c for v in rows.values() for c in v})
    print("kernel," + ",".join(cols))
    for k in KERNELS:
        if k in rows:
            print(k + "," + ",".join("%.4g" % rows[k].get(c, float("nan")) for c in cols))


if __name__ == "__main__":
    main(sys.argv[1])
