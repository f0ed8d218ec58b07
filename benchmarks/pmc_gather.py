"""Rows of the replay gather kernel out of a rocprofv3 --pmc counter_collection.csv (one counter per pass):
  python benchmarks/pmc_gather.py <rocprofv3 output dir> <out.csv>
Keeps Kernel_Name (shortened), Grid_Size, Counter_Name, Counter_Value, duration from the dispatch timestamps."""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "replay_gather_rows_kernel" in r["Kernel_Name"]]
w = csv.writer(open(sys.argv[2], "w"))
w.writerow(["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value", "Duration_ns"])
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    w.writerow([name, r["Grid_Size"], r["Workgroup_Size"], r["Counter_Name"], r["Counter_Value"],
                int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
print(f"{len(rows)} gather dispatches -> {sys.argv[2]}")
