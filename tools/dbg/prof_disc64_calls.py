import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "upfirdn" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"), r.get("Workgroup_Size_X")))
rows.sort()
for r in rows[-8:]:
    print(r[1] / 1e3, r[2:])
