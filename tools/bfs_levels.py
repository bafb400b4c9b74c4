import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "glob_" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[-33:]
print(" ".join("%s:%.0f"%("C" if "claim" in r["Kernel_Name"] else ("W" if "win" in r["Kernel_Name"] else "o"),(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in rows))
