import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if "octave_stream" in r["Kernel_Name"]]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
a=d[0::2][20:]; b=d[1::2][20:]
print("launch0 avg us", sum(a)/len(a)/1e3, "launch1 avg us", sum(b)/len(b)/1e3)
