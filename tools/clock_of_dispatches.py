"""Clock the chip held during each large k_ne_shared dispatch of a `rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d DIR -- ...`
run: GRBM_GUI_ACTIVE summed over the 8 XCDs / 8 / the dispatch's duration.  usage: python tools/clock_of_dispatches.py DIR"""
import csv,glob,sys,collections
f=sorted(glob.glob(sys.argv[1]+"/*/*counter_collection.csv"))[-1]
d=collections.defaultdict(lambda: {"ns":0,"c":0.0})
for r in csv.DictReader(open(f)):
    if "k_ne_shared" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE" and int(r["Grid_Size"])>100000:
        k=r["Dispatch_Id"]; d[k]["ns"]=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); d[k]["c"]+=float(r["Counter_Value"])
rows=[(int(k),v["ns"]/1e3,v["c"]/8/max(v["ns"],1)) for k,v in d.items()]
rows.sort()
for k,us,ghz in rows[-12:]: print(k, round(us,1),"us", round(ghz,3),"GHz")
